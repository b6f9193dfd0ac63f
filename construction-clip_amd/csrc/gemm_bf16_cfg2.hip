// tile configuration 2: 256x128 block (4x2 waves), 3 LDS stages (144 KiB -> 1 block/CU, 2 tiles of DMA in flight)
#include "gemm_bf16_impl.h"
namespace CCLIP_NS {
bool cclip_gemm_launch_cfg2(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  return gemm_launch_cfg<4, 2, 3, 4>(lay, act, grid, stream, a);
}
}  // namespace CCLIP_NS
