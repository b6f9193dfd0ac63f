// tile configuration 1: 128x128 block (2x2 waves), 2 LDS stages (64 KiB -> 2 blocks/CU)
#include "gemm_bf16_impl.h"
namespace CCLIP_NS {
bool cclip_gemm_launch_cfg1(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  return gemm_launch_cfg<2, 2, 2, 4>(lay, act, grid, stream, a);
}
}  // namespace CCLIP_NS
