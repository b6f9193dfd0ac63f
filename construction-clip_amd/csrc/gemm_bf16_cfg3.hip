// tile configuration 3: 256x256 block, 2x4 waves of 128x64, 2 LDS stages (128 KiB -> 1 block/CU):
// half the DMA and LDS-read instructions per MFMA of the 64x64-per-wave configurations
#include "gemm_bf16_impl.h"
namespace CCLIP_NS {
bool cclip_gemm_launch_cfg3(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  return gemm_launch_cfg<2, 4, 2, 8>(lay, act, grid, stream, a);
}
}  // namespace CCLIP_NS
