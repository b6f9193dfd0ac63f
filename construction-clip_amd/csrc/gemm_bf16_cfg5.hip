// tile configuration 5: 192x256 block, 2x4 waves of 96x64, 2 LDS stages (112 KiB -> 1 block/CU).  Same K loop as
// configuration 3 with 6 instead of 8 m-tiles per wave: exists for tile-count quantisation - M = 51 200 rows x 9 column tiles
// is 7.03 rounds of 256 x 256 tiles over the 256 CUs (8 rounds of work) but 9.39 rounds of 192 x 256 (10 x 0.75 = 7.5).
#include "gemm_bf16_impl.h"
namespace CCLIP_NS {
bool cclip_gemm_launch_cfg5(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  return gemm_launch_cfg<2, 4, 2, 6>(lay, act, grid, stream, a);
}
}  // namespace CCLIP_NS
