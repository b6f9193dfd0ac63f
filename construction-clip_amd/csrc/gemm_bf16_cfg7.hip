// tile configuration 7: configuration 3 (256x256 block, 2x4 waves of 128x64, 2 LDS stages) with the ROTATED K loop
// (gemm_bf16_impl.h, k_iter_rot): after every barrier the waves multiply k-step 1 of the previous K-tile, whose fragments are in
// registers, while the new tile's fragments are read - the post-barrier LDS read burst leaves the critical path.
// Forward operand layout only (with the transposed weight shadows that includes every dgrad GEMM).
#include "gemm_bf16_impl.h"
namespace CCLIP_NS {
bool cclip_gemm_launch_cfg7(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  return gemm_launch_cfg<2, 4, 2, 8, 1>(lay, act, grid, stream, a);
}
}  // namespace CCLIP_NS
