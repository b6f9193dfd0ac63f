// Skinny GEMM (M <= 8 rows) for K-strided weights W[K][N] (GPT-2 Conv1D layout): the four projections of a KV-cached
// decode step, where M = number of beams.  The MFMA tile kernels launch N/128 workgroups for such a call (6 for the
// 768-wide projections) and each then streams its 128 x K weight panel through one CU at ~40 GB/s: 20-50 us per GEMM, 70 %
// of the decode step (rocprofv3, tools/decode_bench.py).  This path is weight-read bound by construction, so it is written
// as a GEMV: one workgroup per 32-column block (N/32 workgroups), full K per workgroup, every 64-byte row segment read
// exactly once with 16-byte loads, fp32 FMA against the (tiny) activation rows held in LDS, one deterministic cross-lane
// reduction at the end; bias / activation / fp32 residual / fp32 or 16-bit output as in the tile kernels' epilogue.
#include "gemm_skinny_impl.h"

namespace CCLIP_NS {

template <int MCAP, int ACT>
__global__ __launch_bounds__(256) void gemm_skinny_ks_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];
  skinny_block<MCAP, ACT, 12>(p, blockIdx.x * 32, sk_lds);
}

// returns false when the call is not a skinny K-strided GEMM this path covers (the caller then uses the tile kernels)
bool cclip_gemm_launch_skinny(int lay, int act, hipStream_t stream, const GemmArgs& a) {
  if (lay != 2 || a.M > 8 || a.split_ws || a.aux || (a.N & 7)) return false;
  const size_t lds_a = (size_t)a.M * a.K * sizeof(float);
  const int mcap = a.M <= 4 ? 4 : 8;
  const size_t lds_r = (size_t)256 * mcap * 8 * sizeof(float);
  const size_t lds = lds_a > lds_r ? lds_a : lds_r;
  if (lds > 96 * 1024) return false;
  dim3 grid((a.N + 31) / 32), block(256);
#define SK(MC, ACTV)                                                                                              \
  do {                                                                                                            \
    static size_t attr = 0;                                                                                       \
    if (lds > attr) {                                                                                             \
      hipFuncSetAttribute((const void*)gemm_skinny_ks_kernel<MC, ACTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
      attr = lds;                                                                                                 \
    }                                                                                                             \
    hipLaunchKernelGGL((gemm_skinny_ks_kernel<MC, ACTV>), grid, block, lds, stream, a);                           \
    return true;                                                                                                  \
  } while (0)
#define SKM(ACTV) do { if (mcap == 4) SK(4, ACTV); else SK(8, ACTV); } while (0)
  switch (act) {
    case CCLIP_ACT_NONE: SKM(CCLIP_ACT_NONE);
    case CCLIP_ACT_GELU_NEW: SKM(CCLIP_ACT_GELU_NEW);
    default: return false;
  }
#undef SKM
#undef SK
}

}  // namespace CCLIP_NS
