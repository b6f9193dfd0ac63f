// Skinny GEMM (M <= 8 rows) for K-strided weights W[K][N] (GPT-2 Conv1D layout): the four projections of a KV-cached
// decode step, where M = number of beams.  The MFMA tile kernels launch N/128 workgroups for such a call (6 for the
// 768-wide projections) and each then streams its 128 x K weight panel through one CU at ~40 GB/s: 20-50 us per GEMM, 70 %
// of the decode step (rocprofv3, tools/decode_bench.py).  This path is weight-read bound by construction, so it is written
// as a GEMV: one workgroup per 32-column block (N/32 workgroups), full K per workgroup, every 64-byte row segment read
// exactly once with 16-byte loads, fp32 FMA against the (tiny) activation rows held in LDS, one deterministic cross-lane
// reduction at the end; bias / activation / fp32 residual / fp32 or 16-bit output as in the tile kernels' epilogue.
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

template <int MCAP, int ACT>
__global__ __launch_bounds__(256) void gemm_skinny_ks_kernel(const GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) float sk_lds[];     // A as fp32 [M][K]; reused for the reduction
  constexpr int U = 12;                                              // weight rows in flight per lane (K = 768: all of them)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.M, K = p.K;
  const int n0 = blockIdx.x * 32;
  const int slot = wave * 16 + (lane >> 2);          // 64 row slots per step
  const int chunk = lane & 3;                        // 8 columns each
  const int ncol = n0 + chunk * 8;
  const bool live = ncol < p.N;                      // N % 8 == 0: a chunk is all in or all out
  const bf16* wp = p.B + ncol;
  // first batch of weight rows goes out before anything else: the A rows are staged under its latency
  bf16x8 w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = slot + 64 * u;
#pragma unroll
    for (int j = 0; j < 8; ++j) w[u][j] = (bf16)0.f;
    if (live && k < K) w[u] = *(const bf16x8*)(wp + (long)k * p.ldb);
  }
  if (p.ln_x) {
    // A = LayerNorm(x) (eps 1e-5, fp32 statistics), rounded to the 16-bit operand type exactly as the stand-alone
    // LayerNorm kernel's output would be; wave w normalises rows w, w+4
    for (int m = wave; m < M; m += 4) {
      const float* xr = p.ln_x + (long)m * p.ln_ldx;
      float s1 = 0.f;
      for (int k = lane; k < K; k += 64) s1 += xr[k];
      const float mean = wave_sum(s1) / (float)K;
      float s2 = 0.f;
      for (int k = lane; k < K; k += 64) { const float d = xr[k] - mean; s2 += d * d; }
      const float rstd = rsqrtf(wave_sum(s2) / (float)K + 1e-5f);
      for (int k = lane; k < K; k += 64)
        sk_lds[m * K + k] = (float)(bf16)((xr[k] - mean) * rstd * p.ln_gamma[k] + p.ln_beta[k]);
    }
  } else {
    for (int i = tid; i < M * K; i += 256) {
      const int m = i / K, k = i - m * K;
      sk_lds[i] = (float)p.A[(long)m * p.lda + k];
    }
  }
  __syncthreads();
  float acc[MCAP][8];
#pragma unroll
  for (int m = 0; m < MCAP; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[m][j] = 0.f;
  for (int k0 = slot; k0 < K; k0 += 64 * U) {
    if (k0 != slot) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = k0 + 64 * u;
#pragma unroll
        for (int j = 0; j < 8; ++j) w[u][j] = (bf16)0.f;
        if (live && k < K) w[u] = *(const bf16x8*)(wp + (long)k * p.ldb);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 64 * u;
      if (k < K) {
#pragma unroll
        for (int m = 0; m < MCAP; ++m) {
          if (m < M) {
            const float a = sk_lds[m * K + k];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[m][j] += a * (float)w[u][j];
          }
        }
      }
    }
  }
  __syncthreads();                                   // everyone is done with the A rows: the buffer becomes [256][MCAP*8]
#pragma unroll
  for (int m = 0; m < MCAP; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) sk_lds[tid * (MCAP * 8) + m * 8 + j] = acc[m][j];
  __syncthreads();
  // thread (m, c) sums its column over the 64 row slots, in slot order (deterministic)
  if (tid < M * 32) {
    const int m = tid >> 5, c = tid & 31;
    const int n = n0 + c;
    if (n < p.N) {
      float s = 0.f;
      for (int sl = 0; sl < 64; ++sl) s += sk_lds[(sl * 4 + (c >> 3)) * (MCAP * 8) + m * 8 + (c & 7)];
      float v = s * p.alpha + (p.bias ? p.bias[n] : 0.f);
      if (p.out_pre) p.out_pre[(long)m * p.ldc + n] = (bf16)v;
      v = act_apply<ACT>(v, 0.f);
      if (p.residual) v += p.residual[(long)m * p.ldr + n];
      if (p.out_f32) p.out_f32[(long)m * p.ldc + n] = v;
      if (p.out_bf16) p.out_bf16[(long)m * p.ldc + n] = (bf16)v;
      if (p.kv_k && n >= p.kv_width) {                 // packed q|k|v projection: k and v rows also go to the cache
        if (n < 2 * p.kv_width) p.kv_k[(long)m * p.kv_ld_seq + n - p.kv_width] = (bf16)v;
        else p.kv_v[(long)m * p.kv_ld_seq + n - 2 * p.kv_width] = (bf16)v;
      }
    }
  }
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
