// The 32-column block of the skinny (M <= 8) K-strided GEMV, shared by the stand-alone kernel (gemm_bf16_skinny.hip) and the
// persistent decode kernel (decode_persist.hip).
#pragma once
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

// Coherent (agent-scope) accesses for buffers that workgroups on different XCDs hand to each other INSIDE one kernel (the
// persistent decode kernel): relaxed agent-scope atomics compile to sc1 loads / write-through sc1 stores, which are coherent
// across the XCDs' L2s without the cache-wide write-back + invalidate of a release / acquire fence pair.  COH = false: plain.
template <bool COH, typename T>
__device__ __forceinline__ T ld_coh(const T* p) {
  if constexpr (!COH) {
    return *p;
  } else if constexpr (sizeof(T) == 4) {
    return __builtin_bit_cast(T, __hip_atomic_load((unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else if constexpr (sizeof(T) == 2) {
    return __builtin_bit_cast(T, __hip_atomic_load((unsigned short*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else {
    static_assert(sizeof(T) == 16, "4-, 2- or 16-byte objects");
    struct U2 { unsigned long a, b; } u;
    u.a = __hip_atomic_load((unsigned long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    u.b = __hip_atomic_load((unsigned long*)p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_bit_cast(T, u);
  }
}
template <bool COH, typename T>
__device__ __forceinline__ void st_coh(T* p, T v) {
  if constexpr (!COH) {
    *p = v;
  } else if constexpr (sizeof(T) == 4) {
    __hip_atomic_store((unsigned*)p, __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else if constexpr (sizeof(T) == 2) {
    __hip_atomic_store((unsigned short*)p, __builtin_bit_cast(unsigned short, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    static_assert(sizeof(T) == 16, "4-, 2- or 16-byte objects");
    struct U2 { unsigned long a, b; };
    const U2 u = __builtin_bit_cast(U2, v);
    __hip_atomic_store((unsigned long*)p, u.a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store((unsigned long*)p + 1, u.b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// One 32-column block (columns n0..n0+31) of the skinny GEMM, executed by a 256-thread workgroup; sk_lds: A as fp32 [M][K],
// reused for the reduction (max(M*K, 256*MCAP*8) floats).  U: weight rows in flight per lane (K = 768: 12 = all of them).
// Every thread of the workgroup must call it (three workgroup barriers inside).  COH: activations in / out through ld_coh / st_coh.
struct SkinnyNoWait { __device__ __forceinline__ void operator()() const {} };

// `after_weights()` is called once the first batch of weight loads is in flight and before any activation is read: the persistent
// kernel waits for the previous phase THERE, so the weight fetch overlaps the wait (weights do not depend on the previous phase).
template <int MCAP, int ACT, int U, bool COH = false, typename Wait = SkinnyNoWait>   // U: a multiple of 4
__device__ __forceinline__ void skinny_block(const GemmArgs& p, const int n0, float* sk_lds, Wait after_weights = Wait()) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = p.M, K = p.K;
  const int slot = wave * 16 + (lane >> 2);          // 64 row slots per step
  const int chunk = lane & 3;                        // 8 columns each
  const int ncol = n0 + chunk * 8;
  const bool live = ncol < p.N;                      // N % 8 == 0: a chunk is all in or all out
  const bf16* wp = p.B + ncol;
  // first batch of weight rows goes out before anything else: the A rows are staged under its latency
  // the epilogue's bias / residual element of thread (m, c) is requested now: at the end it would be one more memory round trip
  float pre_bias = 0.f, pre_res = 0.f;
  {
    const int n = n0 + (tid & 31);
    if (tid < M * 32 && n < p.N && p.bias) pre_bias = p.bias[n];
  }
  // (loads are unconditional on clamped addresses and masked afterwards: a predicated load compiles to load - wait - use,
  // one memory round trip per row instead of one per batch)
  const bf16* wpc = p.B + (live ? ncol : 0);
  bf16x8 w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int k = slot + 64 * u;
    w[u] = *(const bf16x8*)(wpc + (long)(k < K ? k : K - 1) * p.ldb);
  }
  after_weights();
  {
    const int m = tid >> 5, n = n0 + (tid & 31);
    if (tid < M * 32 && n < p.N && p.residual) pre_res = ld_coh<COH>(p.residual + (long)m * p.ldr + n);
  }
  if (p.ln_x) {
    // A = LayerNorm(x) (eps 1e-5, fp32 statistics), rounded to the 16-bit operand type exactly as the stand-alone
    // LayerNorm kernel's output would be; wave w normalises rows w, w+4.  Rows of up to 1024 columns are read ONCE (16
    // values per lane, all loads in flight together): three dependent passes over the row were three memory round trips.
    for (int m = wave; m < M; m += 4) {
      const float* xr = p.ln_x + (long)m * p.ln_ldx;
      if (K <= 1024) {
        float xv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int k = lane + 64 * u; xv[u] = ld_coh<COH>(xr + (k < K ? k : 0)); }   // unconditional: all in flight
        float s1 = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (lane + 64 * u < K) s1 += xv[u];
        const float mean = wave_sum(s1) / (float)K;
        float s2 = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (lane + 64 * u < K) { const float d = xv[u] - mean; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) / (float)K + 1e-5f);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int k = lane + 64 * u;
          if (k < K) sk_lds[m * K + k] = (float)(bf16)((xv[u] - mean) * rstd * p.ln_gamma[k] + p.ln_beta[k]);
        }
      } else {
        float s1 = 0.f;
        for (int k = lane; k < K; k += 64) s1 += ld_coh<COH>(xr + k);
        const float mean = wave_sum(s1) / (float)K;
        float s2 = 0.f;
        for (int k = lane; k < K; k += 64) { const float d = ld_coh<COH>(xr + k) - mean; s2 += d * d; }
        const float rstd = rsqrtf(wave_sum(s2) / (float)K + 1e-5f);
        for (int k = lane; k < K; k += 64)
          sk_lds[m * K + k] = (float)(bf16)((ld_coh<COH>(xr + k) - mean) * rstd * p.ln_gamma[k] + p.ln_beta[k]);
      }
    }
  } else if (!(K & 7) && !(p.lda & 7)) {
    // 16-byte pieces, eight per thread in flight (an element-at-a-time loop is one memory round trip per iteration)
    const int K8 = K >> 3, n8 = M * K8;
    for (int i0 = tid; i0 < n8; i0 += 2048) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u < n8 ? i0 + 256 * u : n8 - 1;
        const int m = i / K8, c = i - m * K8;
        v[u] = ld_coh<COH>((const bf16x8*)(p.A + (long)m * p.lda + 8 * c));
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 256 * u;
        if (i < n8) {
          const int m = i / K8, c = i - m * K8;
#pragma unroll
          for (int j = 0; j < 8; ++j) sk_lds[m * K + 8 * c + j] = (float)v[u][j];
        }
      }
    }
  } else {
    for (int i = tid; i < M * K; i += 256) {
      const int m = i / K, k = i - m * K;
      sk_lds[i] = (float)ld_coh<COH>(p.A + (long)m * p.lda + k);
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
        w[u] = *(const bf16x8*)(wpc + (long)(k < K ? k : K - 1) * p.ldb);
      }
    }
    // branch-free: rows past K and columns past N are masked to zero weights, activation rows past M repeat row M-1 (never
    // stored) - with a branch per row the LDS read of every activation was exposed latency (36 x ~130 clocks per block)
#pragma unroll
    for (int u0 = 0; u0 < U; u0 += 4) {                 // four rows at a time: 4 x MCAP LDS reads in flight, then their FMAs
      float av[4][MCAP];
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const int k = k0 + 64 * (u0 + uu);
#pragma unroll
        for (int m = 0; m < MCAP; ++m) av[uu][m] = sk_lds[(m < M ? m : M - 1) * K + (k < K ? k : K - 1)];
      }
#pragma unroll
      for (int uu = 0; uu < 4; ++uu) {
        const int k = k0 + 64 * (u0 + uu);
        const bool on = live && k < K;
        float wf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = on ? (float)w[u0 + uu][j] : 0.f;
#pragma unroll
        for (int m = 0; m < MCAP; ++m)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m][j] += av[uu][m] * wf[j];
      }
    }
  }
  // reduction over the 64 row slots: DPP adds over the 4 slots of a 16-lane row (row_shr 4, 8: lanes 12-15 of the row then hold
  // the row's sums per column chunk), the 16 rows' partials through LDS, summed in row order - a fixed order, so results are
  // reproducible run to run.  (64 VALU adds instead of 128 ds_bpermute.)
#pragma unroll
  for (int m = 0; m < MCAP; ++m)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[m][j];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
      acc[m][j] = v;
    }
  __syncthreads();                                   // everyone is done with the A rows: the buffer becomes [16 rows][4 chunks][MCAP*8]
  if ((lane & 15) >= 12) {
    float* dst = sk_lds + ((wave * 4 + (lane >> 4)) * 4 + chunk) * (MCAP * 8);
#pragma unroll
    for (int m = 0; m < MCAP; ++m)
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[m * 8 + j] = acc[m][j];
  }
  __syncthreads();
  if (tid < M * 32) {
    const int m = tid >> 5, c = tid & 31;
    const int n = n0 + c;
    if (n < p.N) {
      float pr[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) pr[r] = sk_lds[(r * 4 + (c >> 3)) * (MCAP * 8) + m * 8 + (c & 7)];
      float s = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) s += pr[r];
      float v = s * p.alpha + pre_bias;
      if (p.out_pre) st_coh<COH>(p.out_pre + (long)m * p.ldc + n, (bf16)v);
      v = act_apply<ACT>(v, 0.f);
      v += pre_res;
      if (p.out_f32) st_coh<COH>(p.out_f32 + (long)m * p.ldc + n, v);
      if (p.out_bf16) st_coh<COH>(p.out_bf16 + (long)m * p.ldc + n, (bf16)v);
      if (p.kv_k && n >= p.kv_width) {                 // packed q|k|v projection: k and v rows also go to the cache
        if (n < 2 * p.kv_width) st_coh<COH>(p.kv_k + (long)m * p.kv_ld_seq + n - p.kv_width, (bf16)v);
        else st_coh<COH>(p.kv_v + (long)m * p.kv_ld_seq + n - 2 * p.kv_width, (bf16)v);
      }
    }
  }
}


}  // namespace CCLIP_NS
