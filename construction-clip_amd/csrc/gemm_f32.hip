// Exact-fp32 GEMM on the f32-input MFMA (v_mfma_f32_16x16x4_f32: bitwise a k-ordered fmaf chain)
// for the small, accuracy-critical "head" of the model: pooled-token projections
// (`x @ visual.proj`, `x @ text_projection`), the cosine-similarity logits
// `logit_scale.exp() * I @ T.t()` of CLIP.forward (/root/reference/CLIP/train.py:161,
// /root/reference/CLIP/predict.py:46 - where bit-exact argmax against the CPU oracle is decided)
// and their backward products.  ~0.05 % of the step's FLOPs, so generic strides beat tuning:
//
//   C[m*ldc + n] = alpha * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + beta * C[m*ldc + n]
//
// 32x32x32 tile, 4 waves (2x2), each wave one 16x16 MFMA tile; operands staged through LDS
// (+1-padded rows).  The MFMA is issued swapped (B as the A operand) so each lane ends up with
// 4 consecutive n of one m -> 16-byte stores.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

#define FWT 1                 // MFMA tiles per wave and dimension: a 32x32 block tile -> 4x the workgroups of a 64x64 one
#define FBM (32 * FWT)        // (these GEMMs are 1 GFLOP on a 256-CU chip: latency-bound, so occupancy beats operand reuse)
#define FBN (32 * FWT)
#define FBK 32
#define FLD (FBK + 1)

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, long sam, long sak,
                                                       const float* __restrict__ B, long sbn, long sbk, int M, int N,
                                                       int K, float alpha, const float* __restrict__ alpha_log_dev, float beta,
                                                       float* __restrict__ C, long ldc) {
  __shared__ float As[FBM * FLD], Bs[FBN * FLD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int bm0 = blockIdx.y * FBM, bn0 = blockIdx.x * FBN;
  const int wm0 = (wave >> 1) * 16 * FWT, wn0 = (wave & 1) * 16 * FWT;
  f32x4 acc[FWT][FWT];
#pragma unroll
  for (int i = 0; i < FWT; ++i)
#pragma unroll
    for (int j = 0; j < FWT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool a_kfast = sak == 1, b_kfast = sbk == 1;
  // register-staged prefetch: the global loads of K-tile k0+FBK are in flight while tile k0 is multiplied
  constexpr int NL = FBM * FBK / 256;                        // elements per thread per operand tile
  float ra[NL], rb[NL];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { m = idx / FBK; k = idx % FBK; } else { k = idx / FBM; m = idx % FBM; }
      ra[i] = (bm0 + m < M && k0 + k < K) ? A[(long)(bm0 + m) * sam + (long)(k0 + k) * sak] : 0.f;
      int n, kb;
      if (b_kfast) { n = idx / FBK; kb = idx % FBK; } else { kb = idx / FBN; n = idx % FBN; }
      rb[i] = (bn0 + n < N && k0 + kb < K) ? B[(long)(bn0 + n) * sbn + (long)(k0 + kb) * sbk] : 0.f;
    }
  };
  gload(0);
  for (int k0 = 0; k0 < K; k0 += FBK) {
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int idx = tid + 256 * i;
      int m, k;
      if (a_kfast) { m = idx / FBK; k = idx % FBK; } else { k = idx / FBM; m = idx % FBM; }
      As[m * FLD + k] = ra[i];
      int n, kb;
      if (b_kfast) { n = idx / FBK; kb = idx % FBK; } else { kb = idx / FBN; n = idx % FBN; }
      Bs[n * FLD + kb] = rb[i];
    }
    __syncthreads();
    if (k0 + FBK < K) gload(k0 + FBK);
#pragma unroll
    for (int kk = 0; kk < FBK / 4; ++kk) {
      float af[FWT], bf[FWT];
#pragma unroll
      for (int t = 0; t < FWT; ++t) {
        af[t] = As[(wm0 + 16 * t + li) * FLD + 4 * kk + g];
        bf[t] = Bs[(wn0 + 16 * t + li) * FLD + 4 * kk + g];
      }
#pragma unroll
      for (int mt = 0; mt < FWT; ++mt)
#pragma unroll
        for (int nt = 0; nt < FWT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[nt], af[mt], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
  }
  if (alpha_log_dev) alpha *= __expf(*alpha_log_dev);
  // D[i = 4g + r][j = li]: i <-> n (from the B-side operand), j <-> m
  // beta != 0 (gradient accumulation): all 16 previous values are requested first, clamped in-bounds so that no load is
  // predicated - a per-element load -> fma -> store chain is 16 serial memory round trips
  float prev[FWT][FWT][4];
#pragma unroll
  for (int mt = 0; mt < FWT; ++mt)
#pragma unroll
    for (int nt = 0; nt < FWT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        prev[mt][nt][r] = 0.f;
        if (beta != 0.f) {
          const int m = min(bm0 + wm0 + 16 * mt + li, M - 1), n = min(bn0 + wn0 + 16 * nt + 4 * g + r, N - 1);
          prev[mt][nt][r] = C[(long)m * ldc + n];
        }
      }
#pragma unroll
  for (int mt = 0; mt < FWT; ++mt) {
    const int m = bm0 + wm0 + 16 * mt + li;
    if (m >= M) continue;
#pragma unroll
    for (int nt = 0; nt < FWT; ++nt) {
      const int n0 = bn0 + wn0 + 16 * nt + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + r;
        if (n < N) C[(long)m * ldc + n] = alpha * acc[mt][nt][r] + beta * prev[mt][nt][r];
      }
    }
  }
}

extern "C" int cclip_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                              int32_t M, int32_t N, int32_t K, float alpha, const float* alpha_log_dev, float beta, float* C,
                              int64_t ldc, hipStream_t stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0) return CCLIP_ERR_ARG;
  dim3 grid((N + FBN - 1) / FBN, (M + FBM - 1) / FBM), block(256);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, block, 0, stream, A, (long)sam, (long)sak, B, (long)sbn, (long)sbk, M, N, K,
                     alpha, alpha_log_dev, beta, C, (long)ldc);
  return cclip_launch_status();
}
