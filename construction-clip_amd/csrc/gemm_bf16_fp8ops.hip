// fp8 (OCP e4m3) projections for inference: per-row dynamic quantisation + a block-scaled-MFMA GEMM.
//
// BASELINE.json configs[4] asks for an fp8 MFMA path on ViT-L/14@336px encode_image.  On gfx950 the plain fp8 MFMAs
// (v_mfma_f32_16x16x32_fp8_fp8) run at the bf16 rate; only the block-scaled form v_mfma_scale_f32_16x16x128_f8f6f4 doubles
// the FLOPs per clock (MI355X_MICROARCH.md).  It is used here with all block scales = 1.0 (E8M0 0x7F): the scaling is
// per ROW of either operand (per token for activations, per output channel for weights, amax / 448), applied to the
// fp32 accumulator in the epilogue, so the kernel is an ordinary GEMM over e4m3 operands:
//     C[m][n] = act( sa[m] * sb[n] * sum_k A8[m][k] * B8[n][k] + bias[n] )        A8: [M, K], B8: [N, K], both K-contiguous.
// Operand lane map (checked with exact integer data, tools/micro/fp8_mfma_layout.hip): lane l supplies 32 k-values of row
// l & 15; WHICH 32 is free as long as both operands agree (the product is a sum over k) - both take 16-byte chunks 2g and
// 2g+1 of their 128-byte LDS row (g = l >> 4).  The first operand indexes the accumulator rows (4g + reg), the second the
// columns (l & 15), exactly as the 16-bit MFMAs, so tile staging (global_load_lds into the XOR-swizzled [row][128 B]
// image, n-permutation of the weight rows) and the 8-column-run epilogue are those of gemm_bf16_impl.h: one K-tile is
// 128 bytes = 128 fp8 values instead of 64 bf16 values.
//
// There is no reference behaviour for fp8: parity of this path is "unpinned"; the tests bound it
// against the fp32 oracle with an fp8-sized tolerance and check the GEMM itself exactly against the same quantised operands.
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

typedef int v8i __attribute__((ext_vector_type(8)));

// ---- per-row quantisation: x16 [rows, cols] -> e4m3 [rows, cols] + scale[rows] (scale = amax / 448; 1 for an all-zero row)
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const bf16* __restrict__ x, long ldx, int rows, int cols,
                                                                unsigned char* __restrict__ out, long ldo, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const bf16* xr = x + (long)r * ldx;
    float amax = 0.f;
    for (int c = lane * 8; c < cols; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)v[j]));
    }
    amax = wave_max(amax);
    const float s = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / s;
    for (int c = lane * 8; c < cols; c += 512) {
      const bf16x8 v = *(const bf16x8*)(xr + c);
      float f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf((float)v[j] * inv, -448.f), 448.f);
      int w0 = 0, w1 = 0;
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
      w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
      w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
      *(int2*)(out + (long)r * ldo + c) = make_int2(w0, w1);
    }
    if (lane == 0) scale[r] = s;
  }
}

// LayerNorm whose output goes straight to e4m3 + per-row scale: the quantisation of the LN-fed projections' A operand costs
// no extra pass (the row is in registers anyway).  Same statistics as ln_fwd_kernel (layernorm.hip).
template <int NV>
__global__ __launch_bounds__(256) void ln_fwd_fp8_kernel(const float* __restrict__ x, long ldx, int rows, int D,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                         unsigned char* __restrict__ out, long ldo, float* __restrict__ scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)D;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const float* xr = x + (long)r * ldx;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      v[c] = col < D ? *(const float4*)(xr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[c].x + v[c].y + v[c].z + v[c].w;
    }
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        const float a = v[c].x - mean, b = v[c].y - mean, cc = v[c].z - mean, d = v[c].w - mean;
        q += a * a + b * b + cc * cc + d * d;
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_d + eps);
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        const float4 g = *(const float4*)(gamma + col), b = *(const float4*)(beta + col);
        v[c].x = (v[c].x - mean) * rstd * g.x + b.x;
        v[c].y = (v[c].y - mean) * rstd * g.y + b.y;
        v[c].z = (v[c].z - mean) * rstd * g.z + b.z;
        v[c].w = (v[c].w - mean) * rstd * g.w + b.w;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[c].x), fabsf(v[c].y))), fmaxf(fabsf(v[c].z), fabsf(v[c].w)));
      }
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax * (1.0f / 448.0f) : 1.0f;
    const float inv = 1.0f / sc;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[c].x * inv, -448.f), 448.f), fminf(fmaxf(v[c].y * inv, -448.f), 448.f), w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[c].z * inv, -448.f), 448.f), fminf(fmaxf(v[c].w * inv, -448.f), 448.f), w, true);
        *(int*)(out + (long)r * ldo + col) = w;
      }
    }
    if (lane == 0) scale[r] = sc;
  }
}

struct Fp8Args {
  const unsigned char* A; const unsigned char* B; long lda, ldb;      // bytes = elements
  const float* sa; const float* sb;
  int M, N, K;
  const float* bias;
  bf16* out; long ldc;
};

__device__ __forceinline__ v8i frag_rows_fp8(const char* tile, int row0, int lane) {
  const int row = row0 + (lane & 15), g = lane >> 4;
  const int4 lo = *(const int4*)(tile + row * 128 + (((2 * g) ^ (row & 7)) << 4));
  const int4 hi = *(const int4*)(tile + row * 128 + (((2 * g + 1) ^ (row & 7)) << 4));
  return (v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}

// 256x256 tile, 8 waves (2x4) of 128x64, 2 LDS stages of 64 KiB, one barrier per 128-deep K-tile
template <int ACT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const Fp8Args p) {
  constexpr int WN = 4, MT = 8, NW = 8, BM_ = 256, BN_ = 256, STAGES = 2, NSA = 2, NSB = 2;
  constexpr int STAGE_BYTES_ = (NSA + NSB) * TILE_BYTES, KB = 128;
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN_ - 1) / BN_;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bm0 = (bid / tiles_n) * BM_, bn0 = (bid % tiles_n) * BN_;
  const int nkt = (p.K + KB - 1) / KB;
  const int wm = wave / WN, wn = wave % WN;
  const int wm0 = wm * 16 * MT, wn0 = wn * 64;
  const int a_off = (wm0 >> 7) * TILE_BYTES, a_row = wm0 & 127;
  const int b_off = (wn >> 1) * TILE_BYTES, b_row = (wn & 1) * 64;
  // the 16-bit staging routine moves 16-byte chunks of 128-byte rows: address the fp8 matrices in 2-byte units
  const bf16* A2 = (const bf16*)p.A; const bf16* B2 = (const bf16*)p.B;
  const long lda2 = p.lda >> 1, ldb2 = p.ldb >> 1;
  const int K2 = p.K >> 1;
  auto issue = [&](int kt) {
    char* sb = smem + (kt & 1) * STAGE_BYTES_;
    stage_tile<1, 0, NSA, NW>(A2, lda2, p.M, K2, bm0, kt * (KB / 2), sb, wave, lane);
    stage_tile<1, 1, NSB, NW>(B2, ldb2, p.N, K2, bn0, kt * (KB / 2), sb + NSA * TILE_BYTES, wave, lane);
  };
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  issue(0);
  // One 128-deep K-tile = 32 MFMAs (32 cycles each) per wave against 24 ds_read_b128 and 8 LDS-DMA instructions.  Round 2:
  // the DMA group of tile kt+1 used to be issued back to back right after the barrier (~500 clocks of VMEM issue with the
  // matrix pipe idle, the round-1 finding on the 16-bit kernels) and every m-tile's A fragment was read just before its
  // MFMAs.  Now: B fragments + the first A fragment up front, the other fragment reads and the DMA instructions ride one by
  // one between the MFMAs (sched_group_barrier pipeline; the steady state is one basic block).
  auto k_tile = [&](int kt, auto dma_tag) {
    constexpr bool DMA = decltype(dma_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const char* At = smem + (kt & 1) * STAGE_BYTES_ + a_off;
    const char* Bt = smem + (kt & 1) * STAGE_BYTES_ + NSA * TILE_BYTES + b_off;
    v8i wf[4], xf[MT];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = frag_rows_fp8(Bt, b_row + 16 * nt, lane);
#pragma unroll
    for (int mt = 0; mt < MT / 2; ++mt) xf[mt] = frag_rows_fp8(At, a_row + 16 * mt, lane);
    // program order: first-half reads, DMA, second-half reads - an LDS read may not sink below the DMA (an LDS write to the
    // compiler), so this is what lets the DMA group start after 6 MFMAs instead of after the last fragment read
    if (DMA) issue(kt + 1);
#pragma unroll
    for (int mt = MT / 2; mt < MT; ++mt) xf[mt] = frag_rows_fp8(At, a_row + 16 * mt, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[nt], xf[mt], acc[mt][nt], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);          // 4 B fragments + A fragment 0 (two b128 reads each)
#pragma unroll
    for (int i = 0; i < 6; ++i) {                                // A fragments 1..3 under MFMAs 0..5
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    if (DMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {                              // the 8 DMA instructions of tile kt+1 under MFMAs 6..13
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                // A fragments 4..7
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
    __builtin_amdgcn_sched_barrier(0);
  };
  int kt = 0;
  for (; kt + 1 < nkt; ++kt) k_tile(kt, std::true_type{});
  if (kt < nkt) k_tile(kt, std::false_type{});
  // ---- epilogue: lane holds, per (mt, h), columns n0..n0+7 of row m (as in gemm_bf16_impl.h) ----
  const int li = lane & 15, g = lane >> 4;
  float sbv[2][8], bsv[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n0 = bn0 + wn0 + 32 * h + 8 * g;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const bool in = n0 + r < p.N;
      sbv[h][r] = in ? p.sb[n0 + r] : 0.f;
      bsv[h][r] = in && p.bias ? p.bias[n0 + r] : 0.f;
    }
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = bm0 + wm0 + 16 * mt + li;
    if (m >= p.M) continue;
    const float sam = p.sa[m];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n0 = bn0 + wn0 + 32 * h + 8 * g;
      if (n0 >= p.N) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r] * sam * sbv[h][r] + bsv[h][r], 0.f);
      bf16* o = p.out + (long)m * p.ldc + n0;
      if (n0 + 8 <= p.N) {
        bf16x8 t;
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
        *(bf16x8*)o = t;
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) if (n0 + r < p.N) o[r] = (bf16)v[r];
      }
    }
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

extern "C" int CCLIP_FN(cclip_quantize_rows_fp8)(const void* x16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                                                float* scale, hipStream_t stream) {
  if (!x16 || !out_fp8 || !scale || rows <= 0 || cols <= 0 || (cols & 7) || (ldx & 7) || (ldo & 7)) return CCLIP_ERR_ARG;
  if (((uintptr_t)x16 & 15) || ((uintptr_t)out_fp8 & 7)) return CCLIP_ERR_ARG;
  int grid = (rows + 3) / 4; if (grid > 8192) grid = 8192;
  hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3(grid), dim3(256), 0, stream, (const bf16*)x16, (long)ldx, rows, cols,
                     (unsigned char*)out_fp8, (long)ldo, scale);
  return cclip_launch_status();
}

#ifndef CCLIP_F16
extern "C" int cclip_layernorm_fwd_fp8(const float* x, int64_t ldx, int32_t rows, int32_t D, const float* gamma, const float* beta,
                                       float eps, void* out_fp8, int64_t ldo, float* scale, hipStream_t stream) {
  if (!x || !gamma || !beta || !out_fp8 || !scale || rows <= 0 || D <= 0 || (D & 3) || D > 1024 || (ldx & 3) || (ldo & 3)) return CCLIP_ERR_ARG;
  int grid = (rows + 3) / 4; if (grid > 8192) grid = 8192;
  const int nv = (D + 255) / 256;
#define LNQ(NV) hipLaunchKernelGGL((ln_fwd_fp8_kernel<NV>), dim3(grid), dim3(256), 0, stream, x, (long)ldx, rows, D, gamma, beta, eps, (unsigned char*)out_fp8, (long)ldo, scale)
  switch (nv) { case 1: LNQ(1); break; case 2: LNQ(2); break; case 3: LNQ(3); break; default: LNQ(4); }
#undef LNQ
  return cclip_launch_status();
}
#endif

extern "C" int CCLIP_FN(cclip_gemm_fp8)(const void* A8, int64_t lda, const float* scale_a, const void* B8, int64_t ldb,
                                       const float* scale_b, int32_t M, int32_t N, int32_t K, const float* bias, int32_t act,
                                       void* out16, int64_t ldc, hipStream_t stream) {
  if (!A8 || !B8 || !scale_a || !scale_b || !out16 || M <= 0 || N <= 0 || K <= 0) return CCLIP_ERR_ARG;
  if ((K & 15) || (lda & 15) || (ldb & 15) || (ldc & 7)) return CCLIP_ERR_ARG;
  if (((uintptr_t)A8 | (uintptr_t)B8 | (uintptr_t)out16) & 15) return CCLIP_ERR_ARG;
  Fp8Args a;
  a.A = (const unsigned char*)A8; a.B = (const unsigned char*)B8; a.lda = lda; a.ldb = ldb; a.sa = scale_a; a.sb = scale_b;
  a.M = M; a.N = N; a.K = K; a.bias = bias; a.out = (bf16*)out16; a.ldc = ldc;
  const int tiles = ((M + 255) / 256) * ((N + 255) / 256);
  dim3 grid(tiles), block(512);
  switch (act) {
    case CCLIP_ACT_NONE: hipLaunchKernelGGL((gemm_fp8_kernel<CCLIP_ACT_NONE>), grid, block, 0, stream, a); break;
    case CCLIP_ACT_QUICKGELU: hipLaunchKernelGGL((gemm_fp8_kernel<CCLIP_ACT_QUICKGELU>), grid, block, 0, stream, a); break;
    default: return CCLIP_ERR_ARG;
  }
  return cclip_launch_status();
}
