// fp8 (OCP e4m3) projections for inference: per-row dynamic quantisation + a block-scaled-MFMA GEMM.
//
// BASELINE.json configs[4] asks for an fp8 MFMA path on ViT-L/14@336px encode_image.  On gfx950 the plain fp8 MFMAs
// (v_mfma_f32_16x16x32_fp8_fp8) run at the bf16 rate; only the block-scaled form v_mfma_scale_f32_16x16x128_f8f6f4 doubles
// the FLOPs per clock (MI355X_MICROARCH.md).  It is used here with all block scales = 1.0 (E8M0 0x7F): the scaling is
// per ROW of either operand (per token for activations, per output channel for weights, amax / 448), applied to the
// fp32 accumulator in the epilogue, so the kernel is an ordinary GEMM over e4m3 operands:
//     C[m][n] = act( sa[m] * sb[n] * sum_k A8[m][k] * B8[n][k] + bias[n] )        A8: [M, K], B8: [N, K], both K-contiguous.
// Operand lane map (checked with exact integer data, tools/micro/fp8_mfma_layout.hip): lane l supplies 32 k-values of row
// l & 15; WHICH 32 is free as long as both operands agree (the product is a sum over k) - both take 16-byte chunks g and
// 4+g of their 128-byte LDS row (g = l >> 4), the hardware's own order, which the block scales of the MX kernels require.  The first operand indexes the accumulator rows (4g + reg), the second the
// columns (l & 15), exactly as the 16-bit MFMAs, so tile staging (global_load_lds into the XOR-swizzled [row][128 B]
// image, n-permutation of the weight rows) and the 8-column-run epilogue are those of gemm_bf16_impl.h: one K-tile is
// 128 bytes = 128 fp8 values instead of 64 bf16 values.
//
// There is no reference behaviour for fp8: parity of this path is "unpinned"; the tests bound it
// against the fp32 oracle with an fp8-sized tolerance and check the GEMM itself exactly against the same quantised operands.
#include <cstdlib>
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

// ---- MX (block-scaled) quantisation: one E8M0 power-of-two scale per 32 consecutive k of a row -------------------------
// Round 2: the A operands of the out-proj / c_proj GEMMs are produced by the attention / by the fc GEMM's epilogue, where no
// workgroup sees a whole row - a per-row amax would need a second pass over the row.  The block-scaled MFMA takes one E8M0
// exponent per (row, 32-deep k block) of either operand and applies it in hardware, so those operands are quantised in
// 32-element blocks, each from its own amax: x ~= 2^(e - 127) * fp8, e = the smallest exponent with amax / 2^(e-127) <= 448.
__device__ __forceinline__ int e8m0_of(float amax) {
  const unsigned b = __float_as_uint(amax * (1.0f / 448.0f));
  int e = (int)(b >> 23) + ((b & 0x7FFFFFu) ? 1 : 0);
  e = e < 1 ? 1 : e;               // an all-zero block: any scale (its values are zeros)
  return e > 253 ? 253 : e;
}
__device__ __forceinline__ float e8m0_inv(int e) { return __uint_as_float((unsigned)(254 - e) << 23); }   // 2^(127 - e)
__device__ __forceinline__ int2 pack8_fp8(const float (&v)[8], float inv) {
  float f[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = fminf(fmaxf(v[j] * inv, -448.f), 448.f);
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
  return make_int2(w0, w1);
}

// x16 [rows, cols] (cols % 32 == 0) -> e4m3 [rows, cols] + E8M0 block scales; a lane owns 8 elements, a quad one block.
// Scale layout (everywhere in this file): K-tile major, [cols / 128][rows][4] bytes - the scale of (row r, block b) sits at
// (b >> 2) * ld + 4 r + (b & 3).  The consuming GEMM's lane (row i, k block g) of a 16-row m-tile then reads byte 4 i + g of
// a 64-byte run per K-tile: one cache line per wave-instruction.  (First form, row major [rows][cols / 32]: 16 lines per
// instruction, and the K = 4096 projection ran 44 % slower than with row scales - the loads queued in the address unit.)
__global__ __launch_bounds__(256) void quantize_mx_fp8_kernel(const bf16* __restrict__ x, long ldx, int rows, int cols,
                                                              unsigned char* __restrict__ out, long ldo,
                                                              unsigned char* __restrict__ mx, long ldmx) {
  const int cpr = cols >> 3;                                   // 16-byte chunks per row
  const long total = (long)rows * cpr;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < ((total + 3) & ~3L); i += (long)gridDim.x * 256) {
    const bool live = i < total;                               // (total is a multiple of 4: a quad is live or dead as a whole)
    const long r = live ? i / cpr : 0;
    const int c = live ? (int)(i - r * cpr) * 8 : 0;
    const bf16x8 v = *(const bf16x8*)(x + r * ldx + c);
    float f[8], amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { f[j] = (float)v[j]; amax = fmaxf(amax, fabsf(f[j])); }
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    const int e = e8m0_of(amax);
    if (live) {
      *(int2*)(out + r * ldo + c) = pack8_fp8(f, e8m0_inv(e));
      if ((threadIdx.x & 3) == 0) mx[(long)(c >> 7) * ldmx + r * 4 + ((c >> 5) & 3)] = (unsigned char)e;
    }
  }
}

struct Fp8Args {
  const unsigned char* A; const unsigned char* B; long lda, ldb;      // bytes = elements
  const float* sa; const float* sb;                                   // per-row scales (sa unused with block scales on A)
  const unsigned char* mxa; long ldmx;                                // E8M0 block scales of A, [K / 128][M][4] (MXA kernels)
  int M, N, K;
  const float* bias;
  bf16* out; long ldc;                                                // OUT 0: 16-bit
  unsigned char* out8; long ldo8; unsigned char* out_mx; long ldomx;  // OUT 1: e4m3 + E8M0 per 32 output columns
  float* out_f32; const float* residual; long ldf;                    // OUT 2: fp32, + residual (same leading dimension)
  int group_n;                                                        // tile order: column groups of this many tiles (tile_coords); 0 = row-major
  int dbg;                                                            // timing ablation (CCLIP_FP8_DBG bit 0: no epilogue); never set by the product path
};

// The hardware's own k order (tools/micro/fp8_mfma_scale_probe2.hip): registers 0..3 of lane group g hold k = 16g..16g+15 and
// registers 4..7 hold k = 64+16g..64+16g+15, and the scale byte of lane group b acts on k = 32b..32b+31.  With unit scales
// any order both operands share gives the same sum (round 1 read chunks 2g, 2g+1); block scales need this one.
#define FP8_CHUNK_LO(g) (g)
#define FP8_CHUNK_HI(g) (4 + (g))
__device__ __forceinline__ v8i frag_rows_fp8(const char* tile, int row0, int lane) {
  const int row = row0 + (lane & 15), g = lane >> 4;
  const int4 lo = *(const int4*)(tile + row * 128 + ((FP8_CHUNK_LO(g) ^ (row & 7)) << 4));
  const int4 hi = *(const int4*)(tile + row * 128 + ((FP8_CHUNK_HI(g) ^ (row & 7)) << 4));
  return (v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
}

// One of the 8 LDS-DMA instructions a wave issues per K-tile (4 of the A tile, 4 of the B tile; stage_tile's addressing for
// K-contiguous operands), through inline asm: for an LDS-DMA the compiler knows about, its waitcnt pass puts s_waitcnt
// vmcnt(0) in front of the next LDS read of the iteration (measured here: the DMA group was drained in the middle of every
// K-tile).  The only waits this kernel needs are the explicit vmcnt(0) + barrier at the top of each K-tile.
template <int PERM>
__device__ __forceinline__ void dma_piece_fp8(const bf16* __restrict__ G, long ld, int R, int Kend, int r0, int k0, char* lds_tile,
                                              int idx, int lane) {
  const int sub = idx >> 4, rb = idx & 15;
  const int rp = rb * 8 + (lane >> 3);
  const int c = (lane & 7) ^ (rp & 7);
  int r = rp;
  if (PERM) r = (rp & 64) + nperm((rp >> 4) & 3, rp & 15);
  int gr = r0 + sub * 128 + r; gr = gr < R ? gr : R - 1;
  const int gk = k0 + c * 8;
  const bf16* src = G + (long)gr * ld + gk;
  if (gk >= Kend) src = (const bf16*)g_zero16;
  const unsigned off = (unsigned)(size_t)LDS_PTR(lds_tile + sub * TILE_BYTES + rb * 1024);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(off), "v"(src) : "memory", "m0");
}

// 256x256 tile, 8 waves (2x4) of 128x64, 2 LDS stages of 64 KiB, one barrier per 128-deep K-tile.
// MXA: the A operand carries E8M0 block scales (one byte per row and 32 k): lane (i, g) supplies rows i of every m-tile and
// the k block g of the K-tile, so it loads ONE byte per m-tile and K-tile (a tile ahead, in registers) and hands it to the
// MFMA as the scale of its own 32 values.  OUT: 0 = 16-bit, 1 = e4m3 + block scales (the next GEMM's MXA operand), 2 = fp32
// residual stream (out = residual + result).
template <int ACT, int MXA, int OUT>
__global__ __launch_bounds__(512, 2) void gemm_fp8_kernel(const Fp8Args p) {
  constexpr int WN = 4, MT = 8, NW = 8, BM_ = 256, BN_ = 256, STAGES = 2, NSA = 2, NSB = 2;
  constexpr int STAGE_BYTES_ = (NSA + NSB) * TILE_BYTES, KB = 128;
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN_ - 1) / BN_;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tm_i, tn_i;
  tile_coords(bid, (p.M + BM_ - 1) / BM_, tiles_n, p.group_n, tm_i, tn_i);
  const int bm0 = tm_i * BM_, bn0 = tn_i * BN_;
  const int nkt = (p.K + KB - 1) / KB;
  const int wm = wave / WN, wn = wave % WN;
  const int wm0 = wm * 16 * MT, wn0 = wn * 64;
  const int a_off = (wm0 >> 7) * TILE_BYTES, a_row = wm0 & 127;
  const int b_off = (wn >> 1) * TILE_BYTES, b_row = (wn & 1) * 64;
  const int li = lane & 15, g = lane >> 4;
  // the 16-bit staging routine moves 16-byte chunks of 128-byte rows: address the fp8 matrices in 2-byte units
  const bf16* A2 = (const bf16*)p.A; const bf16* B2 = (const bf16*)p.B;
  const long lda2 = p.lda >> 1, ldb2 = p.ldb >> 1;
  const int K2 = p.K >> 1;
  // piece i (0..7) of this wave's share of K-tile kt: 4 instructions of the A tile, then 4 of the B tile
  auto piece = [&](int kt, int i) {
    char* sb = smem + (kt & 1) * STAGE_BYTES_;
    if (i < 4) dma_piece_fp8<0>(A2, lda2, p.M, K2, bm0, kt * (KB / 2), sb, wave + NW * i, lane);
    else dma_piece_fp8<1>(B2, ldb2, p.N, K2, bn0, kt * (KB / 2), sb + NSA * TILE_BYTES, wave + NW * (i - 4), lane);
  };
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // block scales of A: byte offset of (row of m-tile mt, block g) in K-tile 0; rows past the edge read the last row (never stored)
  unsigned moff[MXA ? MT : 1];
  int scn[MXA ? MT : 1];
  if (MXA) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      int m = bm0 + wm0 + 16 * mt + li; m = m < p.M ? m : p.M - 1;
      moff[mt] = (unsigned)(4 * m + g);
      scn[mt] = p.mxa[moff[mt]];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) piece(0, i);
  // One 128-deep K-tile = 32 MFMAs (32 cycles each) per wave against 24 ds_read_b128 and 8 LDS-DMA instructions.  Round 2:
  // the DMA group of tile kt+1 used to be issued back to back right after the barrier (~500 clocks of VMEM issue with the
  // matrix pipe idle, the round-1 finding on the 16-bit kernels) and every m-tile's A fragment was read just before its
  // MFMAs.  Now the order is written out: B fragments + A fragment 0 up front; then per m-tile its four MFMAs with the two
  // reads of the next A fragment and ONE DMA instruction of tile kt+1 between them (sched_barrier pins the order; the
  // compiler still places the lgkmcnt waits).
  auto k_tile = [&](int kt, auto dma_tag) {
    constexpr bool DMA = decltype(dma_tag)::value;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    int sc[MXA ? MT : 1];
    if (MXA) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) sc[mt] = scn[mt];          // loaded a tile ago; the wait above covers them
      if (DMA) {
        const unsigned char* mxk = p.mxa + (long)(kt + 1) * p.ldmx;       // (wave-uniform base + 32-bit lane offset)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) scn[mt] = mxk[moff[mt]];
      }
    }
    const char* At = smem + (kt & 1) * STAGE_BYTES_ + a_off;
    const char* Bt = smem + (kt & 1) * STAGE_BYTES_ + NSA * TILE_BYTES + b_off;
    v8i wf[4];
    v8i xc = frag_rows_fp8(At, a_row, lane);                     // A fragment 0 and B fragment 0 first: the first MFMA waits for 4 reads, not 10
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = frag_rows_fp8(Bt, b_row + 16 * nt, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int scv = MXA ? sc[MXA ? mt : 0] : 0x7F7F7F7F;
      const int row = a_row + 16 * (mt + 1) + li;
      int4 lo, hi;
      acc[mt][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[0], xc, acc[mt][0], 0, 0, 0, 0x7F7F7F7F, 0, scv);
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 1 < MT) lo = *(const int4*)(At + row * 128 + ((FP8_CHUNK_LO(g) ^ (row & 7)) << 4));
      __builtin_amdgcn_sched_barrier(0);
      acc[mt][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[1], xc, acc[mt][1], 0, 0, 0, 0x7F7F7F7F, 0, scv);
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 1 < MT) hi = *(const int4*)(At + row * 128 + ((FP8_CHUNK_HI(g) ^ (row & 7)) << 4));
      __builtin_amdgcn_sched_barrier(0);
      acc[mt][2] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[2], xc, acc[mt][2], 0, 0, 0, 0x7F7F7F7F, 0, scv);
      __builtin_amdgcn_sched_barrier(0);
      if (DMA) piece(kt + 1, mt);
      __builtin_amdgcn_sched_barrier(0);
      acc[mt][3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[3], xc, acc[mt][3], 0, 0, 0, 0x7F7F7F7F, 0, scv);
      __builtin_amdgcn_sched_barrier(0);
      if (mt + 1 < MT) xc = (v8i){lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    }
  };
  int kt = 0;
  for (; kt + 1 < nkt; ++kt) k_tile(kt, std::true_type{});
  if (kt < nkt) k_tile(kt, std::false_type{});
  // ---- epilogue: lane holds, per (mt, h), columns n0..n0+7 of row m (as in gemm_bf16_impl.h) ----
  if (p.dbg & 1) { if (acc[0][0][0] == 12345.678f) p.out_f32[0] = 1.f; return; }
  __syncthreads();                                     // every wave is done with the operand stages (the staged epilogues reuse them)
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
  // Interior wave tiles (128 rows x 64 columns), 16-bit and fp32-residual outputs: STAGED through LDS (round 3, the 16-bit
  // kernels' lesson: an accumulator puts rows on consecutive lanes, so a 16-byte store of it is 64 address pieces per
  // wave-instruction and the epilogue was 44-49 % of a K = 1024 launch - 3.3 TB/s of stores).  Each 16-row m-tile goes into a
  // wave-private XOR-swizzled patch in accumulator order (the operand stages are free after the K loop: 16 KiB per wave, two
  // patch sets) and comes back row-major: 8 rows x 128 contiguous bytes (16-bit) or 4 rows x 256 bytes (fp32) per instruction;
  // the fp32 form reads its residual rows in that same order, one m-tile ahead.  Same arithmetic per element.
  if ((OUT == 0 || OUT == 2) && bm0 + wm0 + 16 * MT <= p.M && bn0 + wn0 + 64 <= p.N && !(p.N & 7)) {
    char* patch = smem + wave * 16384;                 // (the work-group barrier above is outside this per-wave condition)
    auto wave_sync = [&]() {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    // the row scales of all eight m-tiles up front: a load inside the loop is waited for with vmcnt(0), and on this chip that
    // also waits for every store issued before it - each m-tile's stores then complete before the next m-tile starts
    float samv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) samv[mt] = MXA ? 1.0f : p.sa[bm0 + wm0 + 16 * mt + li];
    auto values = [&](int mt, int h, float (&v)[8]) {
      const float sam = samv[mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r] * sam * sbv[h][r] + bsv[h][r], 0.f);
    };
    if (OUT == 0) {
      const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        char* buf = patch + (mt & 1) * 2048;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
          values(mt, h, v);
          bf16x8 t;
#pragma unroll
          for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
          *(bf16x8*)(buf + li * 128 + (((4 * h + g) ^ (li & 7)) << 4)) = t;
        }
        wave_sync();
        bf16* o = p.out + (long)(bm0 + wm0 + 16 * mt) * p.ldc + bn0 + wn0 + 8 * c8;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int row = 8 * ps + r8;
          const bf16x8 t = *(const bf16x8*)(buf + row * 128 + ((c8 ^ (row & 7)) << 4));
          *(bf16x8*)(o + (long)row * p.ldc) = t;
        }
      }
    } else {
      const int r4 = lane >> 4, c16 = lane & 15;
      const long base = (long)(bm0 + wm0) * p.ldf + bn0 + wn0 + 4 * c16;
      float4 rr[2][4];
      auto load_res = [&](int mt, float4 (&q)[4]) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) q[ps] = *(const float4*)(p.residual + base + (long)(16 * mt + 4 * ps + r4) * p.ldf);
      };
      load_res(0, rr[0]);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        char* buf = patch + (mt & 1) * 4096;
        if (mt + 1 < MT) load_res(mt + 1, rr[(mt + 1) & 1]);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
          values(mt, h, v);
          const int ch = 8 * h + 2 * g;                  // 16-byte chunk (4 floats) of the 256-byte row
          *(float4*)(buf + li * 256 + (((ch) ^ li) << 4)) = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)(buf + li * 256 + (((ch + 1) ^ li) << 4)) = make_float4(v[4], v[5], v[6], v[7]);
        }
        wave_sync();
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int row = 4 * ps + r4;
          const float4 t = *(const float4*)(buf + row * 256 + ((c16 ^ row) << 4));
          const float4 q = rr[mt & 1][ps];
          *(float4*)(p.out_f32 + base + (long)(16 * mt + row) * p.ldf) = make_float4(t.x + q.x, t.y + q.y, t.z + q.z, t.w + q.w);
        }
      }
    }
    return;
  }
  if (OUT == 1 && bm0 + wm0 + 16 * MT <= p.M && bn0 + wn0 + 64 <= p.N && !(p.ldo8 & 15) && !((size_t)p.out8 & 15)) {
    // e4m3 + block-scale output, interior wave tile: the 8-byte packs of an m-tile (16 rows x 64 bytes) through a 1 KiB patch,
    // stored as 16 bytes per lane = one instruction of 16 rows x 64 contiguous bytes (was two of 16 rows x 32 bytes, 8 per lane)
    char* patch = smem + wave * 16384;
    const int r16 = lane >> 2, c4 = lane & 3;
    float samv[MT];                                    // (up front: see the 16-bit path)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) samv[mt] = MXA ? 1.0f : p.sa[bm0 + wm0 + 16 * mt + li];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      char* buf = patch + (mt & 1) * 1024;
      const int m = bm0 + wm0 + 16 * mt + li;
      const float sam = samv[mt];
      int eb[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r] * sam * sbv[h][r] + bsv[h][r], 0.f);
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(v[r]));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        eb[h] = e8m0_of(amax);
        *(int2*)(buf + li * 64 + 32 * h + 8 * g) = pack8_fp8(v, e8m0_inv(eb[h]));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const int4 t = *(const int4*)(buf + r16 * 64 + 16 * c4);
      *(int4*)(p.out8 + (long)(bm0 + wm0 + 16 * mt + r16) * p.ldo8 + bn0 + wn0 + 16 * c4) = t;
      if (g == 0)
        *(unsigned short*)(p.out_mx + (long)((bn0 + wn0) >> 7) * p.ldomx + 4 * m + (((bn0 + wn0) >> 5) & 3)) = (unsigned short)(eb[0] | (eb[1] << 8));
    }
    return;
  }
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = bm0 + wm0 + 16 * mt + li;
    const bool mlive = m < p.M;
    if (OUT != 1 && !mlive) continue;                  // (OUT 1 reduces across lanes: every lane stays in)
    const float sam = MXA ? 1.0f : p.sa[mlive ? m : p.M - 1];
    int eb[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n0 = bn0 + wn0 + 32 * h + 8 * g;
      if (OUT != 1 && n0 >= p.N) continue;
      float v[8];
#pragma unroll
      for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r] * sam * sbv[h][r] + bsv[h][r], 0.f);
      if (OUT == 0) {
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
      } else if (OUT == 1) {
        // the 32 columns bn0 + wn0 + 32h .. +31 of row m sit in the four lanes (li, g = 0..3): one block of the next GEMM's A
        float amax = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) amax = fmaxf(amax, fabsf(v[r]));
        amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
        amax = fmaxf(amax, __shfl_xor(amax, 32, 64));
        eb[h] = e8m0_of(amax);
        if (mlive && bn0 + wn0 < p.N) *(int2*)(p.out8 + (long)m * p.ldo8 + n0) = pack8_fp8(v, e8m0_inv(eb[h]));   // N % 64 == 0 (launcher): a wave's 64 columns are in or out as a whole
      } else {
        const float* rp = p.residual + (long)m * p.ldf + n0;
        float* o = p.out_f32 + (long)m * p.ldf + n0;
        if (n0 + 8 <= p.N) {
          const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
          *(float4*)o = make_float4(v[0] + r0.x, v[1] + r0.y, v[2] + r0.z, v[3] + r0.w);
          *(float4*)(o + 4) = make_float4(v[4] + r1.x, v[5] + r1.y, v[6] + r1.z, v[7] + r1.w);
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) if (n0 + r < p.N) o[r] = v[r] + rp[r];
        }
      }
    }
    if (OUT == 1 && mlive && g == 0 && bn0 + wn0 < p.N)
      *(unsigned short*)(p.out_mx + (long)((bn0 + wn0) >> 7) * p.ldomx + 4 * m + (((bn0 + wn0) >> 5) & 3)) = (unsigned short)(eb[0] | (eb[1] << 8));
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

extern "C" int CCLIP_FN(cclip_quantize_mx_fp8)(const void* x16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                                              void* block_scale, int64_t ldmx, hipStream_t stream) {
  if (!x16 || !out_fp8 || !block_scale || rows <= 0 || cols <= 0 || (cols & 31) || (ldx & 7) || (ldo & 7) || ldmx < 4 * (int64_t)rows) return CCLIP_ERR_ARG;
  if (((uintptr_t)x16 & 15) || ((uintptr_t)out_fp8 & 7)) return CCLIP_ERR_ARG;
  const long units = (long)rows * (cols / 8);
  long grid = (units + 255) / 256; if (grid > 16384) grid = 16384;
  hipLaunchKernelGGL(quantize_mx_fp8_kernel, dim3((unsigned)grid), dim3(256), 0, stream, (const bf16*)x16, (long)ldx, rows, cols,
                     (unsigned char*)out_fp8, (long)ldo, (unsigned char*)block_scale, (long)ldmx);
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

// Descriptor form (include/cclip_hip.h: cclip_fp8_gemm_desc).  Exactly one output form; block_scale_a != NULL selects the
// block-scaled A operand (then scale_a is not read).
extern "C" int CCLIP_FN(cclip_gemm_fp8_ex)(const cclip_fp8_gemm_desc* d, hipStream_t stream) {
  if (!d || !d->A || !d->B || !d->scale_b || d->M <= 0 || d->N <= 0 || d->K <= 0) return CCLIP_ERR_ARG;
  if (!d->block_scale_a && !d->scale_a) return CCLIP_ERR_ARG;
  if ((d->K & 15) || (d->lda & 15) || (d->ldb & 15)) return CCLIP_ERR_ARG;
  if (((uintptr_t)d->A | (uintptr_t)d->B) & 15) return CCLIP_ERR_ARG;
  const int nout = (d->out16 != nullptr) + (d->out_fp8 != nullptr) + (d->out_f32 != nullptr);
  if (nout != 1) return CCLIP_ERR_ARG;
  const int mxa = d->block_scale_a != nullptr;
  if (mxa && ((d->K & 127) || d->ld_block_scale_a < 4 * (int64_t)d->M)) return CCLIP_ERR_ARG;   // whole K-tiles of 4 blocks
  int out = 0;
  if (d->out16) { if ((d->ldc & 7) || ((uintptr_t)d->out16 & 15)) return CCLIP_ERR_ARG; }
  if (d->out_fp8) {
    out = 1;
    if (!d->out_block_scale || (d->N & 63) || (d->ld_out_fp8 & 7) || ((uintptr_t)d->out_fp8 & 7) || (d->ld_out_block_scale & 1) ||
        ((uintptr_t)d->out_block_scale & 1) || d->ld_out_block_scale < 4 * (int64_t)d->M) return CCLIP_ERR_ARG;
  }
  if (d->out_f32) {
    out = 2;
    if (!d->residual || (d->ldf & 3) || (((uintptr_t)d->out_f32 | (uintptr_t)d->residual) & 15)) return CCLIP_ERR_ARG;
  }
  Fp8Args a;
  a.A = (const unsigned char*)d->A; a.B = (const unsigned char*)d->B; a.lda = d->lda; a.ldb = d->ldb;
  a.sa = d->scale_a; a.sb = d->scale_b; a.mxa = (const unsigned char*)d->block_scale_a; a.ldmx = d->ld_block_scale_a;
  a.M = d->M; a.N = d->N; a.K = d->K; a.bias = d->bias;
  a.out = (bf16*)d->out16; a.ldc = d->ldc;
  a.out8 = (unsigned char*)d->out_fp8; a.ldo8 = d->ld_out_fp8; a.out_mx = (unsigned char*)d->out_block_scale; a.ldomx = d->ld_out_block_scale;
  a.out_f32 = d->out_f32; a.residual = d->residual; a.ldf = d->ldf;
  { static const int dbg = getenv("CCLIP_FP8_DBG") ? atoi(getenv("CCLIP_FP8_DBG")) : 0; a.dbg = dbg; }
  {
    // tile order (gemm_bf16_impl.h tile_coords): which weight panels one XCD's L2 holds together.  Measured per column-tile count
    // on the ViT-L/14@336px shapes (profiles/r03_tile_order_ab.txt): 4 column tiles (N = 1024) -11 % / -4 % in groups of 3,
    // 16 column tiles (N = 4096) -2 % in groups of 8, 12 column tiles nothing.  CCLIP_FP8_GROUP_N overrides (0 = row-major).
    static const int gn_env = getenv("CCLIP_FP8_GROUP_N") ? atoi(getenv("CCLIP_FP8_GROUP_N")) : -1;
    const int tn = (d->N + 255) / 256;
    a.group_n = gn_env >= 0 ? gn_env : (tn == 4 ? 3 : tn >= 16 ? 8 : 0);
  }
  const int tiles = ((d->M + 255) / 256) * ((d->N + 255) / 256);
  dim3 grid(tiles), block(512);
#define FP8L(ACTV, MXAV, OUTV) hipLaunchKernelGGL((gemm_fp8_kernel<ACTV, MXAV, OUTV>), grid, block, 0, stream, a)
  const int key = (d->act == CCLIP_ACT_QUICKGELU ? 100 : d->act == CCLIP_ACT_NONE ? 0 : -1000) + 10 * mxa + out;
  switch (key) {
    case 0: FP8L(CCLIP_ACT_NONE, 0, 0); break;            // qkv
    case 1: FP8L(CCLIP_ACT_NONE, 0, 1); break;
    case 100: FP8L(CCLIP_ACT_QUICKGELU, 0, 0); break;     // fc, 16-bit hidden
    case 101: FP8L(CCLIP_ACT_QUICKGELU, 0, 1); break;     // fc, hidden straight to block-scaled e4m3
    case 10: FP8L(CCLIP_ACT_NONE, 1, 0); break;
    case 12: FP8L(CCLIP_ACT_NONE, 1, 2); break;           // out-proj / c_proj on the fp32 residual stream
    default: return CCLIP_ERR_ARG;
  }
#undef FP8L
  return cclip_launch_status();
}

extern "C" int CCLIP_FN(cclip_gemm_fp8)(const void* A8, int64_t lda, const float* scale_a, const void* B8, int64_t ldb,
                                       const float* scale_b, int32_t M, int32_t N, int32_t K, const float* bias, int32_t act,
                                       void* out16, int64_t ldc, hipStream_t stream) {
  if (!scale_a || !out16) return CCLIP_ERR_ARG;
  cclip_fp8_gemm_desc d = {};
  d.A = A8; d.lda = lda; d.scale_a = scale_a; d.B = B8; d.ldb = ldb; d.scale_b = scale_b; d.M = M; d.N = N; d.K = K;
  d.bias = bias; d.act = act; d.out16 = out16; d.ldc = ldc;
  return CCLIP_FN(cclip_gemm_fp8_ex)(&d, stream);
}
