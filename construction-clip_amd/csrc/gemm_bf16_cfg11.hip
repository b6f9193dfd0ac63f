// tile configuration 11: the WEIGHT-GRADIENT GEMM (both operands K-strided: dW = dY^T X, contraction over tokens) on the 4-wave
// 128x128-per-wave tile with a hand-scheduled inline-asm K loop (tools/gen_gemm_a4.py -> gemm_a4w_kloop.inc): 256 accumulators
// per wave in AGPRs, fragments by ds_read_b64_tr_b16 from the [64 k][128 col] sub-tile image, DMA and reads dealt into the MFMA
// gaps, one barrier per K-tile.  A weight gradient contracts over tens of thousands of tokens (hundreds of K-tiles per
// work-group even after split-K), so this launch IS its K loop: the 8-wave compiler-scheduled 256x128 kernel it replaces ran at
// 0.34 of the bf16 peak (MFMA-busy 0.40, round 2).
// Layout (0,0) only, K % 64 == 0, every split >= 2 K-tiles; the fused bias gradient (row sums of A) rides in the first column
// block of tiles as 8 extra MFMAs per k-step against an all-ones fragment.  Epilogue: the shared one (split-K slabs / residual).
#include "gemm_bf16_impl.h"

#ifdef CCLIP_F16
#define CCLIP_MFMA_ASM "v_mfma_f32_16x16x32_f16"
#else
#define CCLIP_MFMA_ASM "v_mfma_f32_16x16x32_bf16"
#endif

namespace CCLIP_NS {

typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
#include "gemm_a4w_kloop.inc"

__global__ __launch_bounds__(256, 1) void gemm_a4w_kernel(const GemmArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 32768];     // region(X, stage) = X * 32768 + stage * 16384; X = A0 A1 B0 B1
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + 255) / 256;
  // split_major (= tiles per split): ONE grid dimension of splits x tiles, the work list ordered split by split and cut into one
  // contiguous run per XCD (xcd_remap) - an XCD then works on one or two splits, i.e. one or two token ranges, and its L2 fetches
  // those ranges of both operands once for all the output tiles that contract over them.  (tiles, splits) as a 2-D grid deals
  // every split's tiles to all eight XCDs: each L2 then pulls every token range.
  const int item = xcd_remap(blockIdx.x, gridDim.x);
  const int bid = p.split_major ? item % p.split_major : item;
  const int zid = p.split_major ? item / p.split_major : (int)blockIdx.y;
  const int bm0 = (bid / tiles_n) * 256, bn0 = (bid % tiles_n) * 256;
  const int nkt = p.K / BK;
  const int kt0 = zid * p.ktiles_per_split;
  const int kt1 = (kt0 + p.ktiles_per_split < nkt) ? kt0 + p.ktiles_per_split : nkt;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 15, g = lane >> 4;

  // DMA pieces: piece idx = wave + 4 i of an operand tile: sub-tile idx >> 4, k-rows 4 (idx & 15) + (lane >> 4), chunk position
  // lane & 15 holding logical chunk (lane & 15) ^ fswz(k-row)  (stage_piece's K-strided map); columns past the edge are clamped
  unsigned oa[8], ob[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int idx = wave + 4 * i, sub = idx >> 4, rb = idx & 15;
    const int kr = rb * 4 + (lane >> 4);
    const int c = (lane & 15) ^ fswz(kr);
    const int padA = ((p.M + 7) & ~7) - 8, padB = ((p.N + 7) & ~7) - 8;
    int ga = bm0 + sub * 128 + c * 8; ga = ga <= padA ? ga : padA;
    int gb = bn0 + sub * 128 + c * 8; gb = gb <= padB ? gb : padB;
    oa[i] = (unsigned)(((long)kr * p.lda + (ga - bm0)) * 2);
    ob[i] = (unsigned)(((long)kr * p.ldb + (gb - bn0)) * 2);
  }
  const bf16* abase = p.A + (long)kt0 * BK * p.lda + bm0;
  const bf16* bbase = p.B + (long)kt0 * BK * p.ldb + bn0;
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem);
  const unsigned m0base = lds0 + wave * 1024;
  // transposing-read addresses (frag_cols' map): tile t, k-step ks, half h: k-row 32 ks + 8 g + q + 4 h, 4-column piece p
  u32x4w la[16];
  {
    const int q = (lane >> 2) & 3, pc = lane & 3;
    unsigned v[64];
#pragma unroll
    for (int op = 0; op < 2; ++op)
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int col = op ? 64 * (t >> 2) + 8 * pc + 32 * ((t & 3) >> 1) + 4 * (t & 1) : 16 * t + 4 * pc;
            const int kr = 32 * ks + 8 * g + q + 4 * h;
            const int chunk = col >> 3, sub = (col & 7) * 2;
            v[32 * op + 4 * t + 2 * ks + h] = lds0 + (2 * op + (op ? wn : wm)) * 32768 + kr * 256 + ((chunk ^ fswz(kr)) << 4) + sub;
          }
#pragma unroll
    for (int i = 0; i < 16; ++i) la[i] = (u32x4w){v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]};
  }
  const int niter = __builtin_amdgcn_readfirstlane((kt1 - kt0) - 2);
  const unsigned astride = (unsigned)(BK * p.lda * 2), bstride = (unsigned)(BK * p.ldb * 2);
  const int cs = __builtin_amdgcn_readfirstlane((p.colsum_dst && !p.colsum_b && bn0 == 0) ? 1 : 0);
  const u32x4w oa03 = {oa[0], oa[1], oa[2], oa[3]}, oa47 = {oa[4], oa[5], oa[6], oa[7]};
  const u32x4w ob03 = {ob[0], ob[1], ob[2], ob[3]}, ob47 = {ob[4], ob[5], ob[6], ob[7]};
  bf16x8 ones8;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones8[j] = (bf16)1.0f;
  f32x4 am[64], accb[8];
  GEMM_A4W_KLOOP(am, accb, abase, bbase, niter, m0base, astride, bstride, cs, oa03, oa47, ob03, ob47, la, ones8);

  // fused bias gradient: D[any row][col li] = sum_k A(16 mt + li, k): take row 0 (lanes g == 0) of the wn == 0 waves
  if (cs && wn == 0 && g == 0) {
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      const int m = bm0 + wm * 128 + 16 * mt + li;
      if (m < p.M) {
        if (p.split_ws) p.colsum_ws[(long)zid * p.M + m] = accb[mt][0];
        else p.colsum_dst[m] = (p.colsum_acc ? p.colsum_dst[m] : 0.f) + accb[mt][0];
      }
    }
  }
  const bool interior = bm0 + 256 <= p.M && bn0 + 256 <= p.N && !(p.N & 7) && !p.split_ws;
  float rres[2][2][8];
  bf16x8 raux[1][2];
  float bsv[2][8];
  {
    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mt][j] = am[8 * mt + j];
    gemm_epilogue<CCLIP_ACT_NONE, 8, false, 2, false>(p, acc, bm0 + wm * 128, bn0 + wn * 128, interior, li, g, rres, raux, bsv);
  }
  {
    f32x4 acc[8][4];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[mt][j] = am[8 * mt + 4 + j];
    gemm_epilogue<CCLIP_ACT_NONE, 8, false, 2, false>(p, acc, bm0 + wm * 128, bn0 + wn * 128 + 64, interior, li, g, rres, raux, bsv);
  }
}

bool cclip_gemm_launch_cfg11(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  if (lay != 0 || act != CCLIP_ACT_NONE || (a.K % BK) || a.colsum_b) return false;
  const int nkt = a.K / BK;
  const int last = nkt - (int)(grid.y - 1) * a.ktiles_per_split;          // K-tiles of the last split
  if (a.ktiles_per_split < 2 || last < 2) return false;
  if (grid.y > 1 && !(a.dbg & 32)) {                                       // split-major XCD placement (dbg bit 5: the 2-D grid of round 3's first form)
    GemmArgs b = a;
    b.split_major = (int)grid.x;
    hipLaunchKernelGGL(gemm_a4w_kernel, dim3(grid.x * grid.y), dim3(256), 0, stream, b);
    return true;
  }
  hipLaunchKernelGGL(gemm_a4w_kernel, grid, dim3(256), 0, stream, a);
  return true;
}

}  // namespace CCLIP_NS
