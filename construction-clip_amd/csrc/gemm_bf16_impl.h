// Implementation template of the bf16 MFMA GEMM (see gemm_bf16.hip for the description); included by the
// per-configuration translation units gemm_bf16_cfg*.hip so that they compile in parallel.
#pragma once
#include <type_traits>
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

#define BK 64
#define TILE_BYTES (128 * 64 * 2)        // one 16 KiB sub-tile: 128 rows (or columns) x 64 k

__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0, 0, 0, 0};

struct GemmArgs {
  const bf16* A; const bf16* B;
  long lda, ldb;
  int M, N, K;
  int ktiles_per_split;
  float alpha;
  const float* bias;
  const float* residual; long ldr;
  const bf16* aux; long ldaux;
  float* out_f32; bf16* out_bf16; bf16* out_pre; long ldc;
  int act;
  float* split_ws;   // != nullptr: raw fp32 partial tile stores to split_ws[z][M][N]
  // skinny (decode) path only, set by the native decode driver: A = LayerNorm(ln_x) computed in the kernel; columns
  // [kv_width, 3*kv_width) of the output also go to the KV cache row of each sequence
  const float* ln_x = nullptr; long ln_ldx = 0; const float* ln_gamma = nullptr; const float* ln_beta = nullptr;
  bf16* kv_k = nullptr; bf16* kv_v = nullptr; long kv_ld_seq = 0; int kv_width = 0;
  // wgrad layout only: row sums of A over K (= the bias gradient of the layer whose weight gradient this GEMM is);
  // with split-K the raw partials go to colsum_ws[z][M] and the combine kernel finishes them
  float* colsum_dst = nullptr; float* colsum_ws = nullptr; int colsum_acc = 0;
  int colsum_b = 0;   // 1: row sums of B (size N) instead - the Conv1D weight layout, where dY is the B operand
  // configurations 8 / 10: LayerNorm folded into the projection (cclip_hip.h): (mean, rstd) per row, column sums of the scaled weight;
  // the residual form's extra outputs for the next folded projection
  const float* ln_stats = nullptr; const float* ln_c1 = nullptr; float* rowstats = nullptr;
  int split_major = 0;   // configuration 11: tiles per split when the grid is ONE dimension of splits x tiles walked split-major (0: grid = (tiles, splits))
  int group_n = 0;    // tile order: 0 = row-major over (row tile, column tile); G > 0 = column groups of G tiles, row-major inside a group (tile_coords)
  int dbg = 0;        // timing ablations of configurations 8 / 10 (CCLIP_GEMM_DBG; bit 0: no epilogue) - never set by the product path
};


// n-permutation: position i (0..15) of MFMA n-tile nt (0..3) of a wave's 64-column block maps to
// local column P = 8*(i>>2) + 32*(nt>>1) + 4*(nt&1) + (i&3).  With the accumulator map
// (row = 4*(lane>>4) + reg) a lane then holds columns 8g..8g+7 (nt = 0,1) and 32+8g..32+8g+7
// (nt = 2,3) of its block.
__device__ __forceinline__ int nperm(int nt, int i) {
  return 8 * (i >> 2) + 32 * (nt >> 1) + 4 * (nt & 1) + (i & 3);
}
// Work-item -> tile.  Consecutive work-items run at the same time on one XCD (xcd_remap), so the order decides which operand
// panels that XCD's L2 holds together.  Row-major (gn = 0): a run of tiles covers few row panels x ALL column panels - right
// while the weight is small, but a wide weight (N = 3072: 4.7 MB of 16-bit rows at K = 768, more than one XCD's 4 MB of L2) is
// then cycled through L2 once per couple of row panels.  gn > 0: the column tiles are taken in groups of gn, all row panels of a
// group before the next group - the XCD keeps gn weight panels hot and re-reads the activation rows once per group.
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int gn, int& tm, int& tn) {
  if (gn <= 0 || gn >= tiles_n) { tm = bid / tiles_n; tn = bid % tiles_n; return; }
  const int ngr = (tiles_n + gn - 1) / gn;
  int grp = bid / (tiles_m * gn); grp = grp < ngr ? grp : ngr - 1;
  const int r = bid - grp * tiles_m * gn;
  const int rest = tiles_n - grp * gn, gw = rest < gn ? rest : gn;
  tm = r / gw; tn = grp * gn + r % gw;
}

__device__ __forceinline__ int fswz(int kr) { return ((kr & 3) << 2) | ((kr >> 2) & 3); }

// ---- staging: an operand tile = NSUB sub-tiles of 16 KiB (128 rows/cols x 64 k); one sub-tile = 16
// wave-instructions of 1 KiB; the NSUB*16 instructions are dealt round-robin to the NW waves ----
// PERM: LDS row position rp of a K-contiguous sub-tile holds tile row 64*(rp>>6) + nperm((rp>>4)&3, rp&15)
// (free at staging time because the DMA source address is per lane), so fragment reads stay natural.
// NINSTR: wave-instructions actually issued (default: whole sub-tiles); a 192-row A tile needs 24 of its 32
// one of those wave-instructions (index idx of the tile).  ASM: issued through inline asm, i.e. invisible to the compiler's
// waitcnt pass - see the K-strided K loop below for why.
template <int KC, int PERM, bool ASM = false>
__device__ __forceinline__ void stage_piece(const bf16* __restrict__ G, long ld, int R, int Kend, int r0, int k0,
                                            char* lds_tile, int idx, int lane) {
  const int sub = idx >> 4, rb = idx & 15;
  const bf16* src;
  if (KC) {
    const int rp = rb * 8 + (lane >> 3);                  // LDS row position 0..127 inside the sub-tile
    const int c = (lane & 7) ^ (rp & 7);                  // logical 16-B chunk held at this LDS slot
    int r = rp;
    if (PERM) r = (rp & 64) + nperm((rp >> 4) & 3, rp & 15);
    int gr = r0 + sub * 128 + r; gr = gr < R ? gr : R - 1;   // clamp: rows past the edge are never stored
    const int gk = k0 + c * 8;
    src = G + (long)gr * ld + gk;
    if (gk >= Kend) src = (const bf16*)g_zero16;
  } else {
    const int kr = rb * 4 + (lane >> 4);                  // k-row 0..63
    const int c = (lane & 15) ^ fswz(kr);
    const int rpad = ((R + 7) & ~7) - 8;                  // last 16-B chunk of the (8-padded) row
    int gc = r0 + sub * 128 + c * 8; gc = gc <= rpad ? gc : rpad;
    const int gk = k0 + kr;
    src = G + (long)gk * ld + gc;
    if (gk >= Kend) src = (const bf16*)g_zero16;          // ragged contraction edge contributes zeros
  }
  char* dst = lds_tile + sub * TILE_BYTES + rb * 1024;
  if (ASM) {
    const unsigned off = (unsigned)(size_t)LDS_PTR(dst);
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(off), "v"(src) : "memory", "m0");
  } else {
    glds16(src, dst);
  }
}

template <int KC, int PERM, int NSUB, int NW, int NINSTR = NSUB * 16, bool ASM = false>
__device__ __forceinline__ void stage_tile(const bf16* __restrict__ G, long ld, int R, int Kend, int r0, int k0,
                                           char* lds_tile, int wave, int lane) {
  static_assert(NINSTR % NW == 0, "every wave issues the same number of DMA instructions (counted vmcnt waits)");
#pragma unroll
  for (int i = 0; i < NINSTR / NW; ++i) stage_piece<KC, PERM, ASM>(G, ld, R, Kend, r0, k0, lds_tile, wave + NW * i, lane);
}

// ---- fragment reads (lane l: index i = l&15 of the 16-wide tile, k-group g = l>>4: k = 32ks+8g+j) ----
// K-contiguous tile, natural rows row0..row0+15.
__device__ __forceinline__ bf16x8 frag_rows(const char* tile, int row0, int ks, int lane) {
  const int row = row0 + (lane & 15);
  const int c = 4 * ks + (lane >> 4);
  return *(const bf16x8*)(tile + row * 128 + ((c ^ (row & 7)) << 4));
}
// K-strided tile; the tile's 16 columns are given as four 4-column pieces: piece p starts at column col_of_piece(p).
template <int PERM>
__device__ __forceinline__ bf16x8 frag_cols(const char* tile, int col0, int nt, int ks, int lane) {
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  int col = PERM ? col0 + 8 * p + 32 * (nt >> 1) + 4 * (nt & 1) : col0 + 16 * nt + 4 * p;
  tile += (col >> 7) * TILE_BYTES;          // a wave's columns may run on into the next 128-column sub-tile (192-row tiles)
  col &= 127;
  const int chunk = col >> 3, sub = (col & 7) * 2;
  const int kr0 = 32 * ks + 8 * g + q, kr1 = kr0 + 4;
  bf16x4 lo = lds_read_tr16(tile + kr0 * 256 + ((chunk ^ fswz(kr0)) << 4) + sub);
  bf16x4 hi = lds_read_tr16(tile + kr1 * 256 + ((chunk ^ fswz(kr1)) << 4) + sub);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

template <int ACT>
__device__ __forceinline__ float act_apply(float v, float a) {
  switch (ACT) {
    case CCLIP_ACT_QUICKGELU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));
    case CCLIP_ACT_TANH: return tanhf(v);
    case CCLIP_ACT_GELU_NEW: {
      const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
      return 0.5f * v * (1.0f + tanhf(u));
    }
    case CCLIP_ACT_RELU: return fmaxf(v, 0.0f);
    case CCLIP_ACT_DQUICKGELU: {   // v = upstream grad, a = saved pre-activation
      const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * a));
      return v * s * (1.0f + 1.702f * a * (1.0f - s));
    }
    case CCLIP_ACT_DTANH: return v * (1.0f - a * a);   // a = saved tanh output
    case CCLIP_ACT_DGELU_NEW: {
      const float u = 0.7978845608028654f * (a + 0.044715f * a * a * a);
      const float t = tanhf(u);
      const float du = 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * a * a);
      return v * (0.5f * (1.0f + t) + 0.5f * a * (1.0f - t * t) * du);
    }
    case CCLIP_ACT_DRELU: return a > 0.0f ? v : 0.0f;
    default: return v;
  }
}

__device__ __forceinline__ void epi_bias(const GemmArgs& p, int col_base, int g, float (&bsv)[2][8]) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n0 = col_base + 32 * h + 8 * g;
#pragma unroll
    for (int r = 0; r < 8; ++r) bsv[h][r] = 0.f;
    if (p.bias && !p.split_ws && n0 < p.N) {
      if (n0 + 8 <= p.N) {
        const float4 b0 = *(const float4*)(p.bias + n0), b1 = *(const float4*)(p.bias + n0 + 4);
        bsv[h][0] = b0.x; bsv[h][1] = b0.y; bsv[h][2] = b0.z; bsv[h][3] = b0.w;
        bsv[h][4] = b1.x; bsv[h][5] = b1.y; bsv[h][6] = b1.z; bsv[h][7] = b1.w;
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) if (n0 + r < p.N) bsv[h][r] = p.bias[n0 + r];
      }
    }
  }
}

// Epilogue operand loads (fp32 residual rows, 16-bit aux rows) of the EB m-tiles starting at m-tile mb.  A lane owns,
// per (m-tile, half h), the 8-column run n0..n0+7 of one row; ragged right edges are read per element.
template <int EB, bool HAS_AUX, bool DO_RES, bool DO_AUX>
__device__ __forceinline__ void epi_loads(const GemmArgs& p, int row_base, int col_base, int mb, int li, int g,
                                          float (&rres)[EB][2][8], bf16x8 (&raux)[HAS_AUX ? EB : 1][2]) {
#pragma unroll
  for (int mi = 0; mi < EB; ++mi) {
    const int m = row_base + 16 * (mb + mi) + li;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n0 = col_base + 32 * h + 8 * g;
      const bool live = m < p.M && n0 < p.N && !p.split_ws;
      const bool full = n0 + 8 <= p.N;
      if (DO_RES) {
#pragma unroll
        for (int r = 0; r < 8; ++r) rres[mi][h][r] = 0.f;
      }
      if (DO_RES && live && p.residual) {
        const float* rp = p.residual + (long)m * p.ldr + n0;
        if (full) {
          const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
          rres[mi][h][0] = r0.x; rres[mi][h][1] = r0.y; rres[mi][h][2] = r0.z; rres[mi][h][3] = r0.w;
          rres[mi][h][4] = r1.x; rres[mi][h][5] = r1.y; rres[mi][h][6] = r1.z; rres[mi][h][7] = r1.w;
        } else {
#pragma unroll
          for (int r = 0; r < 8; ++r) if (n0 + r < p.N) rres[mi][h][r] = rp[r];
        }
      }
      if (HAS_AUX && DO_AUX) {
        bf16x8 ax;
#pragma unroll
        for (int r = 0; r < 8; ++r) ax[r] = (bf16)0.f;
        if (live) {
          const bf16* ap = p.aux + (long)m * p.ldaux + n0;
          if (full) {
            ax = *(const bf16x8*)ap;
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (n0 + r < p.N) ax[r] = ap[r];
          }
        }
        raux[HAS_AUX ? mi : 0][h] = ax;
      }
    }
  }
}

// ---- fused epilogue of one wave's (16*MT) x 64 accumulator block whose first row / column are row_base / col_base ----
// (shared by every tile configuration; `interior`: the whole block tile lies inside M x N, N % 8 == 0, no split-K slab)
template <int ACT, int MT, bool PRE, int EB, bool HAS_AUX>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4 (&acc)[MT][4], const int row_base, const int col_base,
                                              const bool interior, const int li, const int g, float (&rres)[EB][2][8],
                                              bf16x8 (&raux)[HAS_AUX ? EB : 1][2], float (&bsv)[2][8]) {
  // ---- epilogue: lane holds, per (mt, h): 8 consecutive columns n0..n0+7 of row m ----
  if (!PRE) epi_bias(p, col_base, g, bsv);
  // FAST PATHS.  The general epilogue below serves every combination of outputs, ragged M / N edges and unaligned tails; it
  // compiles to ~5700 instructions in ~1000 basic blocks (exec-mask branches around every optional piece), of which a wave
  // executes a couple of thousand per tile - with K = 512 / 768 that is a visible share of a tile's life (PMC, round 2:
  // 3 VALU + 1 SALU instructions per MFMA over the whole kernel against 0.6 + 0.6 inside the K loop).  Interior tiles of the
  // four forms the hot path issues take a branch-free straight-line version instead; same arithmetic, same rounding.
  if (interior) {
    const long row0 = (long)(row_base + li) * p.ldc + (col_base + 8 * g);
    if (ACT < CCLIP_ACT_DQUICKGELU && p.out_bf16 && !p.out_f32 && !p.residual && (ACT != CCLIP_ACT_NONE || !p.out_pre)) {
      // 16-bit output (+ pre-activation when the activation's input is saved for backward): qkv, fc, every plain dgrad
      bf16* ob = p.out_bf16 + row0;
      bf16* op = p.out_pre ? p.out_pre + row0 : nullptr;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] = acc[mt][2 * h][r] * p.alpha + bsv[h][r];
            v[4 + r] = acc[mt][2 * h + 1][r] * p.alpha + bsv[h][4 + r];
          }
          const long o = (long)(16 * mt) * p.ldc + 32 * h;
          if (ACT != CCLIP_ACT_NONE) {
            if (op) {
              bf16x8 t;
#pragma unroll
              for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
              *(bf16x8*)(op + o) = t;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r], 0.f);
          }
          bf16x8 t;
#pragma unroll
          for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
          *(bf16x8*)(ob + o) = t;
        }
      return;
    }
    if (ACT == CCLIP_ACT_NONE && p.out_f32 && p.residual && !p.out_bf16 && !p.out_pre && p.ldr == p.ldc) {
      // fp32 residual stream: out = alpha*acc + bias + residual (usually in place): out-proj, c_proj.  The 64x64-per-wave
      // configurations fetched the residual rows before the K loop (rres); the others load them here, two m-tiles per batch
      const float* rp = p.residual + row0;
      float* of = p.out_f32 + row0;
#pragma unroll
      for (int mb = 0; mb < MT; mb += 2) {
        float rr[2][2][8];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            if (PRE) {
#pragma unroll
              for (int r = 0; r < 8; ++r) rr[mi][h][r] = rres[PRE ? mb + mi : 0][h][r];
            } else {
              const long o = (long)(16 * (mb + mi)) * p.ldc + 32 * h;
              const float4 t0 = *(const float4*)(rp + o), t1 = *(const float4*)(rp + o + 4);
              rr[mi][h][0] = t0.x; rr[mi][h][1] = t0.y; rr[mi][h][2] = t0.z; rr[mi][h][3] = t0.w;
              rr[mi][h][4] = t1.x; rr[mi][h][5] = t1.y; rr[mi][h][6] = t1.z; rr[mi][h][7] = t1.w;
            }
          }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int mt = mb + mi;
            const long o = (long)(16 * mt) * p.ldc + 32 * h;
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              v[r] = acc[mt][2 * h][r] * p.alpha + bsv[h][r];
              v[4 + r] = acc[mt][2 * h + 1][r] * p.alpha + bsv[h][4 + r];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] += rr[mi][h][r];
            *(float4*)(of + o) = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(of + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          }
      }
      return;
    }
    if (HAS_AUX && p.out_bf16 && !p.out_f32 && !p.out_pre && !p.residual && p.ldaux == p.ldc) {
      // activation derivative: out16 = act'(aux) * (alpha*acc + bias): the dgrad of the MLP's second projection
      const bf16* ap = p.aux + row0;
      bf16* ob = p.out_bf16 + row0;
#pragma unroll
      for (int mb = 0; mb < MT; mb += 2) {
        bf16x8 ax[2][2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            ax[mi][h] = PRE ? raux[(PRE && HAS_AUX) ? mb + mi : 0][h] : *(const bf16x8*)(ap + (long)(16 * (mb + mi)) * p.ldc + 32 * h);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int mt = mb + mi;
            bf16x8 t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              t[r] = (bf16)act_apply<ACT>(acc[mt][2 * h][r] * p.alpha + bsv[h][r], (float)ax[mi][h][r]);
              t[4 + r] = (bf16)act_apply<ACT>(acc[mt][2 * h + 1][r] * p.alpha + bsv[h][4 + r], (float)ax[mi][h][4 + r]);
            }
            *(bf16x8*)(ob + (long)(16 * mt) * p.ldc + 32 * h) = t;
          }
      }
      return;
    }
  }
  // General epilogue.  Two passes per batch of m-tiles: first ALL global loads of the batch (residual / aux) are issued, then
  // the math and the stores - one memory round trip per batch instead of one per 8-column run.
#pragma unroll
  for (int mb = 0; mb < MT; mb += EB) {
    // pass 1: loads (already in flight since kernel start when PRE)
    if (!PRE) epi_loads<EB, HAS_AUX, true, true>(p, row_base, col_base, mb, li, g, rres, raux);
    else if (HAS_AUX) epi_loads<EB, HAS_AUX, true, false>(p, row_base, col_base, mb, li, g, rres, raux);
    // pass 2: math + stores
#pragma unroll
    for (int mi = 0; mi < EB; ++mi) {
      const int mt = mb + mi;
      const int m = row_base + 16 * mt + li;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n0 = col_base + 32 * h + 8 * g;
        if (m >= p.M || n0 >= p.N) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
        if (p.split_ws) {
          float* o = p.split_ws + ((long)(p.split_major ? xcd_remap(blockIdx.x, gridDim.x) / p.split_major : (int)blockIdx.y) * p.M + m) * p.N + n0;
          *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
          if (n0 + 4 < p.N) *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          continue;
        }
        const bool full = n0 + 8 <= p.N;        // N need not be a multiple of 8: the last run is handled per element
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = v[r] * p.alpha + bsv[h][r];
        if (p.out_pre) {
          bf16* o = p.out_pre + (long)m * p.ldc + n0;
          if (full) {
            bf16x8 t;
#pragma unroll
            for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
            *(bf16x8*)o = t;
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (n0 + r < p.N) o[r] = (bf16)v[r];
          }
        }
        if (ACT != CCLIP_ACT_NONE) {
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r], (float)raux[HAS_AUX ? mi : 0][h][r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += rres[mi][h][r];
        if (p.out_f32) {
          float* o = p.out_f32 + (long)m * p.ldc + n0;
          if (full) {
            *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
          } else {
#pragma unroll
            for (int r = 0; r < 8; ++r) if (n0 + r < p.N) o[r] = v[r];
          }
        }
        if (p.out_bf16) {
          bf16* o = p.out_bf16 + (long)m * p.ldc + n0;
          if (full) {
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
  }
}

// WM x WN waves, each a (16*MT)x64 output sub-tile: block tile = (16*MT*WM) x (64*WN).  STAGES LDS stages; the DMA
// for tile kt+STAGES-1 is issued while tile kt is multiplied, with a COUNTED s_waitcnt vmcnt so that the
// younger stages stay in flight across the barrier (a plain __syncthreads would drain them).
// ROT = 1: ROTATED K loop (k_iter_rot below): the barrier of K-tile kt is followed by k-step 1 of tile kt-1 - whose
// fragments are already in registers - while the fragments of tile kt stream in, so the post-barrier fragment-read burst
// (8 waves x 12 ds_read_b128 before the first MFMA can issue) is no longer exposed time.
template <int A_KC, int B_KC, int ACT, int WM, int WN, int STAGES, int MT, int ROT = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_bf16_kernel(const GemmArgs p) {
  constexpr int NW = WM * WN, BM_ = 16 * MT * WM, BN_ = 64 * WN;
  constexpr int NSA = (BM_ + 127) / 128, NSB = (BN_ + 127) / 128;   // 128-wide sub-tiles per operand
  constexpr int STAGE_BYTES_ = (NSA + NSB) * TILE_BYTES;
  // A tiles that do not fill their last 128-row sub-tile (MT = 6: 192 rows) only stage the 8-row groups they use
  constexpr int AI = (A_KC && (BM_ & 127)) ? ((BM_ / 8 + NW - 1) / NW) * NW : NSA * 16;
  constexpr int G = (AI + NSB * 16) / NW;                         // DMA instructions per wave per stage
  constexpr int PD = STAGES - 1;                                  // prefetch distance in K-tiles
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES_];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN_ - 1) / BN_;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  int tm_i, tn_i;
  tile_coords(bid, (p.M + BM_ - 1) / BM_, tiles_n, p.group_n, tm_i, tn_i);
  const int bm0 = tm_i * BM_, bn0 = tn_i * BN_;
  const int nkt = (p.K + BK - 1) / BK;
  const int kt0 = blockIdx.y * p.ktiles_per_split;
  const int kt1 = (kt0 + p.ktiles_per_split < nkt) ? kt0 + p.ktiles_per_split : nkt;
  const int wm = wave / WN, wn = wave % WN;
  const int wm0 = wm * 16 * MT, wn0 = wn * 64;                    // offsets inside the block tile
  const int a_off = (wm0 >> 7) * TILE_BYTES, a_row = wm0 & 127;   // sub-tile + row/col offset inside it
  const int b_off = (wn >> 1) * TILE_BYTES, b_row = (wn & 1) * 64;
  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // 64x64-per-wave configurations have the registers to fetch the epilogue's residual / aux rows NOW, ahead of the
  // operand DMA: the loads are the oldest entries of the in-order vmcnt queue (every counted wait below stays
  // valid), they land under the K loop's MFMA phase, and the epilogue is left with math and fire-and-forget
  // stores - one full memory round trip per tile and half of the epilogue's HBM traffic leave the serial path.
  constexpr bool HAS_AUX = ACT >= CCLIP_ACT_DQUICKGELU;
  constexpr bool PRE = MT <= 4 && (A_KC || B_KC);          // (the wgrad layout spends those registers on the fused bias gradient)
  constexpr int EB = PRE ? MT : 2;                               // m-tiles per epilogue batch (register budget)
  const int li = lane & 15, g = lane >> 4;
  float rres[EB][2][8];
  bf16x8 raux[HAS_AUX ? EB : 1][2];
  float bsv[2][8];
  if (PRE) {
    epi_bias(p, bn0 + wn0, g, bsv);
    // (the activation-derivative kernels are at the 256-VGPR limit: they prefetch their aux rows only)
    epi_loads<EB, HAS_AUX, !HAS_AUX, HAS_AUX>(p, bm0 + wm0, bn0 + wn0, 0, li, g, rres, raux);
  }

  // A K-strided operand (the weight-gradient layout: both; Conv1D weights / dgrad without the transposed shadows: B) is read
  // with ds_read_b64_tr_b16 intrinsics, and for those the compiler's waitcnt pass puts s_waitcnt vmcnt(0) in front of the
  // first read after ANY LDS-DMA it knows to be in flight (found in round 2 by reading the loops' ISA: "s_waitcnt vmcnt(6);
  // s_barrier; s_waitcnt vmcnt(0); ds_read..." - the 3-stage kernels drained their whole prefetch queue every K-tile; the
  // 128x128 kernel, which issues its DMA right after the barrier, waited for the NEXT tile before multiplying this one).
  // The plain ds_read_b128 of the K-contiguous layout do not get that wait.  So these layouts issue their DMA through inline
  // asm (PIN): the compiler sees no LDS-DMA, the counted waits in front of the barriers are the only ones, and the DMA
  // instructions are placed between k-step 1's MFMAs by hand (sched_barrier pins; sched_group_barrier cannot see inline asm).
  constexpr bool PIN = !(A_KC && B_KC);
#pragma unroll
  for (int s = 0; s < PD; ++s) {
    if (kt0 + s < kt1) {
      char* sb = smem + s * STAGE_BYTES_;
      stage_tile<A_KC, 0, NSA, NW, AI, PIN>(p.A, p.lda, p.M, p.K, bm0, (kt0 + s) * BK, sb, wave, lane);
      stage_tile<B_KC, 1, NSB, NW, NSB * 16, PIN>(p.B, p.ldb, p.N, p.K, bn0, (kt0 + s) * BK, sb + NSA * TILE_BYTES, wave, lane);
    }
  }
  int cur = 0;                                                   // stage holding tile kt
  constexpr bool WG_LAYOUT = !A_KC && !B_KC && MT <= 6;     // (the 128x64-per-wave kernels have no registers left for it)
  // (Round 2 measured the B fragments of the weight-gradient layout in NATURAL column order - by the bank model their four
  // 4-column pieces sit in the same half of their 16-byte chunks, a 2-way conflict per ds_read_b64_tr_b16 - against this
  // permuted order: no difference, 232.9 -> 235.3 us on the qkv weight gradient.  The permutation stays everywhere.)
  constexpr int BPERM = 1;
  f32x4 accb[WG_LAYOUT ? MT : 1];
  bf16x8 ones8;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones8[j] = (bf16)1.0f;
#pragma unroll
  for (int i = 0; i < (WG_LAYOUT ? MT : 1); ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // One K-tile iteration.  DMA = true (steady state, kt + PD < kt1): the operand DMA of tile kt+PD is part of the same
  // basic block as the MFMAs and is dealt out BETWEEN them (one global_load_lds per few MFMAs): issuing the 6-8 DMA
  // instructions back to back cost ~500 clocks per wave per iteration with the matrix pipe idle (round-1 in-kernel
  // timestamps), almost as much as the iteration's MFMAs themselves.  DMA = false: the last PD iterations.
  auto k_iter = [&](int kt, auto dma_tag, auto cs_tag, int wait_tiles) {
    constexpr bool DMA = decltype(dma_tag)::value;
    constexpr int CS = decltype(cs_tag)::value;            // 0: none; 1: row sums of A; 2: row sums of B
    // In-process A/B against the previous build (tools/gemm_ab.py, MI355X): dealing the DMA out between k-step 1's
    // MFMAs (ILV) is worth -14..-34 % on the K-strided layouts of the 8-wave configurations (dgrad / wgrad: twice the
    // LDS read instructions) and -2..-7 % with 3 stages, but +5..10 % on 2-stage forward-layout kernels, whose DMA
    // then starts too late to land within one iteration: the 256x256 configuration deals it out between k-step 0's
    // FIRST MFMAs instead (ILV_EARLY, neutral), and the 128x128 one keeps the DMA ahead of the fragment reads.
    constexpr bool ILV = !PIN && (STAGES >= 3 || (MT >= 8 && !(A_KC && B_KC)) || (!A_KC && !B_KC));
    constexpr bool ILV_EARLY = !PIN && !ILV && MT >= 6;
    // tile kt has landed for this wave once at most the younger stages' DMAs are outstanding; the barrier then
    // (a) publishes every wave's part of tile kt and (b) proves every wave is done reading stage cur-1
    if (PD >= 3 && wait_tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");
    else if (PD >= 2 && wait_tiles >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (DMA && !PIN && !ILV && !ILV_EARLY) {
      int ns = cur + PD; ns = ns >= STAGES ? ns - STAGES : ns;
      char* sb = smem + ns * STAGE_BYTES_;
      stage_tile<A_KC, 0, NSA, NW, AI>(p.A, p.lda, p.M, p.K, bm0, (kt + PD) * BK, sb, wave, lane);
      stage_tile<B_KC, 1, NSB, NW>(p.B, p.ldb, p.N, p.K, bn0, (kt + PD) * BK, sb + NSA * TILE_BYTES, wave, lane);
      __builtin_amdgcn_sched_barrier(0);
    }
    const char* At = smem + cur * STAGE_BYTES_ + a_off;
    const char* Bt = smem + cur * STAGE_BYTES_ + NSA * TILE_BYTES + b_off;
    // register double-buffered fragments: the LDS reads of k-step 1 are issued before, and interleaved with,
    // the MFMAs of k-step 0, so LDS latency is exposed once per K-tile instead of once per 8 MFMAs
    bf16x8 xf[2][MT], wf[2][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[0][mt] = A_KC ? frag_rows(At, a_row + 16 * mt, 0, lane) : frag_cols<0>(At, a_row, mt, 0, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      wf[0][nt] = B_KC ? frag_rows(Bt, b_row + 16 * nt, 0, lane) : frag_cols<BPERM>(Bt, b_row, nt, 0, lane);
    if (DMA && ILV_EARLY) {   // program order: k-step 0 reads, DMA, k-step 1 reads
      int ns = cur + PD; ns = ns >= STAGES ? ns - STAGES : ns;
      char* sb = smem + ns * STAGE_BYTES_;
      stage_tile<A_KC, 0, NSA, NW, AI>(p.A, p.lda, p.M, p.K, bm0, (kt + PD) * BK, sb, wave, lane);
      stage_tile<B_KC, 1, NSB, NW>(p.B, p.ldb, p.N, p.K, bn0, (kt + PD) * BK, sb + NSA * TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      xf[1][mt] = A_KC ? frag_rows(At, a_row + 16 * mt, 1, lane) : frag_cols<0>(At, a_row, mt, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      wf[1][nt] = B_KC ? frag_rows(Bt, b_row + 16 * nt, 1, lane) : frag_cols<BPERM>(Bt, b_row, nt, 1, lane);
    constexpr int RDP = (A_KC ? MT : 2 * MT) + (B_KC ? 4 : 8);    // LDS read instructions per k-step
    if constexpr (PIN) {
      // k-step 0's MFMAs with k-step 1's reads between them (scheduled), then k-step 1's MFMAs with one DMA instruction of tile
      // kt+PD after every DSTEP-th of them (written out and pinned)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[0][nt], xf[0][mt], acc[mt][nt]);
      if (CS == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accb[mt] = CCLIP_MFMA_16x16x32(ones8, xf[0][mt], accb[mt]);
      }
      if (CS == 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accb[nt] = CCLIP_MFMA_16x16x32(wf[0][nt], ones8, accb[nt]);
      }
      __builtin_amdgcn_sched_group_barrier(0x100, RDP, 0);
#pragma unroll
      for (int i = 0; i < RDP; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * MT) / RDP > 0 ? (4 * MT) / RDP : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * MT + 4, 0);
      __builtin_amdgcn_sched_barrier(0);
      int ns = cur + PD; ns = ns >= STAGES ? ns - STAGES : ns;
      char* sb = smem + ns * STAGE_BYTES_;
      constexpr int GA = AI / NW;                                   // this wave's DMA instructions of the A tile; the rest are B's
      constexpr int DSTEP = (4 * MT) / G > 0 ? (4 * MT) / G : 1;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[1][nt], xf[1][mt], acc[mt][nt]);
          const int idx = 4 * mt + nt;
          if (DMA && idx % DSTEP == DSTEP - 1 && idx / DSTEP < G) {
            const int i = idx / DSTEP;
            __builtin_amdgcn_sched_barrier(0);
            if (i < GA) stage_piece<A_KC, 0, true>(p.A, p.lda, p.M, p.K, bm0, (kt + PD) * BK, sb, wave + NW * i, lane);
            else stage_piece<B_KC, 1, true>(p.B, p.ldb, p.N, p.K, bn0, (kt + PD) * BK, sb + NSA * TILE_BYTES, wave + NW * (i - GA), lane);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      if (CS == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accb[mt] = CCLIP_MFMA_16x16x32(ones8, xf[1][mt], accb[mt]);
      }
      if (CS == 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accb[nt] = CCLIP_MFMA_16x16x32(wf[1][nt], ones8, accb[nt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      cur = cur + 1 == STAGES ? 0 : cur + 1;
      return;
    }
    if (DMA && ILV) {     // after the fragment reads in program order (the DMA writes LDS: the reads may not sink below it)
      int ns = cur + PD; ns = ns >= STAGES ? ns - STAGES : ns;
      char* sb = smem + ns * STAGE_BYTES_;
      stage_tile<A_KC, 0, NSA, NW, AI>(p.A, p.lda, p.M, p.K, bm0, (kt + PD) * BK, sb, wave, lane);
      stage_tile<B_KC, 1, NSB, NW>(p.B, p.ldb, p.N, p.K, bn0, (kt + PD) * BK, sb + NSA * TILE_BYTES, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[ks][nt], xf[ks][mt], acc[mt][nt]);
      if (CS == 1) {     // row sums of A: one more MFMA per m-tile against an all-ones fragment (every output row then holds the sum)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accb[mt] = CCLIP_MFMA_16x16x32(ones8, xf[ks][mt], accb[mt]);
      }
      if (CS == 2) {     // row sums of B: ones on the other side (every output column then holds the sum)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) accb[nt] = CCLIP_MFMA_16x16x32(wf[ks][nt], ones8, accb[nt]);
      }
    }
    // schedule: the LDS reads of k-step 0 up front; k-step 1's fragments stream in between k-step 0's MFMAs; the DMA
    // instructions of tile kt+PD go between k-step 1's MFMAs
    constexpr int RD = (A_KC ? MT : 2 * MT) + (B_KC ? 4 : 8);     // LDS read instructions per k-step
    constexpr int NM = 4 * MT;                                    // MFMAs per k-step
    __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
    if (DMA && ILV_EARLY) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < RD; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM / RD > 0 ? NM / RD : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
    }
    if (DMA && ILV) {
#pragma unroll
      for (int i = 0; i < G; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NM / G > 0 ? NM / G : 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NM, 0);
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  };
  // ---- rotated iteration (ROT): loop-carried fragments of k-step 1.  Iteration kt: barrier; k-step 1 of tile kt-1 while
  // k-step 0 of tile kt is read; the DMA group of tile kt+PD; k-step 0 of tile kt while its k-step 1 is read.  Every LDS
  // read of stage(kt) still sits between barrier kt and barrier kt+1, so the stage hand-over is that of the plain loop.
  bf16x8 lxf[2][ROT ? MT : 1], lwf[2][4];
  auto k_iter_rot = [&](int kt, auto dma_tag, auto first_tag, int wait_tiles) {
    constexpr bool DMA = decltype(dma_tag)::value;
    constexpr bool FIRST = decltype(first_tag)::value;
    constexpr int LMT = ROT ? MT : 1;
    if (PD >= 3 && wait_tiles >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * G) : "memory");
    else if (PD >= 2 && wait_tiles >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const char* At = smem + cur * STAGE_BYTES_ + a_off;
    const char* Bt = smem + cur * STAGE_BYTES_ + NSA * TILE_BYTES + b_off;
#pragma unroll
    for (int mt = 0; mt < LMT; ++mt)
      lxf[0][mt] = A_KC ? frag_rows(At, a_row + 16 * mt, 0, lane) : frag_cols<0>(At, a_row, mt, 0, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      lwf[0][nt] = B_KC ? frag_rows(Bt, b_row + 16 * nt, 0, lane) : frag_cols<1>(Bt, b_row, nt, 0, lane);
    if (!FIRST) {
#pragma unroll
      for (int mt = 0; mt < LMT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(lwf[1][nt], lxf[1][mt], acc[mt][nt]);
    }
    if (DMA) {
      int ns = cur + PD; ns = ns >= STAGES ? ns - STAGES : ns;
      char* sb = smem + ns * STAGE_BYTES_;
      stage_tile<A_KC, 0, NSA, NW, AI>(p.A, p.lda, p.M, p.K, bm0, (kt + PD) * BK, sb, wave, lane);
      stage_tile<B_KC, 1, NSB, NW>(p.B, p.ldb, p.N, p.K, bn0, (kt + PD) * BK, sb + NSA * TILE_BYTES, wave, lane);
    }
    constexpr int RD = (A_KC ? LMT : 2 * LMT) + (B_KC ? 4 : 8);
    constexpr int NM = 4 * LMT;
    if (!FIRST) {      // k-step 1 of the previous tile carries this tile's k-step 0 reads, then the DMA group
#pragma unroll
      for (int i = 0; i < RD; ++i) {      // MFMA first: the wait for the loop-carried fragments then precedes every new read
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      if (DMA) {
#pragma unroll
        for (int i = 0; i < G; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, (NM - RD) / G > 0 ? (NM - RD) / G : 1, 0);
        }
      }
      __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, RD, 0);
      if (DMA) __builtin_amdgcn_sched_group_barrier(0x010, G, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < LMT; ++mt)
      lxf[1][mt] = A_KC ? frag_rows(At, a_row + 16 * mt, 1, lane) : frag_cols<0>(At, a_row, mt, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      lwf[1][nt] = B_KC ? frag_rows(Bt, b_row + 16 * nt, 1, lane) : frag_cols<1>(Bt, b_row, nt, 1, lane);
#pragma unroll
    for (int mt = 0; mt < LMT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(lwf[0][nt], lxf[0][mt], acc[mt][nt]);
#pragma unroll
    for (int i = 0; i < RD; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, NM / RD > 0 ? NM / RD : 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
    __builtin_amdgcn_sched_barrier(0);
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  };
  int kt = kt0;
  int cs = 0;
  if constexpr (WG_LAYOUT) {
    // one column block (A sums) / one row block (B sums) of tiles carries the bias gradient
    if (p.colsum_dst) cs = p.colsum_b ? (bm0 == 0 ? 2 : 0) : (bn0 == 0 ? 1 : 0);
    if (cs == 1) {
      for (; kt + PD < kt1; ++kt) k_iter(kt, std::true_type{}, std::integral_constant<int, 1>{}, PD - 1);
      for (; kt < kt1; ++kt) k_iter(kt, std::false_type{}, std::integral_constant<int, 1>{}, kt1 - 1 - kt);
    } else if (cs == 2) {
      for (; kt + PD < kt1; ++kt) k_iter(kt, std::true_type{}, std::integral_constant<int, 2>{}, PD - 1);
      for (; kt < kt1; ++kt) k_iter(kt, std::false_type{}, std::integral_constant<int, 2>{}, kt1 - 1 - kt);
    }
  }
  if (cs == 0) {
    if constexpr (ROT != 0) {
      if (kt < kt1) {
        if (kt + PD < kt1) k_iter_rot(kt, std::true_type{}, std::true_type{}, PD - 1);
        else k_iter_rot(kt, std::false_type{}, std::true_type{}, kt1 - 1 - kt);
        ++kt;
        for (; kt + PD < kt1; ++kt) k_iter_rot(kt, std::true_type{}, std::false_type{}, PD - 1);
        for (; kt < kt1; ++kt) k_iter_rot(kt, std::false_type{}, std::false_type{}, kt1 - 1 - kt);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)          // k-step 1 of the last tile
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(lwf[1][nt], lxf[1][mt], acc[mt][nt]);
      }
    } else {
      for (; kt + PD < kt1; ++kt) k_iter(kt, std::true_type{}, std::integral_constant<int, 0>{}, PD - 1);
      for (; kt < kt1; ++kt) k_iter(kt, std::false_type{}, std::integral_constant<int, 0>{}, kt1 - 1 - kt);
    }
  }
  if constexpr (WG_LAYOUT) {
    if (cs == 1 && wn == 0 && (lane >> 4) == 0) {         // D[any row][col li] = sum_k A(16 mt + li, k): take row 0
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = bm0 + wm0 + 16 * mt + (lane & 15);
        if (m < p.M) {
          if (p.split_ws) p.colsum_ws[(long)blockIdx.y * p.M + m] = accb[mt][0];
          else p.colsum_dst[m] = (p.colsum_acc ? p.colsum_dst[m] : 0.f) + accb[mt][0];
        }
      }
    }
    if (cs == 2 && wm == 0 && (lane & 15) == 0) {         // D[row 4g + r][any col] = sum_k B(n, k), n by the n-permutation
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = BPERM ? bn0 + wn0 + 32 * (nt >> 1) + 8 * (lane >> 4) + 4 * (nt & 1) + r
                              : bn0 + wn0 + 16 * nt + 4 * (lane >> 4) + r;
          if (n < p.N) {
            if (p.split_ws) p.colsum_ws[(long)blockIdx.y * p.N + n] = accb[nt][r];
            else p.colsum_dst[n] = (p.colsum_acc ? p.colsum_dst[n] : 0.f) + accb[nt][r];
          }
        }
    }
  }
  gemm_epilogue<ACT, MT, PRE, EB, HAS_AUX>(p, acc, bm0 + wm0, bn0 + wn0,
                                           bm0 + BM_ <= p.M && bn0 + BN_ <= p.N && !(p.N & 7) && !p.split_ws, li, g, rres, raux, bsv);
}

// launcher for one tile configuration; instantiates exactly the (layout, activation) pairs the hot path issues
template <int WM, int WN, int STAGES, int MT, int ROT = 0>
static bool gemm_launch_cfg(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a) {
  dim3 block(64 * WM * WN);
  if (a.colsum_dst && MT > 6) return false;                 // fused bias gradient: not in the 128x64-per-wave configuration (registers)
  if (ROT && lay != 3) return false;                         // the rotated loop is instantiated for the forward layout only
#define LAUNCH(AK, BKC, ACTV) hipLaunchKernelGGL((gemm_bf16_kernel<AK, BKC, ACTV, WM, WN, STAGES, MT, ROT>), grid, block, 0, stream, a)
  if (lay == 3) {
    switch (act) {
      case CCLIP_ACT_NONE: LAUNCH(1, 1, CCLIP_ACT_NONE); return true;
      case CCLIP_ACT_QUICKGELU: LAUNCH(1, 1, CCLIP_ACT_QUICKGELU); return true;
      case CCLIP_ACT_TANH: LAUNCH(1, 1, CCLIP_ACT_TANH); return true;
      case CCLIP_ACT_RELU: LAUNCH(1, 1, CCLIP_ACT_RELU); return true;
      case CCLIP_ACT_DGELU_NEW: LAUNCH(1, 1, CCLIP_ACT_DGELU_NEW); return true;
      case CCLIP_ACT_DQUICKGELU: LAUNCH(1, 1, CCLIP_ACT_DQUICKGELU); return true;   // dgrad through transposed weight shadows
      case CCLIP_ACT_DRELU: LAUNCH(1, 1, CCLIP_ACT_DRELU); return true;
      default: return false;
    }
  }
  if constexpr (ROT == 0) {      // (the rotated loop is instantiated for the forward layout only)
    if (lay == 2) {
      switch (act) {
        case CCLIP_ACT_NONE: LAUNCH(1, 0, CCLIP_ACT_NONE); return true;
        case CCLIP_ACT_GELU_NEW: LAUNCH(1, 0, CCLIP_ACT_GELU_NEW); return true;
        case CCLIP_ACT_DQUICKGELU: LAUNCH(1, 0, CCLIP_ACT_DQUICKGELU); return true;
        case CCLIP_ACT_DTANH: LAUNCH(1, 0, CCLIP_ACT_DTANH); return true;
        case CCLIP_ACT_DRELU: LAUNCH(1, 0, CCLIP_ACT_DRELU); return true;
        default: return false;
      }
    }
    if (lay == 0 && act == CCLIP_ACT_NONE) { LAUNCH(0, 0, CCLIP_ACT_NONE); return true; }
  }
  return false;
#undef LAUNCH
}

}  // namespace CCLIP_NS
