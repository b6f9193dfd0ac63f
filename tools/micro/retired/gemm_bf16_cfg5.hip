// tile configuration 5 ("deep"): 256x256 block, 2x4 waves of 128x64 as in configuration 3, but BK = 32 LDS stages: four
// 32 KiB stages instead of two 64 KiB ones, operand DMA issued FOUR k-steps ahead of its MFMAs and the fragments of k-step
// kt+1 read while k-step kt is multiplied (in place: every fragment register is refreshed right after its last use).
// Forward layout only.
//
// Why: with two 64-deep stages the DMA of K-tile kt+1 is issued at the start of iteration kt and
// must have landed by its end - an iteration can not be shorter than one HBM -> LDS round trip (1.5-2 us under load)
// although its MFMAs need 1 us; a 128x128 tile of configuration 1 lives as long as a 256x256 one for the same reason
// (tile life is latency, not work).  Here the round trip has four iterations of 0.5 us to hide in.
//
// LDS image of a 128-row x 32-k sub-tile (8 KiB): 64-byte rows would put rows r and r+4 on the same banks, so two
// consecutive rows share one 128-byte line (row 2R in slots 0-3, row 2R+1 in slots 4-7) and the line is swizzled like the
// 64-deep tiles (slot ^ (R & 7)): a ds_read_b128 fragment read (16 rows x one 16-byte chunk per 16-lane group) then touches
// 16 distinct 16-byte bank groups.  The DMA writes a 1 KiB block (8 lines) per wave-instruction; its per-lane SOURCE
// address carries the permutation.
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

#define SUB5 (128 * 32 * 2)                // one 8 KiB sub-tile: 128 rows x 32 k

// 16-byte global -> LDS DMA with a wave-uniform 64-bit base (SGPR pair) and a per-lane 32-bit byte offset: the per-lane part
// of an operand's source address is loop-invariant (4 VGPRs per wave for its 4 instructions), the k-step advances the base
__device__ __forceinline__ void glds16_sbase(const void* sbase, unsigned voff, const void* lds_wave_base) {
  const unsigned off = (unsigned)(size_t)LDS_PTR(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(off), "v"(voff), "s"(sbase) : "memory", "m0");
}

// per-lane byte offset (row * ld + k-chunk, in bytes) of DMA instruction `idx` (0 .. NSUB*8-1) of an operand tile
template <int PERM>
__device__ __forceinline__ unsigned stage5_off(long ld, int R, int r0, int idx, int lane) {
  const int sub = idx >> 3, rb = idx & 7;
  const int line = rb * 8 + (lane >> 3);                    // 128-byte line 0..63 of the sub-tile
  const int slot = (lane & 7) ^ (line & 7);                 // logical slot held at this lane's LDS position
  const int rp = 2 * line + (slot >> 2);                    // row position 0..127
  int r = rp;
  if (PERM) r = (rp & 64) + nperm((rp >> 4) & 3, rp & 15);
  int gr = r0 + sub * 128 + r; gr = gr < R ? gr : R - 1;    // clamp: rows past the edge are never stored
  return (unsigned)(((long)gr * ld + (slot & 3) * 8) * 2);
}

// fragment of row positions row0..row0+15: lane (i = l & 15, g = l >> 4) holds k = 8g..8g+7 of row row0 + i
__device__ __forceinline__ bf16x8 frag5(const char* tile, int row0, int lane) {
  const int row = row0 + (lane & 15), line = row >> 1;
  const int slot = 4 * (row & 1) + (lane >> 4);
  return *(const bf16x8*)(tile + line * 128 + ((slot ^ (line & 7)) << 4));
}

// timing-only ablation build (tools/micro/cfg5_ablate.py; results are wrong by construction): -DCCLIP_CFG5_ABLATE=mask with
// 1 = no in-loop DMA, 2 = no fragment refresh, 4 = no per-step wait + barrier, 8 = no epilogue
#ifdef CCLIP_CFG5_ABLATE
#define ABL5 CCLIP_CFG5_ABLATE
#else
#define ABL5 0
#endif
#define W5_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ") lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;

template <int ACT>
__global__ __launch_bounds__(512, 2) void gemm_deep_kernel(const GemmArgs p) {
  constexpr int MT = 8, NW = 8, WN = 4, BM_ = 256, BN_ = 256, STAGES = 4, PD = 3;
  constexpr int STAGE5 = 4 * SUB5;                          // A: 2 sub-tiles, B: 2 sub-tiles
  constexpr int G = 4;                                      // DMA instructions per wave per stage (32 / 8 waves)
  // ONE LDS object and inline-asm DMA: every vmcnt wait in this kernel is explicit (see gemm_bf16_cfg4.hip)
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE5];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = (p.N + BN_ - 1) / BN_;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int bm0 = (bid / tiles_n) * BM_, bn0 = (bid % tiles_n) * BN_;
  const int nkt = (p.K + 31) / 32;
  const int wm = wave / WN, wn = wave % WN;
  const int wm0 = wm * 128, wn0 = wn * 64;
  const int a_off = wm * SUB5;                              // this wave's A rows: sub-tile wm, rows 0..127
  const int b_off = 2 * SUB5 + (wn >> 1) * SUB5, b_row = (wn & 1) * 64;
  const int li = lane & 15, g = lane >> 4;

  f32x4 acc[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // this wave's four DMA instructions of a stage: two of A (idx wave, wave + 8), two of B
  const unsigned voff[4] = {stage5_off<0>(p.lda, p.M, bm0, wave, lane), stage5_off<0>(p.lda, p.M, bm0, wave + NW, lane),
                            stage5_off<1>(p.ldb, p.N, bn0, wave, lane), stage5_off<1>(p.ldb, p.N, bn0, wave + NW, lane)};
  auto dma = [&](int kt, int which) {
    char* sb = smem + (kt & (STAGES - 1)) * STAGE5 + (which < 2 ? 0 : 2 * SUB5);
    const int idx = wave + NW * (which & 1);
    const bf16* base = (which < 2 ? p.A : p.B) + kt * 32;   // K % 32 == 0 (launcher): no ragged contraction edge
    glds16_sbase(base, voff[which], sb + (idx >> 3) * SUB5 + (idx & 7) * 1024);
  };
  // Fragments: ONE register set each (8 of A, 4 of B), refreshed IN PLACE for k-step kt+1 as soon as k-step kt is done with
  // them - no double buffer: acc (128) + fragments (48) leave room for addresses, and the kernel must not spill (a scratch
  // reload is a vmcnt wait that would drain the DMA pipeline).  A k-step multiplies in two halves: n-tiles 0,1 against every
  // m-tile, then n-tiles 2,3 - wf[0], wf[1] are free after the first half, xf[mt] after its pair of the second half, wf[2],
  // wf[3] at the end; every refreshed fragment is next used at least 14 MFMAs (224 clocks) later.
  bf16x8 xf[MT], wf[4];

  // ---- prologue: stages 0..PD in flight, stage 0 landed, its fragments in registers ----
#pragma unroll
  for (int s = 0; s <= PD; ++s)
    if (s < nkt) { dma(s, 0); dma(s, 1); dma(s, 2); dma(s, 3); }
  {
    const int young = (nkt - 1 < PD ? nkt - 1 : PD) * G;    // DMA instructions younger than stage 0's
    switch (young) { W5_CASE(0) W5_CASE(4) W5_CASE(8) default: asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
  }
  {
    const char* At = smem + a_off;
    const char* Bt = smem + b_off;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = frag5(At, 16 * mt, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[nt] = frag5(Bt, b_row + 16 * nt, lane);
  }

  // One k-step.  RD: stage kt+1 exists (wait for it, refresh the fragments from it); DMA: stage kt+1+PD exists.
  auto step = [&](int kt, auto rd_tag, auto dma_tag) {
    constexpr bool RD = decltype(rd_tag)::value, DMA = decltype(dma_tag)::value;
    const char* At = smem + ((kt + 1) & (STAGES - 1)) * STAGE5 + a_off;
    const char* Bt = smem + ((kt + 1) & (STAGES - 1)) * STAGE5 + b_off;
    if (RD && !(ABL5 & 4)) {
      // stage kt+1 has landed for this wave once only the younger stages' DMAs are outstanding; the barrier publishes every
      // wave's part of it and proves every wave is done reading stage kt (whose buffer the DMA below overwrites)
      int last = kt + PD; last = last < nkt - 1 ? last : nkt - 1;          // youngest stage issued so far
      const int young = (last - (kt + 1)) * G;
      switch (young) { W5_CASE(0) W5_CASE(4) default: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt][0] = CCLIP_MFMA_16x16x32(wf[0], xf[mt], acc[mt][0]);
      acc[mt][1] = CCLIP_MFMA_16x16x32(wf[1], xf[mt], acc[mt][1]);
    }
    // refresh for k-step kt+1.  In place where the register is already free (wf[0], wf[1] now; xf[0..3] after their second-
    // half pair); through a second register set for the fragments that stay in use until the end of the step (wf[2], wf[3],
    // xf[4..7]: read early, moved over at the end) - no LDS read is issued in the last 8 MFMAs of a step, so the wait at the
    // top of the next one finds them all complete.
    bf16x8 wn[2], xn[4];
    if (RD && (ABL5 & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = xf[4 + i];
      wn[0] = wf[2]; wn[1] = wf[3];
    }
    if (RD && !(ABL5 & 2)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = frag5(At, 16 * (4 + i), lane);
      wn[0] = frag5(Bt, b_row + 32, lane); wn[1] = frag5(Bt, b_row + 48, lane);
      wf[0] = frag5(Bt, b_row, lane); wf[1] = frag5(Bt, b_row + 16, lane);
    }
    // program order pins the DMA instructions behind the fragment reads (both are memory operations for the compiler);
    // the scheduling groups below deal the reads out between the MFMA pairs
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[mt][2] = CCLIP_MFMA_16x16x32(wf[2], xf[mt], acc[mt][2]);
      acc[mt][3] = CCLIP_MFMA_16x16x32(wf[3], xf[mt], acc[mt][3]);
      if (RD && mt < 4 && !(ABL5 & 2)) xf[mt] = frag5(At, 16 * mt, lane);
      if (DMA && mt >= 4 && !(ABL5 & 1)) dma(kt + 1 + PD, mt - 4);
    }
    if (RD) {
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_barrier(0);
      wf[2] = wn[0]; wf[3] = wn[1];
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[4 + i] = xn[i];
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  int kt = 0;
  for (; kt + 1 + PD < nkt; ++kt) step(kt, std::true_type{}, std::true_type{});      // steady state
  for (; kt + 1 < nkt; ++kt) step(kt, std::true_type{}, std::false_type{});            // last PD stages: nothing left to issue
  step(kt, std::false_type{}, std::false_type{});                                      // last k-step

  // ---- epilogue (as in gemm_bf16_kernel, batches of two m-tiles: loads of the batch first, then math + stores) ----
  constexpr bool HAS_AUX = false;
  constexpr int EB = 2;
  float rres[EB][2][8];
  bf16x8 raux[1][2];
  float bsv[2][8];
  epi_bias(p, bn0 + wn0, g, bsv);
#pragma unroll
  for (int mb = 0; mb < MT; mb += EB) {
    epi_loads<EB, HAS_AUX, true, true>(p, bm0 + wm0, bn0 + wn0, mb, li, g, rres, raux);
#pragma unroll
    for (int mi = 0; mi < EB; ++mi) {
      const int mt = mb + mi;
      const int m = bm0 + wm0 + 16 * mt + li;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int n0 = bn0 + wn0 + 32 * h + 8 * g;
        if (m >= p.M || n0 >= p.N || ((ABL5 & 8) && p.alpha != 123.f)) continue;
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = acc[mt][2 * h][r]; v[4 + r] = acc[mt][2 * h + 1][r]; }
        const bool full = n0 + 8 <= p.N;
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
          for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r], 0.f);
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

// Supported: forward layout (both operands K-contiguous), K % 32 == 0, operands below 2 GiB, no split-K, no aux operand,
// activation none / QuickGELU.
bool cclip_gemm_launch_cfg5(int lay, int act, hipStream_t stream, const GemmArgs& a) {
  if (lay != 3 || a.split_ws || a.aux || a.colsum_dst) return false;
  if ((a.K & 31) || (long)a.M * a.lda >= (1L << 30) || (long)a.N * a.ldb >= (1L << 30)) return false;   // 32-bit DMA byte offsets
  const int tiles = ((a.M + 255) / 256) * ((a.N + 255) / 256);
  dim3 grid(tiles), block(512);
  switch (act) {
    case CCLIP_ACT_NONE: hipLaunchKernelGGL((gemm_deep_kernel<CCLIP_ACT_NONE>), grid, block, 0, stream, a); return true;
    case CCLIP_ACT_QUICKGELU: hipLaunchKernelGGL((gemm_deep_kernel<CCLIP_ACT_QUICKGELU>), grid, block, 0, stream, a); return true;
    default: return false;
  }
}

}  // namespace CCLIP_NS
