// RETIRED (round 2): never selected by the tuner - see DESIGN.md section 6.  Kept for reference, not built.
// tile configuration 6 ("streaming 192x256"): PERSISTENT workgroups with the K loop of configuration 5 (8 waves of 96x64,
// two 56 KiB LDS stages) whose finished tile is written out UNDER THE NEXT TILE'S K LOOP.
//
// Why: on the model's shapes (K = 512 / 768) a tile's epilogue is not hidden by anything - a CU moves its 96-384 KiB of
// outputs at ~15 B/clk while its matrix pipe idles, the same on every CU at the same time - and costs 25-40 % of the launch
// (round-2 A/B: the fc projection with its two 16-bit outputs takes 327 us, with one 268 us).
// Configuration 4 already streams the epilogue, but keeps the finished tile as a second fp32 accumulator set, which only
// fits 64x64 wave tiles, whose K iteration is LDS-bound; it never beat configuration 3 on the large projections.
// Here the finished tile is PARKED as packed 16-bit values (96 accumulators -> 48 registers; bias already added): with the
// 96x64 wave tile the kernel needs 186 + 48 registers.  One 8-column run per lane (one 16-byte store; two for the
// pre-activation + activation form) leaves per K iteration of the following tile, interleaved with its MFMAs.
//
// Epilogue forms (forward operand layout only; with the transposed weight shadows that covers the dgrad GEMMs too):
//   EPI 0: out16 = alpha*acc + bias                                  (qkv projection, dgrad outputs)
//   EPI 1: pre16 = alpha*acc + bias;  out16 = ACT(pre16 as stored)   (compiled, NOT offered: the parked tile is 16-bit, so
//          the activation would see the rounded pre-activation - different bits from every other configuration, and a
//          training forward that differs from the inference forward; it also measured 43 % slower than configuration 3)
// vmcnt: a wave has ONE in-order counter for its LDS-DMA and its stores, so "my DMA has landed" also means "every older store
// of mine has been acknowledged".  Measured (round 2, PMC): with every wave issuing both each iteration a store had less
// than one iteration to retire and the DMA waits grew by ~900 clocks per iteration (267 vs 225 us on the qkv projection).
// The two wave groups (waves 0-3 / 4-7) therefore ALTERNATE roles: in an even iteration group A issues the whole DMA group
// of the next K-tile (14 instructions per wave) while group B issues its streamed stores (two iterations' worth), in an
// odd iteration the other way round.  A wave waits (vmcnt(0)) only at the top of the iteration after its DMA turn - by
// then its last stores are ~1.5 iterations old - and never carries a store across its own DMA wait window.
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

// LDS-DMA with a SCALAR base and a 32-bit per-lane offset: `global_load_lds_dwordx4 voff, s[base:base+1]`.  The per-lane
// part of an operand address (row * ld + swizzled 16-byte chunk) never changes during the launch - every tile origin and
// K offset is workgroup-uniform - so each wave keeps 7 offset registers for the whole kernel and a K-tile's DMA group is
// 7 x (s_mov m0, load) plus two scalar adds, instead of ~200 address instructions per iteration.
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(sbase) : "memory", "m0");
}

template <int EPI, int ACT, int CPI>
__global__ __launch_bounds__(512, 2) void gemm_stream6_kernel(const GemmArgs p, int ntiles) {
  constexpr int MT = 6, BM_ = 192, BN_ = 256, STAGES = 2;
  constexpr int A_BYTES = BM_ * 128, B_BYTES = BN_ * 128, STAGE_BYTES_ = A_BYTES + B_BYTES;     // 24 + 32 = 56 KiB
  constexpr int NCH = 2 * MT;                                     // 12 chunks (m-tile, half) per tile
  constexpr int NPOS = NCH / CPI;                                 // iterations of the streaming window (12 or 6)
  constexpr int CPS = 2 * CPI;                                    // chunks per store turn (a wave stores every other iteration)
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES_ + 4096 * 4];
  float* bias_s = (float*)(smem + STAGES * STAGE_BYTES_);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool grpA = wave < 4;
  const int w4 = wave & 3;
  const int tiles_n = p.N / BN_;
  const int KT = p.K / BK;                                        // even (host-checked): role parity is the same in every tile
  const int nwg = gridDim.x;
  const int wq = xcd_remap(blockIdx.x, nwg);
  const int nmine = (ntiles - wq + nwg - 1) / nwg;
  const int wm = wave >> 2, wn = wave & 3;
  const int wm0 = wm * 96, wn0 = wn * 64;
  const int li = lane & 15, g = lane >> 4;

  for (int i = tid; i < p.N; i += 512) bias_s[i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();

  // ---- operand DMA: instruction idx = w4 + 4 i (i < 6 for A, < 8 for B) of the issuing group covers LDS rows
  // 8 idx .. 8 idx + 7; lane -> row rp = 8 w4 + (lane >> 3) + 32 i and 16-byte slot lane & 7, holding logical chunk
  // (lane & 7) ^ (rp & 7).  rp & 7 does not depend on i, and the n-permuted source row of a B position is 32 i + a lane
  // constant too (nperm is affine in the 32-row block index) - so ONE byte offset per operand and lane serves every
  // instruction, the rest (32 i rows, tile origin, K offset) is scalar.
  const int rlow = 8 * w4 + (lane >> 3);                          // 0..31
  const unsigned swz = (unsigned)(((lane & 7) ^ (rlow & 7)) << 4);
  const unsigned offA = (unsigned)rlow * (unsigned)p.lda * 2u + swz;
  const unsigned offB = (unsigned)nperm(rlow >> 4, rlow & 15) * (unsigned)p.ldb * 2u + swz;
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(smem) + (unsigned)w4 * 1024u;
  const unsigned strideA = 64u * (unsigned)p.lda, strideB = 64u * (unsigned)p.ldb;   // bytes per 32-row block
  // the last row tile is anchored at M - 192: it overlaps its predecessor (those rows are computed twice, bit-identically)
  // instead of running past M, so no load is clamped and no store is predicated
  auto tile_origin = [&](int t, int& m0, int& n0) {
    m0 = (t / tiles_n) * BM_; if (m0 > p.M - BM_) m0 = p.M - BM_;
    n0 = (t % tiles_n) * BN_;
  };
  // ---- operand DMA cursor: one K-tile ahead of the multiply, across tile boundaries (advanced by every wave alike) ----
  int ij = 0, ikt = 0, istage = 0;
  int ibm0, ibn0;
  tile_origin(wq, ibm0, ibn0);
  auto cursor_advance = [&]() {
    istage ^= 1;
    if (++ikt == KT) {
      ikt = 0;
      if (ij + 1 < nmine) {        // past the last tile the cursor re-stages that tile: nobody reads it
        ++ij;
        tile_origin(wq + ij * nwg, ibm0, ibn0);
      }
    }
  };
  auto issue_group = [&]() {       // the whole stage: 14 instructions per wave of the issuing group
    const bf16* ga = p.A + (long)ibm0 * p.lda + ikt * BK;
    const bf16* gb = p.B + (long)ibn0 * p.ldb + ikt * BK;
    const unsigned la = lds0 + (unsigned)istage * STAGE_BYTES_;
    unsigned va = offA, vb = offB;             // running per-lane offsets (one VALU add per instruction, one scalar base each)
#pragma unroll
    for (int i = 0; i < BM_ / 32; ++i) { glds16_s(va, ga, la + i * 4096u); va += strideA; }
#pragma unroll
    for (int i = 0; i < BN_ / 32; ++i) { glds16_s(vb, gb, la + A_BYTES + i * 4096u); vb += strideB; }
  };

  f32x4 acc[MT][4];
  bf16x4 park[MT][4];                                             // the previous tile: alpha*acc + bias, rounded to 16 bits
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      park[i][j] = (bf16x4){(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
    }

  int bm0 = ibm0, bn0 = ibn0, pbm0 = 0, pbn0 = 0;                // current / previous tile origin
  if (!grpA) issue_group();                                       // K-tile 0: by the group that waits at the top of iteration 0
  cursor_advance();
  int cur = 0;

  // stores of chunk c = (m-tile c >> 1, half c & 1) of the PREVIOUS tile: the lane's 8-column run n0..n0+7 of row m
  auto write_chunk = [&](auto ctag) {
    constexpr int C = decltype(ctag)::value;
    constexpr int mt = C >> 1, h = C & 1;
    const int m = pbm0 + wm0 + 16 * mt + li;
    const int n0 = pbn0 + wn0 + 32 * h + 8 * g;
    const bf16x8 t = __builtin_shufflevector(park[mt][2 * h], park[mt][2 * h + 1], 0, 1, 2, 3, 4, 5, 6, 7);
    if (EPI == 1) {
      *(bf16x8*)(p.out_pre + (long)m * p.ldc + n0) = t;
      bf16x8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (bf16)act_apply<ACT>((float)t[r], 0.f);
      *(bf16x8*)(p.out_bf16 + (long)m * p.ldc + n0) = o;
    } else {
      *(bf16x8*)(p.out_bf16 + (long)m * p.ldc + n0) = t;
    }
  };

  // One K iteration.  PAR = parity of the iteration inside its tile (KT is even).  The ISSUER group of this iteration
  // (group A in even ones) puts the whole DMA group of the next K-tile in flight; the other group issued the DMA of the
  // K-tile being multiplied now: it waits for it (vmcnt(0): its last stores are ~1.5 iterations old by then) and - inside a
  // streaming window (TURN >= 0) - writes chunks [CPS*TURN, CPS*TURN + CPS) of the previous tile.  Role tests are scalar
  // branches around three small blocks; the MFMA / fragment-read schedule is one path for both roles.
  auto k_iter = [&](auto par_tag, auto turn_tag) {
    constexpr int PAR = decltype(par_tag)::value;
    constexpr int TURN = decltype(turn_tag)::value;
    const bool issuer = grpA == (PAR == 0);
    if (!issuer) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    const char* At = smem + cur * STAGE_BYTES_;
    const char* Bt = At + A_BYTES;
    // fragments: A is refreshed IN PLACE one k-step ahead (xf[mt] of k-step 1 is read right after the four MFMAs that
    // consumed xf[mt] of k-step 0), B is double-buffered: 24 + 32 registers instead of 80 - the parked tile's 48 fit
    // without a spill (a spill reload inside this loop would wait on vmcnt, i.e. on the operand DMA in flight)
    bf16x8 xf[MT], wf[2][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[mt] = frag_rows(At, wm0 + 16 * mt, 0, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[0][nt] = frag_rows(Bt, wn0 + 16 * nt, 0, lane);
    if (issuer) issue_group();       // inline asm, fixed in place: k-step 0's reads are in flight while the DMA issues
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[1][nt] = frag_rows(Bt, wn0 + 16 * nt, 1, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[0][nt], xf[mt], acc[mt][nt]);
      xf[mt] = frag_rows(At, wm0 + 16 * mt, 1, lane);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);            // k-step 1's B fragments first
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (TURN >= 0 && !issuer) {                                   // the streamed chunks of this group's store turn
      write_chunk(std::integral_constant<int, (TURN >= 0 ? CPS * TURN : 0)>{});
      write_chunk(std::integral_constant<int, (TURN >= 0 ? CPS * TURN + 1 : 0)>{});
      if (CPS == 4) {
        write_chunk(std::integral_constant<int, (TURN >= 0 && CPS == 4 ? CPS * TURN + 2 : 0)>{});
        write_chunk(std::integral_constant<int, (TURN >= 0 && CPS == 4 ? CPS * TURN + 3 : 0)>{});
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[1][nt], xf[mt], acc[mt][nt]);
    __builtin_amdgcn_sched_barrier(0);
    cursor_advance();
    cur ^= 1;
  };
  // iteration pair (even, odd): group B stores turn T in the even iteration, group A in the odd one
#define PAIR(T)                                                                     \
  do {                                                                              \
    k_iter(std::integral_constant<int, 0>{}, std::integral_constant<int, (T)>{});   \
    k_iter(std::integral_constant<int, 1>{}, std::integral_constant<int, (T)>{});   \
  } while (0)

  for (int j = 0; j < nmine; ++j) {
    if (j == 0) {
      for (int kt = 0; kt < KT; kt += 2) PAIR(-1);
    } else {
      PAIR(0); PAIR(1); PAIR(2);
      if (NPOS > 6) { PAIR(NPOS > 6 ? 3 : 0); PAIR(NPOS > 6 ? 4 : 0); PAIR(NPOS > 6 ? 5 : 0); }
      for (int kt = NPOS; kt < KT; kt += 2) PAIR(-1);
    }
    // tile j is complete: park it (alpha, bias, rounding) - it is streamed out during the next tile's K loop
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n0 = bn0 + wn0 + 32 * h + 8 * g;
      const float4 b0 = *(const float4*)(bias_s + n0), b1 = *(const float4*)(bias_s + n0 + 4);
      const float bl[4] = {b0.x, b0.y, b0.z, b0.w}, bh[4] = {b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x4 lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          lo[r] = (bf16)(acc[mt][2 * h][r] * p.alpha + bl[r]);
          hi[r] = (bf16)(acc[mt][2 * h + 1][r] * p.alpha + bh[r]);
        }
        park[mt][2 * h] = lo; park[mt][2 * h + 1] = hi;
        acc[mt][2 * h] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[mt][2 * h + 1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    pbm0 = bm0; pbn0 = bn0;
    if (j + 1 < nmine) tile_origin(wq + (j + 1) * nwg, bm0, bn0);
  }
#undef PAIR
  // ---- the last tile has no successor to hide under: plain stores ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  write_chunk(std::integral_constant<int, 0>{}); write_chunk(std::integral_constant<int, 1>{});
  write_chunk(std::integral_constant<int, 2>{}); write_chunk(std::integral_constant<int, 3>{});
  write_chunk(std::integral_constant<int, 4>{}); write_chunk(std::integral_constant<int, 5>{});
  write_chunk(std::integral_constant<int, 6>{}); write_chunk(std::integral_constant<int, 7>{});
  write_chunk(std::integral_constant<int, 8>{}); write_chunk(std::integral_constant<int, 9>{});
  write_chunk(std::integral_constant<int, 10>{}); write_chunk(std::integral_constant<int, 11>{});
}

// Supported: forward layout, N % 256 == 0, N <= 4096, K % 128 == 0 and K >= 384 (the streaming window must end before the
// tile does: 6 iterations at two chunks per iteration, 12 at one), any M >= 192 (the last row tile is anchored at M - 192), 16-byte aligned 16-bit outputs; no aux / residual / fp32 output / split-K.
bool cclip_gemm_launch_cfg6(int lay, int act, hipStream_t stream, const GemmArgs& a) {
  if (lay != 3 || a.split_ws || a.aux || a.residual || a.out_f32 || !a.out_bf16) return false;
  if ((a.N & 255) || a.N > 4096 || (a.K & 127) || (a.ldc & 7)) return false;   // (an even number of K-tiles: the role parity)
  const int kt = a.K / BK;
  if (kt < 6) return false;
  // (the pre-activation + activation form compiles - EPI 1 - but its two stores and the activation VALU per chunk made it
  //  43 % slower than configuration 3 on the fc projection, 481 vs 337 us: not offered)
  if (a.out_pre || act != CCLIP_ACT_NONE) return false;
  const int epi = 0;
  if (a.M < 192 || (a.lda & 7) || (a.ldb & 7)) return false;
  if ((size_t)192 * a.lda * 2 >= (1ull << 31) || (size_t)256 * a.ldb * 2 >= (1ull << 31)) return false;   // 32-bit DMA offsets
  const int ntiles = ((a.M + 191) / 192) * (a.N / 256);
  const int grid = ntiles < 256 ? ntiles : 256;
  dim3 block(512);
#define L6(E, ACTV, C) hipLaunchKernelGGL((gemm_stream6_kernel<E, ACTV, C>), dim3(grid), block, 0, stream, a, ntiles)
  (void)epi;
  if (kt >= 12) L6(0, CCLIP_ACT_NONE, 1);
  else L6(0, CCLIP_ACT_NONE, 2);
#undef L6
  return true;
}

}  // namespace CCLIP_NS
