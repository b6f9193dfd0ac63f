// tile configuration 4 ("streaming"): PERSISTENT 256x128 workgroups (8 waves of 64x64, 3 LDS stages) for the forward
// layout, whose epilogue is streamed out under the NEXT tile's K loop.
//
// Why (round-1 in-kernel timelines): with one tile per workgroup a tile lives 17-23 us of which
// only 10-12 us are K loop - the fill (first operand tiles, residual rows) and the epilogue stores are HBM time during
// which the matrix pipe idles, and they do not overlap across workgroups either.  Here a workgroup walks a static list
// of tiles: the operand DMA runs PD = 2 K-tiles ahead ACROSS tile boundaries (no fill bubble after the first tile), and
// the finished tile's accumulators move to a second register set that is written out one 8-column run per K
// iteration of the following tile (fp32 residual rows are fetched two iterations ahead of their use).
//
// The obstacle is the single in-order vmcnt queue: the K loop waits for "the DMA of K-tile kt" with s_waitcnt vmcnt(N),
// which must leave exactly the N younger operations in flight - now a mix of DMA, residual loads and stores.  Every
// iteration therefore issues its memory operations in a fixed order (residual loads, DMA group, stores; scheduling
// fences keep the groups apart) and carries the running counts, so N is exact: stores never have to be complete until
// two iterations later.
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {

// The operand DMA is issued through inline asm in this kernel: the waitcnt pass guards the first LDS read after every
// LDS-DMA it knows about with s_waitcnt vmcnt(0) (in straight-line code; the rolled loops of configurations 1-3 escape
// it), which would drain stores and prefetches every iteration.  All vmcnt waits here are explicit and exact.
__device__ __forceinline__ void glds16_asm(const void* gsrc, const void* lds_wave_base) {
  const unsigned off = (unsigned)(size_t)LDS_PTR(lds_wave_base);
  asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(off), "v"(gsrc) : "memory", "m0");
}

template <int PERM, int NSUB, int NW>
__device__ __forceinline__ void stage_tile_kc_asm(const bf16* __restrict__ G, long ld, int R, int Kend, int r0, int k0,
                                                  char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < NSUB * 16 / NW; ++i) {
    const int idx = wave + NW * i;
    const int sub = idx >> 4, rb = idx & 15;
    const int rp = rb * 8 + (lane >> 3);
    const int c = (lane & 7) ^ (rp & 7);
    int r = rp;
    if (PERM) r = (rp & 64) + nperm((rp >> 4) & 3, rp & 15);
    int gr = r0 + sub * 128 + r; gr = gr < R ? gr : R - 1;
    const int gk = k0 + c * 8;
    const bf16* src = G + (long)gr * ld + gk;
    if (gk >= Kend) src = (const bf16*)g_zero16;
    glds16_asm(src, lds_tile + sub * TILE_BYTES + rb * 1024);
  }
}

#define S4_WAIT_CASE(n) case n: asm volatile("s_waitcnt vmcnt(" #n ")\n\ts_barrier" ::: "memory"); break;

// EPI 0: one 16-bit output (bias, activation).  EPI 1: pre-activation + activation outputs (training fc).
// EPI 2: fp32 output = alpha*acc + bias + fp32 residual (out-proj / proj; may be in place).
template <int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_stream_kernel(const GemmArgs p, int ntiles) {
  constexpr int MT = 4, NW = 8, BM_ = 256, BN_ = 128, STAGES = 3, PD = 2;
  constexpr int NSA = 2, NSB = 1, STAGE_BYTES_ = (NSA + NSB) * TILE_BYTES, G = (NSA + NSB) * 16 / NW;
  constexpr int R = EPI == 2 ? 2 : 0;                 // loads per streamed chunk (fp32 residual run = two float4)
  constexpr int S = EPI == 0 ? 1 : 2;                 // stores per streamed chunk
  constexpr int LA = EPI == 2 ? 2 : 0;                // chunk c is loaded at position c and written at position c + LA
  constexpr int NPOS = 8 + LA;                        // unrolled (compile-time chunk index) positions per tile
  // ONE LDS object (stages + bias): with two, the LDS lowering attaches alias scopes and the waitcnt pass then guards every
  // fragment read with its own s_waitcnt vmcnt(0) against the in-flight LDS-DMA - which would serialise the whole pipeline
  __shared__ __attribute__((aligned(16))) char smem[STAGES * STAGE_BYTES_ + 4096 * 4];
  float* bias_s = (float*)(smem + STAGES * STAGE_BYTES_);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n = p.N / BN_;
  const int KT = (p.K + BK - 1) / BK;
  const int nwg = gridDim.x;
  const int wq = xcd_remap(blockIdx.x, nwg);
  const int nmine = (ntiles - wq + nwg - 1) / nwg;
  const int wm = wave >> 1, wn = wave & 1;
  const int wm0 = wm * 64, wn0 = wn * 64;
  const int a_off = (wm0 >> 7) * TILE_BYTES, a_row = wm0 & 127;
  const int b_row = wn * 64;
  const int li = lane & 15, g = lane >> 4;

  for (int i = tid; i < p.N; i += 512) bias_s[i] = p.bias ? p.bias[i] : 0.f;
  __syncthreads();

  // ---- operand DMA cursor: runs PD K-tiles ahead of the multiply, across tile boundaries ----
  int ij = 0, ikt = 0, istage = 0;
  int ibm0 = (wq / tiles_n) * BM_, ibn0 = (wq % tiles_n) * BN_;
  auto issue_one = [&]() {
    char* sb = smem + istage * STAGE_BYTES_;
    stage_tile_kc_asm<0, NSA, NW>(p.A, p.lda, p.M, p.K, ibm0, ikt * BK, sb, wave, lane);
    stage_tile_kc_asm<1, NSB, NW>(p.B, p.ldb, p.N, p.K, ibn0, ikt * BK, sb + NSA * TILE_BYTES, wave, lane);
    istage = istage + 1 == STAGES ? 0 : istage + 1;
    if (++ikt == KT) {
      ikt = 0;
      if (ij + 1 < nmine) {        // past the last tile the cursor re-stages that tile: nobody reads it, counts stay uniform
        ++ij;
        const int t = wq + ij * nwg;
        ibm0 = (t / tiles_n) * BM_; ibn0 = (t % tiles_n) * BN_;
      }
    }
  };

  f32x4 acc[MT][4], accp[MT][4];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; accp[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float rres[3][8];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 8; ++r) rres[i][r] = 0.f;

  int bm0 = ibm0, bn0 = ibn0, pbm0 = 0, pbn0 = 0;    // current / previous tile origin
  issue_one();
  issue_one();
  int tail_prev2 = 0, tail_prev = 0, all_prev = G;   // vmcnt bookkeeping: operations younger than the DMA about to be awaited
  int cur = 0;

  // math + stores of one 8-column run (chunk c = (m-tile c>>1, half c&1)) of the PREVIOUS tile
  // the bias run of a chunk is read from LDS at the TOP of the iteration (before that iteration's DMA group in program
  // order): an LDS read placed after an LDS-DMA issue makes the waitcnt pass insert s_waitcnt vmcnt(0) in front of it
  auto load_bias = [&](auto ctag, float (&bb)[8]) {
    constexpr int C = decltype(ctag)::value;
    const int n0 = pbn0 + wn0 + 32 * (C & 1) + 8 * g;
    const float4 b0 = *(const float4*)(bias_s + n0), b1 = *(const float4*)(bias_s + n0 + 4);
    bb[0] = b0.x; bb[1] = b0.y; bb[2] = b0.z; bb[3] = b0.w; bb[4] = b1.x; bb[5] = b1.y; bb[6] = b1.z; bb[7] = b1.w;
  };
  auto write_chunk = [&](auto ctag, const float (&res)[8], const float (&bb)[8]) {
    constexpr int C = decltype(ctag)::value;
    constexpr int mt = C >> 1, h = C & 1;
    const int m = pbm0 + wm0 + 16 * mt + li;
    const int nl = wn0 + 32 * h + 8 * g;
    const int n0 = pbn0 + nl;
    float v[8];
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[r] = accp[mt][2 * h][r]; v[4 + r] = accp[mt][2 * h + 1][r]; }
#pragma unroll
    for (int r = 0; r < 8; ++r) v[r] = v[r] * p.alpha + bb[r];
    if (EPI == 1) {
      bf16x8 t;
#pragma unroll
      for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
      *(bf16x8*)(p.out_pre + (long)m * p.ldc + n0) = t;
    }
    if (ACT != CCLIP_ACT_NONE) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] = act_apply<ACT>(v[r], 0.f);
    }
    if (EPI == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += res[r];
      float* o = p.out_f32 + (long)m * p.ldc + n0;
      *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
      bf16x8 t;
#pragma unroll
      for (int r = 0; r < 8; ++r) t[r] = (bf16)v[r];
      *(bf16x8*)(p.out_bf16 + (long)m * p.ldc + n0) = t;
    }
  };
  auto load_res = [&](auto ctag, float (&res)[8]) {
    constexpr int C = decltype(ctag)::value;
    constexpr int mt = C >> 1, h = C & 1;
    const float* rp = p.residual + (long)(pbm0 + wm0 + 16 * mt + li) * p.ldr + pbn0 + wn0 + 32 * h + 8 * g;
    const float4 r0 = *(const float4*)rp, r1 = *(const float4*)(rp + 4);
    res[0] = r0.x; res[1] = r0.y; res[2] = r0.z; res[3] = r0.w; res[4] = r1.x; res[5] = r1.y; res[6] = r1.z; res[7] = r1.w;
  };

  // One K iteration.  POS = unrolled position inside a tile whose predecessor is being streamed out (-1: no streaming)
  auto k_iter = [&](auto pos_tag) {
    constexpr int POS = decltype(pos_tag)::value;
    constexpr bool LD = EPI == 2 && POS >= 0 && POS < 8;                // residual loads of chunk POS
    constexpr bool ST = POS >= LA && POS < 8 + LA;                      // math + stores of chunk POS - LA
    const int nyoung = tail_prev2 + all_prev;
    switch (nyoung) {
      S4_WAIT_CASE(0) S4_WAIT_CASE(1) S4_WAIT_CASE(2) S4_WAIT_CASE(3) S4_WAIT_CASE(4) S4_WAIT_CASE(5) S4_WAIT_CASE(6)
      S4_WAIT_CASE(7) S4_WAIT_CASE(8) S4_WAIT_CASE(9) S4_WAIT_CASE(10) S4_WAIT_CASE(11) S4_WAIT_CASE(12)
      default: asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (LD) load_res(std::integral_constant<int, (LD ? POS : 0)>{}, rres[(LD ? POS : 0) % 3]);
    float bb[8];
    if (ST) load_bias(std::integral_constant<int, (ST ? POS - LA : 0)>{}, bb);
    __builtin_amdgcn_sched_barrier(0);
    const char* At = smem + cur * STAGE_BYTES_ + a_off;
    const char* Bt = smem + cur * STAGE_BYTES_ + NSA * TILE_BYTES;
    bf16x8 xf[2][MT], wf[2][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[0][mt] = frag_rows(At, a_row + 16 * mt, 0, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[0][nt] = frag_rows(Bt, b_row + 16 * nt, 0, lane);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) xf[1][mt] = frag_rows(At, a_row + 16 * mt, 1, lane);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) wf[1][nt] = frag_rows(Bt, b_row + 16 * nt, 1, lane);
    issue_one();
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[0][nt], xf[0][mt], acc[mt][nt]);
    // k-step 0's MFMAs carry k-step 1's fragment reads and the DMA group
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    }
#pragma unroll
    for (int i = 0; i < G; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    // k-step 1's MFMAs carry the streamed chunk's math and stores
    if (ST) write_chunk(std::integral_constant<int, (ST ? POS - LA : 0)>{}, rres[(ST ? POS - LA : 0) % 3], bb);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = CCLIP_MFMA_16x16x32(wf[1][nt], xf[1][mt], acc[mt][nt]);
    if (ST) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
      }
#pragma unroll
      for (int i = 0; i < S; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
      }
    }
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
    __builtin_amdgcn_sched_barrier(0);
    tail_prev2 = tail_prev;
    tail_prev = ST ? S : 0;
    all_prev = (LD ? R : 0) + G + (ST ? S : 0);
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  };

  for (int j = 0; j < nmine; ++j) {
    if (j == 0) {
      for (int kt = 0; kt < KT; ++kt) k_iter(std::integral_constant<int, -1>{});
    } else {
      k_iter(std::integral_constant<int, 0>{}); k_iter(std::integral_constant<int, 1>{});
      k_iter(std::integral_constant<int, 2>{}); k_iter(std::integral_constant<int, 3>{});
      k_iter(std::integral_constant<int, 4>{}); k_iter(std::integral_constant<int, 5>{});
      k_iter(std::integral_constant<int, 6>{}); k_iter(std::integral_constant<int, 7>{});
      if (NPOS > 8) { k_iter(std::integral_constant<int, (NPOS > 8 ? 8 : 0)>{}); k_iter(std::integral_constant<int, (NPOS > 8 ? 9 : 0)>{}); }
      for (int kt = NPOS; kt < KT; ++kt) k_iter(std::integral_constant<int, -1>{});
    }
    // tile j is complete: it becomes the streamed-out tile of the next round
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) { accp[i][q] = acc[i][q]; acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    pbm0 = bm0; pbn0 = bn0;
    if (j + 1 < nmine) {
      const int t = wq + (j + 1) * nwg;
      bm0 = (t / tiles_n) * BM_; bn0 = (t % tiles_n) * BN_;
    }
  }
  // ---- the last tile has no successor to hide under: plain epilogue ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define FLUSH(C, I) do { float fb[8]; load_bias(std::integral_constant<int, C>{}, fb); write_chunk(std::integral_constant<int, C>{}, res[I], fb); } while (0)
#pragma unroll
  for (int half = 0; half < 2; ++half) {      // two batches of four runs: loads first, then math + stores
    float res[4][8];
    if (EPI == 2) {
      if (half == 0) {
        load_res(std::integral_constant<int, 0>{}, res[0]); load_res(std::integral_constant<int, 1>{}, res[1]);
        load_res(std::integral_constant<int, 2>{}, res[2]); load_res(std::integral_constant<int, 3>{}, res[3]);
      } else {
        load_res(std::integral_constant<int, 4>{}, res[0]); load_res(std::integral_constant<int, 5>{}, res[1]);
        load_res(std::integral_constant<int, 6>{}, res[2]); load_res(std::integral_constant<int, 7>{}, res[3]);
      }
    }
    if (half == 0) {
      FLUSH(0, 0); FLUSH(1, 1);
      FLUSH(2, 2); FLUSH(3, 3);
    } else {
      FLUSH(4, 0); FLUSH(5, 1);
      FLUSH(6, 2); FLUSH(7, 3);
    }
  }
}

// Supported: forward layout, M % 256 == 0, N % 128 == 0, N <= 4096, K >= 64 * (8 + LA + PD) so that a tile's streaming
// window ends before its own last PD iterations; 16-byte aligned outputs; no aux / split-K.
bool cclip_gemm_launch_cfg4(int lay, int act, hipStream_t stream, const GemmArgs& a) {
  if (lay != 3 || a.split_ws || a.aux) return false;
  if ((a.M & 255) || (a.N & 127) || a.N > 4096) return false;
  const int kt = (a.K + BK - 1) / BK;
  int epi;
  if (a.out_f32 && a.residual && !a.out_bf16 && !a.out_pre) epi = 2;
  else if (a.out_bf16 && a.out_pre && !a.out_f32 && !a.residual) epi = 1;
  else if (a.out_bf16 && !a.out_pre && !a.out_f32 && !a.residual) epi = 0;
  else return false;
  if (kt < (epi == 2 ? 10 : 8)) return false;
  if (epi == 2 && (a.ldr & 3)) return false;
  const int ntiles = (a.M / 256) * (a.N / 128);
  int grid = ntiles < 256 ? ntiles : 256;
  dim3 block(512);
#define L4(E, ACTV) hipLaunchKernelGGL((gemm_stream_kernel<E, ACTV>), dim3(grid), block, 0, stream, a, ntiles)
  if (epi == 2) { if (act != CCLIP_ACT_NONE) return false; L4(2, CCLIP_ACT_NONE); return true; }
  if (epi == 1) {
    if (act == CCLIP_ACT_QUICKGELU) { L4(1, CCLIP_ACT_QUICKGELU); return true; }
    return false;
  }
  switch (act) {
    case CCLIP_ACT_NONE: L4(0, CCLIP_ACT_NONE); return true;
    case CCLIP_ACT_QUICKGELU: L4(0, CCLIP_ACT_QUICKGELU); return true;
    default: return false;
  }
#undef L4
}

}  // namespace CCLIP_NS
