// Persistent KV-cached beam-search decoder (see cclip_gpt2_beam_search in include/cclip_hip.h).
//
// The reference's generate_beam (/root/reference/CLIP_prefix_caption/test.py:353-441, application.py:152-229) runs, per new
// token, the whole GPT-2 on the growing sequence and then ~15 small torch ops on the [beams, V] logits.  With a KV cache the
// step's arithmetic is a chain of GEMVs over 170 MB of weights (40 us of HBM time) - what is left is overhead: the
// launch-by-launch form of this repo (cclip_gpt2_decode_step: 62 dependent 5-13 us launches, then the selection ops and a
// cache gather from Python) takes 1.14 ms per step.  Here ONE kernel of one workgroup per CU runs every step of a caption:
//
//   per layer   P1  LayerNorm + qkv GEMV, k / v rows appended to the cache        (3D/32 column blocks)
//               P2  decode attention, one wave per (beam, head)
//               P3  out-proj GEMV + residual                                      (D/32 blocks)
//               P4  LayerNorm + fc GEMV + activation                              (hidden/32 blocks)
//               P5  proj GEMV + residual                                          (D/32 blocks)
//   then        LN_f + tied lm_head GEMV over a contiguous vocabulary slice per workgroup, which also leaves the slice's
//               per-beam (max, sum-exp, top-k) partials, and
//               the selection (workgroup 0): temperature / log-softmax / stopped-beam rule / length-normalised top-k, token
//               append, beam reorder, next input embedding - the arithmetic of test.py:395-428, restated.
//
// Phases are separated by a grid barrier (one atomic counter); the small buffers one phase hands to the next are written with
// write-through (sc1) stores and read with sc1 loads, which are coherent across the XCDs' L2s without cache-wide fences.  Beam reorder never copies the cache: `slot_of[t][b]` names the cache slot that holds beam
// b's key / value of position t, and reordering permutes that table.  A stopped caption ends the kernel (flag checked after a
// barrier, so every workgroup takes the same exit); a barrier that does not fill within ~1 s sets an error flag and is never
// waited on again, so the grid always drains.
#include "gemm_skinny_impl.h"

namespace CCLIP_NS {

#define BEAM_MAXL 24
#define BEAM_PS 20          // floats per (workgroup, beam) selection partial: max, sum, 8 x (value, index) + pad
#define BEAM_MAXR 256       // vocabulary rows per workgroup slice (4 per lane in the local top-k)

struct BeamArgs {
  int n_layer, nb, D, H, Hd, act, V, pos0, n_steps, first, stop_token, ld_tokens, max_len, rows_per_wg;
  float temperature;
  cclip_block_ptrs blocks[BEAM_MAXL];
  float* x;
  bf16* kc; bf16* vc; long ld_layer, ld_seq;
  bf16* scratch;
  const float* lnf_w; const float* lnf_b; const bf16* wte16;
  float* logits; long ld_logits; const float* first_logits;
  const float* wte32; const float* wpe32;
  int* slot_of; int* tokens; float* scores; float* seq_len; int* stopped;
  int* state;                 // [0] barrier counter, [1] error, [2] done, [3] selections made when every beam had stopped, [4] tokens per beam
  float* part;
};

// ---- phase hand-over --------------------------------------------------------------------------------------------------
// One monotonic counter.  A phase's PRODUCERS (the workgroups that had a column block / task / slice in it) add 1 when their
// part is written; every workgroup keeps the same running total of producers (`target`), and a workgroup waits for that total
// only when it is about to work in the next phase.  Idle workgroups neither add nor poll: with 24-96 of 256 workgroups active
// in a projection phase, a full barrier's 256 serialized atomics and 256 pollers were most of its 2-5 us.
// (The phase's hand-over buffers are written with write-through sc1 stores and read with sc1 loads - st_coh / ld_coh - so no
// cache-wide write-back / invalidate is needed: a release + acquire fence pair per workgroup per phase cost ~30 us per phase.)
struct PhaseSync {
  int* counter; int* err; int target; bool dead;
  // end of a phase that `nprod` workgroups worked in; `worked`: this workgroup was one of them
  __device__ __forceinline__ void arrive(int nprod, bool worked) {
    __syncthreads();                                               // every wave's stores of the phase are out (vmcnt(0) + barrier)
    target += nprod;
    if (worked && !dead && threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // before reading what the phases so far produced
  __device__ __forceinline__ void wait() {
    if (!dead && threadIdx.x == 0) {
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 23) || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) dead = true;   // uniform per workgroup: read after the barrier
  }
};

// ---- decode attention for one (beam, head) by one wave; p_l: S floats, ro_l: S row offsets, q_l: 64 floats of this wave ----
// Memory round trips are what this phase costs, so they are kept to two: the slot table of every key (one batch), then ALL key
// rows and value rows of a 128-key chunk in flight together (S <= 128 is one chunk - the caption lengths of this path).
__device__ __forceinline__ void attn_task(const BeamArgs& a, const bf16* q, long ldq, const bf16* kc, const bf16* vc, bf16* out,
                                          long ldo, int b, int h, int S, bool valid, float* p_l, int* ro_l, float* q_l) {
  const int lane = threadIdx.x & 63;
  const int nb = a.nb;
  for (int k0 = 0; k0 < S; k0 += 256) {                           // row offsets: cache slot of (position, beam) from the slot table
    int sl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int key = k0 + lane + 64 * u; sl[u] = ld_coh<true>(a.slot_of + (long)(key < S ? key : S - 1) * 8 + b); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + lane + 64 * u;
      const int v = sl[u] < 0 ? 0 : (sl[u] >= nb ? nb - 1 : sl[u]);
      if (key < S) ro_l[key] = (int)((long)v * a.ld_seq + (long)key * a.D + h * 64);
    }
  }
  q_l[lane] = (float)ld_coh<true>(q + (long)b * ldq + h * 64 + lane);
  __syncthreads();
  const int c = lane & 7, kg = lane >> 3;
  float m = -__builtin_inff(), l = 0.f;
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
  for (int c0 = 0; c0 < S; c0 += 128) {                           // online softmax over 128-key chunks
    bf16x8 kv[2][8], vv[4][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = c0 + lane + 64 * t;
      const bf16* kr = kc + ro_l[key < S ? key : S - 1];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) kv[t][cc] = ld_coh<true>((const bf16x8*)(kr + 8 * cc));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = c0 + 32 * t + 8 * u + kg;
        vv[t][u] = ld_coh<true>((const bf16x8*)(vc + ro_l[key < S ? key : S - 1] + 8 * c));
      }
    float sc[2], cm = -__builtin_inff();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = c0 + lane + 64 * t;
      float acc = 0.f;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += q_l[8 * cc + j] * (float)kv[t][cc][j];
      sc[t] = key < S ? acc * 0.125f : -__builtin_inff();
      cm = fmaxf(cm, sc[t]);
    }
    cm = wave_max(cm);
    const float mn = fmaxf(m, cm);
    const float resc = __expf(m - mn);                              // (first chunk: exp(-inf) = 0 on l = 0, o = 0)
    float cl = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int key = c0 + lane + 64 * t;
      const float e = key < S ? __expf(sc[t] - mn) : 0.f;
      p_l[lane + 64 * t] = e;
      cl += e;
    }
    l = l * resc + wave_sum(cl);
    m = mn;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= resc;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = p_l[32 * t + 8 * u + kg];                   // 0 for keys past S
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += w * (float)vv[t][u][j];
      }
    __syncthreads();
  }
  const float inv = 1.0f / l;
  bf16x8 ov;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float t = o[j];
    t += __shfl_xor(t, 8, 64);
    t += __shfl_xor(t, 16, 64);
    t += __shfl_xor(t, 32, 64);
    ov[j] = (bf16)(t * inv);
  }
  if (valid && kg == 0) st_coh<true>((bf16x8*)(out + (long)b * ldo + h * 64 + 8 * c), ov);
  __syncthreads();
}

// ---- per-beam partials of one vocabulary slice (sl: [n_in][BEAM_MAXR] logits of rows r0..r0+nr) ----------------------------
// wave w handles beams w, w+4: slice max and sum-exp of z = logit / T, and the slice's top-k by z (k = a.nb)
__device__ __forceinline__ void select_partials(const BeamArgs& a, const float* sl, int n_in, int r0, int nr, float inv_t_is_div) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float T = inv_t_is_div;
  for (int m = wave; m < n_in; m += 4) {
    float z[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = lane + 64 * i;
      z[i] = j < nr ? sl[m * BEAM_MAXR + j] / T : -__builtin_inff();
    }
    float mx = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += (lane + 64 * i < nr) ? expf(z[i] - mx) : 0.f;
    s = wave_sum(s);
    float* pp = a.part + ((long)blockIdx.x * 8 + m) * BEAM_PS;
    if (lane == 0) { st_coh<true>(pp, nr > 0 ? mx : -__builtin_inff()); st_coh<true>(pp + 1, nr > 0 ? s : 0.f); }
    for (int r = 0; r < a.nb; ++r) {                                // k rounds of wave arg-max (ties: the lower row first)
      float bv = z[0]; int bi = lane;
#pragma unroll
      for (int i = 1; i < 4; ++i) if (z[i] > bv) { bv = z[i]; bi = lane + 64 * i; }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      if (lane == 0) { st_coh<true>(pp + 2 + 2 * r, bv); st_coh<true>((int*)pp + 3 + 2 * r, r0 + bi); }
#pragma unroll
      for (int i = 0; i < 4; ++i) if (lane + 64 * i == bi) z[i] = -__builtin_inff();
    }
  }
}

// ---- the selection proper, by workgroup 0 (256 threads); lds: 64 + 512 + 4096 + 2 * n_in*G*k floats -------------------------------
template <int MCAP>
__device__ __forceinline__ void select_merge(const BeamArgs& a, int n_in, bool first, int it, int cur_pos, int G, int ntok, float* lds) {
  const int tid = threadIdx.x;
  const int nb = a.nb, k = a.nb;
  float* bM = lds;            // [8] global max per beam
  float* bS = lds + 8;        // [8] global sum per beam
  float* o_sc = lds + 16;     // [8] scores, [8] current lengths, [8] stopped (old beams)
  float* o_len = lds + 24;
  int* o_st = (int*)(lds + 32);
  float* w_avg = lds + 40;    // [8] winners
  int* w_flat = (int*)(lds + 48);
  float* red_v = lds + 64;    // [256] reduction scratch
  int* red_i = (int*)(lds + 64 + 256);
  float* stat = lds + 64 + 512;          // [8][256 max | 256 sum] slice statistics
  float* cav = stat + 4096;              // candidate averages [n_in * G * k]
  int* cfl = (int*)(cav + n_in * G * k); // candidate flat indices
  // per-beam softmax statistics from the workgroups' slice partials: thread g fetches slice g's (max, sum) of every beam (all
  // loads in flight), the reduction runs out of LDS in slice order (deterministic).  A serial loop over the slices is one
  // memory round trip per slice: 2 x 256 of them were 0.8 ms of a 1.4 ms step.
  {
    float pm[8], ps[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float* pp = a.part + ((long)(tid < G ? tid : 0) * 8 + (m < n_in ? m : 0)) * BEAM_PS;
      pm[m] = ld_coh<true>(pp); ps[m] = ld_coh<true>(pp + 1);
    }
#pragma unroll
    for (int m = 0; m < 8; ++m) { stat[m * 512 + tid] = tid < G ? pm[m] : -__builtin_inff(); stat[m * 512 + 256 + tid] = tid < G ? ps[m] : 0.f; }
  }
  __syncthreads();
  {
    // block max / sum by wave shuffles + the four waves' results through LDS (fixed order: reproducible)
    const int lane = tid & 63, wave = tid >> 6;
    float wm[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) wm[m] = wave_max(stat[m * 512 + tid]);
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) red_v[wave * 8 + m] = wm[m];
    }
    __syncthreads();
    if (tid < 8) bM[tid] = fmaxf(fmaxf(red_v[tid], red_v[8 + tid]), fmaxf(red_v[16 + tid], red_v[24 + tid]));
    __syncthreads();
    float ws[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      const float sg = stat[m * 512 + 256 + tid];
      ws[m] = wave_sum(sg > 0.f ? sg * expf(stat[m * 512 + tid] - bM[m]) : 0.f);
    }
    __syncthreads();
    if (lane == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) red_v[wave * 8 + m] = ws[m];
    }
    __syncthreads();
  }
  if (tid < 8) {
    const int m = tid;
    if (m < n_in) {
      bS[m] = ((red_v[m] + red_v[8 + m]) + red_v[16 + m]) + red_v[24 + m];
      const bool st = first ? false : a.stopped[m] != 0;
      o_st[m] = st ? 1 : 0;
      o_sc[m] = first ? 0.f : a.scores[m];
      o_len[m] = first ? 1.f : a.seq_len[m] + (st ? 0.f : 1.f);    // seq_lengths[~is_stopped] += 1 (not in the first selection)
    }
  }
  __syncthreads();
  const int C = n_in * G * k;
  for (int c0 = tid; c0 < C; c0 += 2304) {                          // candidates: nine per thread in flight
    float zz[9]; int tk[9];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int c = c0 + 256 * u < C ? c0 + 256 * u : C - 1;
      const int r = c % k, g = (c / k) % G, m = c / (k * G);
      const float* pp = a.part + ((long)g * 8 + m) * BEAM_PS;
      zz[u] = ld_coh<true>(pp + 2 + 2 * r);
      tk[u] = ld_coh<true>((const int*)pp + 3 + 2 * r);
    }
#pragma unroll
    for (int u = 0; u < 9; ++u) {
      const int c = c0 + 256 * u;
      if (c >= C) continue;
      const int r = c % k, g = (c / k) % G, m = c / (k * G);
      const float z = zz[u];
      const int tok = tk[u];
      float avg = -__builtin_inff();
      int flat = 0x7fffffff;
      if (o_st[m]) {                                                // logits[is_stopped] = -inf; logits[is_stopped, 0] = 0
        if (g == 0 && r == 0) { avg = (o_sc[m] + 0.f) / o_len[m]; flat = m * a.V; }
      } else if (z > -__builtin_inff() && tok >= 0 && tok < a.V) {
        const float pr = expf(z - bM[m]) / bS[m];                   // softmax(-1) ...
        const float lp = logf(pr);                                  // ... .log()
        avg = (o_sc[m] + lp) / o_len[m];
        flat = m * a.V + tok;
      }
      cav[c] = avg; cfl[c] = flat;
    }
  }
  __syncthreads();
  for (int r = 0; r < k; ++r) {                                     // top-k of the flattened [beams x V] averages, best first
    float bv = -__builtin_inff(); int bc = -1, bf = 0x7fffffff;
    for (int c = tid; c < C; c += 256) {
      const float v = cav[c]; const int f = cfl[c];
      if (f != 0x7fffffff && (bc < 0 || v > bv || (v == bv && f < bf))) { bv = v; bc = c; bf = f; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {                              // wave arg-max: larger average, then the lower flat index
      const float ov = __shfl_xor(bv, o, 64);
      const int oc = __shfl_xor(bc, o, 64), of = __shfl_xor(bf, o, 64);
      if (oc >= 0 && (bc < 0 || ov > bv || (ov == bv && of < bf))) { bv = ov; bc = oc; bf = of; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bc; red_i[4 + (tid >> 6)] = bf; }
    __syncthreads();
    if (tid == 0) {
      float v0 = -__builtin_inff(); int c0 = -1, f0 = 0x7fffffff;
      for (int t = 0; t < 4; ++t) {
        const int cc = red_i[t];
        if (cc < 0) continue;
        const float v = red_v[t]; const int f = red_i[4 + t];
        if (c0 < 0 || v > v0 || (v == v0 && f < f0)) { v0 = v; c0 = cc; f0 = f; }
      }
      w_avg[r] = v0; w_flat[r] = c0 >= 0 ? f0 : 0;
      if (c0 >= 0) cfl[c0] = 0x7fffffff;
    }
    __syncthreads();
  }
  // bookkeeping: everything below reads the OLD beam state from LDS / registers before it writes the new one
  int src[8], tok[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int f = i < k ? w_flat[i] : 0;
    src[i] = f / a.V; tok[i] = f % a.V;
    if (src[i] >= n_in) src[i] = n_in - 1;
  }
  // every load of the bookkeeping goes out first (token rows, slot-table rows, embedding rows: unconditional, clamped
  // addresses), then the stores: issued phase by phase this was ~8 dependent memory round trips
  const int next_pos = first ? a.pos0 : cur_pos + 1;
  const bool has_next = next_pos < a.max_len;
  int told[MCAP], sold[MCAP];
  float e[MCAP][4], pe[4];
  {
    const int j = tid < ntok ? tid : 0;
#pragma unroll
    for (int m = 0; m < MCAP; ++m) told[m] = a.tokens[(long)(m < n_in ? m : 0) * a.ld_tokens + j];
    const int t = tid <= cur_pos ? tid : 0;
#pragma unroll
    for (int m = 0; m < MCAP; ++m) sold[m] = ld_coh<true>(a.slot_of + (long)(t < 0 ? 0 : t) * 8 + m);
    const int np = has_next ? next_pos : 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int d = tid + 256 * u < a.D ? tid + 256 * u : 0;
      pe[u] = a.wpe32[(long)np * a.D + d];
#pragma unroll
      for (int i = 0; i < MCAP; ++i) e[i][u] = a.wte32[(long)tok[i < nb ? i : 0] * a.D + d];
    }
  }
  if (tid < ntok) {                                                 // tokens = cat(tokens[next_tokens_source], next_tokens)
#pragma unroll
    for (int i = 0; i < MCAP; ++i) {
      int v = told[0];
#pragma unroll
      for (int m = 1; m < MCAP; ++m) v = src[i] == m ? told[m] : v;          // (select chain: no dynamically indexed register array)
      if (i < nb) a.tokens[(long)i * a.ld_tokens + tid] = v;
    }
  }
  if (!first && tid <= cur_pos) {                                   // cache reorder = permute the slot table
#pragma unroll
    for (int i = 0; i < MCAP; ++i) {
      int v = sold[0];
#pragma unroll
      for (int m = 1; m < MCAP; ++m) v = src[i] == m ? sold[m] : v;
      if (i < nb) st_coh<true>(a.slot_of + (long)tid * 8 + i, v);
    }
  }
  if (has_next) {
    if (tid < nb) st_coh<true>(a.slot_of + (long)next_pos * 8 + tid, tid);        // the next step appends beam b's row to slot b
#pragma unroll
    for (int i = 0; i < MCAP; ++i)                                  // next input: wte[token] + wpe[position] (D <= 1024: 4 per thread)
#pragma unroll
      for (int u = 0; u < 4; ++u) if (i < nb && tid + 256 * u < a.D) st_coh<true>(a.x + (long)i * a.D + tid + 256 * u, e[i][u] + pe[u]);
  }
  for (int j = tid + 256; j < ntok; j += 256) {                     // (prompts longer than 256 tokens)
    int old[8];
#pragma unroll
    for (int m = 0; m < 8; ++m) old[m] = m < n_in ? a.tokens[(long)m * a.ld_tokens + j] : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) if (i < nb) a.tokens[(long)i * a.ld_tokens + j] = old[src[i]];
  }
  if (!first) {
    for (int t = tid + 256; t <= cur_pos; t += 256) {               // (positions past 256)
      int old[8];
#pragma unroll
      for (int m = 0; m < 8; ++m) old[m] = ld_coh<true>(a.slot_of + (long)t * 8 + m);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int v = old[0];
#pragma unroll
        for (int m = 1; m < 8; ++m) v = src[i] == m ? old[m] : v;
        if (i < nb) st_coh<true>(a.slot_of + (long)t * 8 + i, v);
      }
    }
  }
  __syncthreads();
  if (tid == 0) {
    bool all = true;
    for (int i = 0; i < nb; ++i) {
      const float len = o_len[src[i]];
      const int st = (o_st[src[i]] != 0) || tok[i] == a.stop_token;
      if (ntok < a.ld_tokens) a.tokens[(long)i * a.ld_tokens + ntok] = tok[i];
      a.seq_len[i] = len;
      a.scores[i] = w_avg[i] * len;                                 // scores = scores_sum_average * seq_lengths
      a.stopped[i] = st;
      all = all && st;
    }
    a.state[4] = ntok + 1;
    if (all && !a.state[2]) { a.state[3] = it + 1; __hip_atomic_store(a.state + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  }
  __syncthreads();
}

// LN_f + tied lm_head over the vocabulary slice [r0, r0 + nr): xs = LN_f(x) rounded to the operand type, fp32 [nb][D] at lds;
// the slice's logits go to sl = lds + MCAP*D as [nb][BEAM_MAXR] (and to a.logits when given)
template <int MCAP>
__device__ __forceinline__ void head_phase(const BeamArgs& a, float* lds, int r0, int nr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int D = a.D, nb = a.nb;
  {
    float* xs = lds;
    float* sl = lds + MCAP * D;
    for (int m = wave; m < nb; m += 4) {                              // (D <= 1024 checked by the launcher: one read of the row)
      const float* xr = a.x + (long)m * D;
      float xv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) { const int kk = lane + 64 * u; xv[u] = ld_coh<true>(xr + (kk < D ? kk : 0)); }
      float s1 = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) if (lane + 64 * u < D) s1 += xv[u];
      const float mean = wave_sum(s1) / (float)D;
      float s2 = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) if (lane + 64 * u < D) { const float dd = xv[u] - mean; s2 += dd * dd; }
      const float rstd = rsqrtf(wave_sum(s2) / (float)D + 1e-5f);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int kk = lane + 64 * u;
        if (kk < D) xs[m * D + kk] = (float)(bf16)((xv[u] - mean) * rstd * a.lnf_w[kk] + a.lnf_b[kk]);
      }
    }
    __syncthreads();
    {
      // lane (r = lane >> 4, c = lane & 15): vocabulary row r of the wave's 4, 16-byte chunks c, c + 16, ... (D <= 1024: up to
      // 4 per lane per half); 48 rows of the slice per trip with every load of the trip in flight (8 x 16 bytes per lane
      // per half-row pass): the slice is weight-read latency, not bandwidth
      const int rr = lane >> 4, c16 = lane & 15;
      const int nch = D >> 3;                                       // 16-byte chunks per row
      for (int g0 = 0; g0 < nr; g0 += 48) {
        float acc[3][MCAP];
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int m = 0; m < MCAP; ++m) acc[h][m] = 0.f;
        bf16x8 wv[3][8];
#pragma unroll
        for (int h = 0; h < 3; ++h) {
          const int row = g0 + 16 * h + 4 * wave + rr;
          const bf16* wr = a.wte16 + (long)(r0 + (row < nr ? row : nr - 1)) * D;
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ch = c16 + 16 * u;
            wv[h][u] = *(const bf16x8*)(wr + 8 * (ch < nch ? ch : nch - 1));
          }
        }
#pragma unroll
        for (int h = 0; h < 3; ++h)
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ch = c16 + 16 * u;
            if (ch < nch) {
#pragma unroll
              for (int m = 0; m < MCAP; ++m) {
                if (m < nb) {
                  const float* xm = xs + m * D + 8 * ch;
#pragma unroll
                  for (int j = 0; j < 8; ++j) acc[h][m] += xm[j] * (float)wv[h][u][j];
                }
              }
            }
          }
#pragma unroll
        for (int h = 0; h < 3; ++h) {
          const int row = g0 + 16 * h + 4 * wave + rr;
#pragma unroll
          for (int m = 0; m < MCAP; ++m) {
            float v = acc[h][m];
            v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
            if (c16 == 0 && m < nb && row < nr) {
              sl[m * BEAM_MAXR + row] = v;
              if (a.logits) a.logits[(long)m * a.ld_logits + r0 + row] = v;
            }
          }
        }
      }
    }
  }
}

// one projection phase: this workgroup's column blocks
template <int MCAP, int ACT, int U>
__device__ __forceinline__ void proj_phase(const GemmArgs& p, int nblk, int G, float* lds, PhaseSync* ps) {
  auto pw = [&]() { ps->wait(); };           // (waiting twice for the same total is free: the second call returns at once)
  for (int cb = blockIdx.x; cb < nblk; cb += G) skinny_block<MCAP, ACT, U, true>(p, cb * 32, lds, pw);
}

template <int MCAP>
__global__ __launch_bounds__(256) void gpt2_beam_persist_kernel(const BeamArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = gridDim.x, D = a.D, Hd = a.Hd, nb = a.nb;
  PhaseSync ps{a.state, a.state + 1, 0, false};
  auto nprod = [&](int nblk) { return nblk < G ? nblk : G; };
  const long ldrow = 5L * D + Hd;
  bf16* qkv = a.scratch + D;
  bf16* att = a.scratch + 4L * D;
  bf16* hid = a.scratch + 5L * D;
  const float T = a.temperature > 0.f ? a.temperature : 1.0f;
  const int R = a.rows_per_wg;
  const int r0 = blockIdx.x * R;
  const int nr = r0 >= a.V ? 0 : (a.V - r0 < R ? a.V - r0 : R);
  int it = 0;
  const int ntok0 = a.state[4];                                    // token columns present at launch; one more per selection
  if (a.first) {
    // the prefill's last-position logits: selection with one input beam (test.py:396-405)
    for (int j = tid; j < nr; j += 256) lds[j] = a.first_logits[r0 + j];
    __syncthreads();
    select_partials(a, lds, 1, r0, nr, T);
    ps.arrive(G, true);
    if (blockIdx.x == 0) { ps.wait(); select_merge<MCAP>(a, 1, true, it, a.pos0 - 1, G, ntok0 + it, lds); }
    ps.arrive(1, blockIdx.x == 0);
    ++it;
  }
  for (int s = 0; s < a.n_steps; ++s, ++it) {
    ps.wait();                                                      // the selection: next input rows, slot table, the stop flag
    if (__hip_atomic_load(a.state + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;     // every beam has stopped
    const int pos = a.pos0 + s;
    if (pos >= a.max_len) break;
    GemmArgs p;
    p.alpha = 1.0f; p.aux = nullptr; p.ldaux = 0; p.out_pre = nullptr; p.split_ws = nullptr; p.ktiles_per_split = 0; p.M = nb;
    for (int l = 0; l < a.n_layer; ++l) {
      const cclip_block_ptrs& w = a.blocks[l];
      bf16* kc = a.kc + (long)l * a.ld_layer;
      bf16* vc = a.vc + (long)l * a.ld_layer;
      // P1: LayerNorm + qkv projection, k / v appended at `pos` of each beam's own slot
      p.A = nullptr; p.lda = 0; p.B = (const bf16*)w.w_qkv; p.ldb = 3 * D; p.N = 3 * D; p.K = D; p.bias = w.b_qkv; p.act = 0;
      p.residual = nullptr; p.ldr = 0; p.out_f32 = nullptr; p.out_bf16 = qkv; p.ldc = ldrow;
      p.ln_x = a.x; p.ln_ldx = D; p.ln_gamma = w.ln1_w; p.ln_beta = w.ln1_b;
      p.kv_k = kc + (long)pos * D; p.kv_v = vc + (long)pos * D; p.kv_ld_seq = a.ld_seq; p.kv_width = D;
      {
        const int nblk = (3 * D + 31) / 32;
        proj_phase<MCAP, CCLIP_ACT_NONE, 12>(p, nblk, G, lds, &ps);
        ps.arrive(nprod(nblk), blockIdx.x < nblk);
      }
      // P2: attention of the new token against positions [0, pos]
      const int ntask4 = (nb * a.H + 3) / 4;
      if (blockIdx.x < ntask4) ps.wait();
      for (int t0 = blockIdx.x * 4; t0 < nb * a.H; t0 += G * 4) {
        const int t = t0 + wave;
        const bool valid = t < nb * a.H;
        const int tt = valid ? t : nb * a.H - 1;
        float* wl = lds + wave * (128 + a.max_len + 64);           // per wave: 128 probabilities, max_len row offsets, 64 q
        attn_task(a, qkv, ldrow, kc, vc, att, ldrow, tt / a.H, tt % a.H, pos + 1, valid, wl, (int*)(wl + 128), wl + 128 + a.max_len);
      }
      ps.arrive(nprod(ntask4), blockIdx.x < ntask4);
      // P3: out-proj + residual (x += ...)
      p.ln_x = nullptr; p.kv_k = nullptr; p.kv_v = nullptr; p.kv_width = 0;
      p.A = att; p.lda = ldrow; p.B = (const bf16*)w.w_o; p.ldb = D; p.N = D; p.K = D; p.bias = w.b_o;
      p.residual = a.x; p.ldr = D; p.out_f32 = a.x; p.out_bf16 = nullptr; p.ldc = D;
      {
        const int nblk = (D + 31) / 32;
        proj_phase<MCAP, CCLIP_ACT_NONE, 12>(p, nblk, G, lds, &ps);
        ps.arrive(nprod(nblk), blockIdx.x < nblk);
      }
      // P4: LayerNorm + fc + activation
      p.A = nullptr; p.lda = 0; p.B = (const bf16*)w.w_fc; p.ldb = Hd; p.N = Hd; p.K = D; p.bias = w.b_fc;
      p.residual = nullptr; p.ldr = 0; p.out_f32 = nullptr; p.out_bf16 = hid; p.ldc = ldrow;
      p.ln_x = a.x; p.ln_ldx = D; p.ln_gamma = w.ln2_w; p.ln_beta = w.ln2_b;
      {
        const int nblk = (Hd + 31) / 32;
        if (a.act == CCLIP_ACT_GELU_NEW) proj_phase<MCAP, CCLIP_ACT_GELU_NEW, 12>(p, nblk, G, lds, &ps);
        else proj_phase<MCAP, CCLIP_ACT_NONE, 12>(p, nblk, G, lds, &ps);
        ps.arrive(nprod(nblk), blockIdx.x < nblk);
      }
      // P5: proj + residual
      p.ln_x = nullptr;
      p.A = hid; p.lda = ldrow; p.B = (const bf16*)w.w_proj; p.ldb = D; p.N = D; p.K = Hd; p.bias = w.b_proj;
      p.residual = a.x; p.ldr = D; p.out_f32 = a.x; p.out_bf16 = nullptr; p.ldc = D;
      {
        const int nblk = (D + 31) / 32;
        proj_phase<MCAP, CCLIP_ACT_NONE, 16>(p, nblk, G, lds, &ps);
        ps.arrive(nprod(nblk), blockIdx.x < nblk);
      }
    }
    // LN_f + tied lm_head over this workgroup's vocabulary slice [r0, r0 + nr), then the slice's selection partials.
    // xs: LN_f(x) rounded to the operand type, fp32 [nb][D]; sl: the slice's logits [nb][BEAM_MAXR]
    ps.wait();
    float* sl = lds + MCAP * D;
    head_phase<MCAP>(a, lds, r0, nr);
    __syncthreads();
    select_partials(a, sl, nb, r0, nr, T);
    ps.arrive(G, true);
    if (blockIdx.x == 0) { ps.wait(); select_merge<MCAP>(a, nb, false, it, pos, G, ntok0 + it, lds); }
    ps.arrive(1, blockIdx.x == 0);
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

extern "C" int CCLIP_FN(cclip_gpt2_beam_search)(const cclip_beam_desc* d, hipStream_t stream) {
  if (!d || !d->step.blocks || !d->step.x || !d->step.kcache || !d->step.vcache || !d->step.scratch16) return CCLIP_ERR_ARG;
  const cclip_decode_desc& s = d->step;
  if (s.n_layer <= 0 || s.n_layer > BEAM_MAXL || s.n_seq <= 0 || s.n_seq > 8 || s.linear_layout) return CCLIP_ERR_ARG;
  if (s.width <= 0 || (s.width & 63) || s.width > 1024 || s.width != s.heads * 64 || s.hidden <= 0 || (s.hidden & 31) || s.pos < 0) return CCLIP_ERR_ARG;
  if (s.act != CCLIP_ACT_NONE && s.act != CCLIP_ACT_GELU_NEW) return CCLIP_ERR_ARG;
  if (!s.lnf_w || !s.lnf_b || !s.wte16 || s.vocab <= 0 || (s.ld_seq & 7) || (s.ld_layer & 7)) return CCLIP_ERR_ARG;
  if (s.logits && (s.ld_logits < s.vocab)) return CCLIP_ERR_ARG;
  if (!d->wte_f32 || !d->wpe_f32 || !d->slot_of || !d->tokens || !d->scores || !d->seq_lengths || !d->is_stopped || !d->state || !d->select_ws)
    return CCLIP_ERR_ARG;
  if (d->n_steps < 0 || d->max_len <= 0 || d->max_len > 2048 || d->ld_tokens <= 0 || (d->first && !d->first_logits)) return CCLIP_ERR_ARG;
  if (!d->first && d->n_steps == 0) return CCLIP_OK;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0; hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return CCLIP_ERR_LAUNCH;
    n_cu = prop.multiProcessorCount;
  }
  // at most one workgroup per CU (every workgroup resident: the hand-overs can fill).  The projection phases have 24-96 column
  // blocks, so more workgroups than that only add pollers at the step boundaries: 96, or as many as the vocabulary slices need
  // (measured on GPT-2-small, V = 21128: 0.685 ms / step with 96 workgroups, 0.727 with 256)
  int G = (s.vocab + 223) / 224; if (G < 96) G = 96;
  if (G > 256) G = 256;
  if (G > n_cu) G = n_cu;
  if (d->grid_cap > 0 && d->grid_cap < G) G = d->grid_cap;
  int R = (s.vocab + G - 1) / G; R = (R + 31) / 32 * 32;
  if (R > BEAM_MAXR) return CCLIP_ERR_ARG;                      // vocabulary too large for one slice per workgroup
  const int mcap = s.n_seq <= 4 ? 4 : 8;
  // LDS: GEMV A rows / reduction, attention rows, lm_head rows + slice, selection candidates
  size_t fl = (size_t)s.n_seq * s.hidden;
  const size_t red = (size_t)256 * mcap * 8; if (red > fl) fl = red;
  const size_t att = (size_t)4 * (128 + d->max_len + 64); if (att > fl) fl = att;
  const size_t head = (size_t)mcap * s.width + (size_t)s.n_seq * BEAM_MAXR; if (head > fl) fl = head;
  const size_t sel = 64 + 512 + 4096 + 2 * (size_t)s.n_seq * G * s.n_seq; if (sel > fl) fl = sel;
  const size_t lds = fl * sizeof(float);
  if (lds > 150 * 1024) return CCLIP_ERR_ARG;
  BeamArgs a;
  a.n_layer = s.n_layer; a.nb = s.n_seq; a.D = s.width; a.H = s.heads; a.Hd = s.hidden; a.act = s.act; a.V = s.vocab; a.pos0 = s.pos;
  a.n_steps = d->n_steps; a.first = d->first ? 1 : 0; a.stop_token = d->stop_token; a.ld_tokens = d->ld_tokens; a.max_len = d->max_len;
  a.rows_per_wg = R; a.temperature = d->temperature;
  for (int l = 0; l < s.n_layer; ++l) a.blocks[l] = s.blocks[l];
  a.x = s.x; a.kc = (bf16*)s.kcache; a.vc = (bf16*)s.vcache; a.ld_layer = s.ld_layer; a.ld_seq = s.ld_seq; a.scratch = (bf16*)s.scratch16;
  a.lnf_w = s.lnf_w; a.lnf_b = s.lnf_b; a.wte16 = (const bf16*)s.wte16; a.logits = s.logits; a.ld_logits = s.ld_logits;
  a.first_logits = d->first_logits; a.wte32 = d->wte_f32; a.wpe32 = d->wpe_f32;
  a.slot_of = d->slot_of; a.tokens = d->tokens; a.scores = d->scores; a.seq_len = d->seq_lengths; a.stopped = d->is_stopped;
  a.state = d->state; a.part = d->select_ws;
  if (hipMemsetAsync(d->state, 0, 2 * sizeof(int), stream) != hipSuccess) return CCLIP_ERR_LAUNCH;   // barrier counter, error flag
#define BEAM_LAUNCH(MC)                                                                                                         \
  do {                                                                                                                          \
    static size_t attr = 0;                                                                                                     \
    if (lds > attr) {                                                                                                           \
      if (hipFuncSetAttribute((const void*)gpt2_beam_persist_kernel<MC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return CCLIP_ERR_LAUNCH;                                                                                                \
      attr = lds;                                                                                                               \
    }                                                                                                                           \
    hipLaunchKernelGGL((gpt2_beam_persist_kernel<MC>), dim3(G), dim3(256), lds, stream, a);                                      \
  } while (0)
  if (mcap == 4) BEAM_LAUNCH(4); else BEAM_LAUNCH(8);
#undef BEAM_LAUNCH
  return cclip_launch_status();
}
