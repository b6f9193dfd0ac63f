// Fused multi-head attention (dh = 64) for gfx950, forward and backward: one workgroup per (batch, head) for T <= 128,
// key-block-tiled online softmax forward and a two-kernel (dK/dV by key block, dQ by query block) backward beyond.
//
// Replaces the scaled-dot-product core of nn.MultiheadAttention inside the `clip` package's
// ResidualAttentionBlock (image tower: T=50 no mask; text tower: T=77 additive causal mask),
// reached from /root/reference/CLIP/train.py:161, and GPT-2's causal + key-padding attention
// behind /root/reference/CLIP_prefix_caption/train.py:268.  The T x T score matrix never
// touches HBM; forward keeps only the per-row log-sum-exp for backward.
//
// One workgroup (4 waves) per (batch, head).  K and V of the head live in LDS (<= 32 KiB), every
// wave owns 16-query tiles.  All products are v_mfma_f32_16x16x32_bf16:
//   forward  S^T = K Q^T   (keys on accumulator rows, query on the lane): row max / sum are
//            in-lane + 2 shuffles, and the exponentiated accumulator IS the B operand of
//            O^T = V^T P^T (contraction over its row index) - P never goes through LDS.
//            V^T fragments come from ds_read_b64_tr_b16 (hardware transpose).
//   backward S = Q K^T, dP = dO V^T (query on accumulator rows): P and dS are the B operands of
//            dV^T = dO^T P and dK^T = Q^T dS with no data movement; only dS crosses LDS once,
//            for dQ^T = K^T dS^T.  Each wave owns key tiles, so dK/dV need no cross-wave sums.
// LDS images are [row][64] bf16 with 128-byte rows, 16-byte chunk c of row r stored at
// chunk c ^ (r & 7): conflict-free for both ds_read_b128 row reads and the transposed reads.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

struct AttnArgs {
  const bf16* q; const bf16* k; const bf16* v;   // row (b*T + t), head h at column h*64
  long ldq, ldk, ldv;
  bf16* o; long ldo;
  float* lse;                                     // [B, H, T]
  const float* keep;                              // [B, T] 1 = attend, 0 = masked key (or null)
  int B, T, H, causal;
  float scale;
  // backward only
  const bf16* dout; long lddo;
  bf16* dq; bf16* dk; bf16* dv; long lddq, lddk, lddv;
  // forward, fp8 inference path: o8 != null -> the output goes out as e4m3 + E8M0 block scales instead of 16-bit (o is not written)
  unsigned char* o8; long ldo8; unsigned char* omx; long ldomx;
  // packed (variable-length) batches, T <= 128 kernels: sequence b occupies rows [cu[b], cu[b+1]) and T is the longest length
  // (tile-count dispatch, lse row stride); null: row b*T + t
  const int* cu;
};

// Output of one query row of one head as the block-scaled A operand of the out-proj GEMM (gemm_bf16_fp8ops.hip: e4m3, one E8M0
// exponent per 32 columns, scale layout [columns / 128][rows][4]): a head's 64 dims are two blocks - dims 0..31 sit in o[0], o[1]
// of the four lanes (row, g = 0..3), dims 32..63 in o[2], o[3].  Every lane of the wave calls (cross-lane amax); `live` guards
// the stores.
__device__ __forceinline__ void attn_store_mx(const AttnArgs& a, long row, int h, int g, const f32x4 (&o)[4], float inv, bool live) {
  float am[2] = {0.f, 0.f};
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int j = 0; j < 4; ++j) am[dt >> 1] = fmaxf(am[dt >> 1], fabsf(o[dt][j] * inv));
  int e[2];
#pragma unroll
  for (int hb = 0; hb < 2; ++hb) {
    float m = fmaxf(am[hb], __shfl_xor(am[hb], 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const unsigned bits = __float_as_uint(m * (1.0f / 448.0f));          // (e8m0_of of gemm_bf16_fp8ops.hip)
    int ee = (int)(bits >> 23) + ((bits & 0x7FFFFFu) ? 1 : 0);
    ee = ee < 1 ? 1 : ee;
    e[hb] = ee > 253 ? 253 : ee;
  }
  if (!live) return;
  unsigned char* op = a.o8 + row * a.ldo8 + h * 64 + 4 * g;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const float sc = inv * __uint_as_float((unsigned)(254 - e[dt >> 1]) << 23);
    float f[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = fminf(fmaxf(o[dt][j] * sc, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
    *(int*)(op + 16 * dt) = w;
  }
  if (g == 0) *(unsigned short*)(a.omx + (long)(h >> 1) * a.ldomx + 4 * row + ((2 * h) & 3)) = (unsigned short)(e[0] | (e[1] << 8));
}

__device__ __forceinline__ int at_off(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// A-operand fragment of X^T (16 columns d0..d0+15 as MFMA rows) over 8 rows given as two 4-row blocks
__device__ __forceinline__ bf16x8 frag_tr(const char* img, int rowblk0, int rowblk1, int dt, int lane) {
  const int q = (lane >> 2) & 3, p = lane & 3;
  const int chunk = 2 * dt + (p >> 1), sub = 8 * (p & 1);
  const bf16x4 lo = lds_read_tr16(img + at_off(rowblk0 + q, chunk) + sub);
  const bf16x4 hi = lds_read_tr16(img + at_off(rowblk1 + q, chunk) + sub);
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
__device__ __forceinline__ bf16x8 frag_row(const char* img, int row, int chunk) {
  return *(const bf16x8*)(img + at_off(row, chunk));
}

// Staging of a [T][64] head slice into LDS in two halves: head_load issues every 16-byte global load of the slice (IT per
// thread, rows clamped so that no load is predicated) and head_store writes them to the swizzled image, zeroing rows >= T.
// All loads of all operands go out before the first wait: a predicated load -> wait -> ds_write loop costs one HBM round
// trip per iteration (8-10 serial round trips were most of a T=50 workgroup's life, rocprofv3 + ISA).
template <int IT>
__device__ __forceinline__ void head_load(const bf16* g, long ld, long row0, int T, uint4 (&r)[IT], int tid) {
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = tid + 256 * it, row = idx >> 3, c = idx & 7;
    r[it] = *(const uint4*)(g + (row0 + (row < T ? row : T - 1)) * ld + c * 8);
  }
}
template <int IT>
__device__ __forceinline__ void head_store(char* img, int T, int rows_total, const uint4 (&r)[IT], int tid) {
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int idx = tid + 256 * it, row = idx >> 3, c = idx & 7;
    if (row < rows_total) *(uint4*)(img + at_off(row, c)) = row < T ? r[it] : make_uint4(0, 0, 0, 0);
  }
}

template <int NKT>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
  constexpr int NKS = (NKT + 1) / 2, TP = 16 * NKT, TP32 = 32 * NKS;
  __shared__ __attribute__((aligned(16))) char smem[(TP + TP32) * 128 + TP32 * 4];
  char* Ks = smem;
  char* Vs = smem + TP * 128;
  float* keep_s = (float*)(smem + (TP + TP32) * 128);      // 1 = key exists and is not padding-masked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
  const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
  const int nqt = (T + 15) >> 4;
  bf16x8 qn0, qn1;                                         // this wave's first query tile rides along with K and V
  {
    uint4 rk[NKS], rv[NKS];
    head_load<NKS>(a.k + h * 64, a.ldk, row0, T, rk, tid);
    head_load<NKS>(a.v + h * 64, a.ldv, row0, T, rv, tid);
    const int qi = 16 * wave + li;
    const bf16* qp = a.q + (row0 + (qi < T ? qi : T - 1)) * a.ldq + h * 64 + 8 * g;
    qn0 = *(const bf16x8*)qp; qn1 = *(const bf16x8*)(qp + 32);
    if (tid < TP32) keep_s[tid] = (tid < T && (!a.keep || a.keep[row0 + tid] != 0.f)) ? 1.f : 0.f;
    head_store<NKS>(Ks, T, TP, rk, tid);
    head_store<NKS>(Vs, T, TP32, rv, tid);
  }
  __syncthreads();
  const float NEG = -__builtin_inff();
  for (int qt = wave; qt < nqt; qt += 4) {
    const int qi = 16 * qt + li;
    const bf16x8 qf0 = qn0, qf1 = qn1;
    if (qt + 4 < nqt) {                                    // T > 64 only: the next tile's rows load under this tile's math
      const int qx = qi + 64;
      const bf16* qp = a.q + (row0 + (qx < T ? qx : T - 1)) * a.ldq + h * 64 + 8 * g;
      qn0 = *(const bf16x8*)qp; qn1 = *(const bf16x8*)(qp + 32);
    }
    int ktmax = NKT - 1;
    if (a.causal && qt < ktmax) ktmax = qt;
    f32x4 s[NKT];
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (kt <= ktmax) {
        s[kt] = CCLIP_MFMA_16x16x32(frag_row(Ks, 16 * kt + li, g), qf0, s[kt]);
        s[kt] = CCLIP_MFMA_16x16x32(frag_row(Ks, 16 * kt + li, 4 + g), qf1, s[kt]);
      }
    }
    float m = NEG;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const float4 kp4 = *(const float4*)(keep_s + 16 * kt + 4 * g);
      const float kp[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = 16 * kt + 4 * g + r;
        const bool ok = kt <= ktmax && kp[r] != 0.f && (!a.causal || key <= qi);
        const float val = ok ? s[kt][r] * a.scale : NEG;
        s[kt][r] = val;
        m = fmaxf(m, val);
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    const float msafe = m == NEG ? 0.f : m;
    float l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = __expf(s[kt][r] - msafe);
        s[kt][r] = pv;
        l += pv;
      }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < NKS; ++ss) {
      if (2 * ss <= ktmax) {
        bf16x8 pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pf[j] = (bf16)s[2 * ss][j];
          pf[4 + j] = (2 * ss + 1 < NKT) ? (bf16)s[(2 * ss + 1 < NKT) ? 2 * ss + 1 : 0][j] : (bf16)0.f;
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          o[dt] = CCLIP_MFMA_16x16x32(frag_tr(Vs, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane), pf, o[dt]);
      }
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (a.o8) {
      attn_store_mx(a, row0 + (qi < T ? qi : 0), h, g, o, inv, qi < T);
    } else if (qi < T) {
      bf16* op = a.o + (row0 + qi) * a.ldo + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(bf16)(o[dt][0] * inv), (bf16)(o[dt][1] * inv), (bf16)(o[dt][2] * inv), (bf16)(o[dt][3] * inv)};
        *(bf16x4*)(op + 16 * dt) = ov;
      }
      if (g == 0 && a.lse) a.lse[((long)b * a.H + h) * a.T + qi] = msafe + __logf(l);
    }
  }
}

// ---------------------------------------------------------------------------------------------
template <int NKT>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const AttnArgs a) {
  constexpr int NKS = (NKT + 1) / 2, TP32 = 32 * NKS;
  constexpr int DS_LD = 2 * TP32 + 16;                     // padded dS row stride (bytes)
  __shared__ __attribute__((aligned(16))) char smem[4 * TP32 * 128 + TP32 * DS_LD + 3 * TP32 * 4];
  char* Qs = smem;
  char* Ks = Qs + TP32 * 128;
  char* Vs = Ks + TP32 * 128;
  char* Os = Vs + TP32 * 128;                              // dO
  char* dSs = Os + TP32 * 128;                             // [q][key] bf16
  float* lse_s = (float*)(dSs + TP32 * DS_LD);
  float* del_s = lse_s + TP32;
  float* keep_s = del_s + TP32;                            // 1 = key exists and is not padding-masked
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const int T = a.cu ? a.cu[b + 1] - a.cu[b] : a.T;
  const long row0 = a.cu ? (long)a.cu[b] : (long)b * a.T;
  {
    uint4 rq[NKS], rk[NKS], rv[NKS], rdo[NKS], ro[NKS];    // TP32 * 8 chunks = 256 * NKS: NKS per thread and operand
    head_load<NKS>(a.q + h * 64, a.ldq, row0, T, rq, tid);
    head_load<NKS>(a.k + h * 64, a.ldk, row0, T, rk, tid);
    head_load<NKS>(a.v + h * 64, a.ldv, row0, T, rv, tid);
    head_load<NKS>(a.dout + h * 64, a.lddo, row0, T, rdo, tid);
    head_load<NKS>(a.o + h * 64, a.ldo, row0, T, ro, tid);
    float lse_r = 1e30f;                                   // rows >= T: P = exp(s - 1e30) = 0
    if (tid < T) lse_r = a.lse[((long)b * a.H + h) * a.T + tid];
    const float keep_r = (tid < T && (!a.keep || a.keep[row0 + tid] != 0.f)) ? 1.f : 0.f;
    // zero dS (tiles skipped by the causal structure are read as zeros) while the loads fly
    for (int i = tid; i < TP32 * DS_LD / 16; i += 256) *(uint4*)(dSs + i * 16) = make_uint4(0, 0, 0, 0);
    head_store<NKS>(Qs, T, TP32, rq, tid);
    head_store<NKS>(Ks, T, TP32, rk, tid);
    head_store<NKS>(Vs, T, TP32, rv, tid);
    head_store<NKS>(Os, T, TP32, rdo, tid);
    // delta = rowsum(dO * O): the 8 threads that hold a row's chunks each dot their 8 elements
#pragma unroll
    for (int it = 0; it < NKS; ++it) {
      const int row = (tid >> 3) + 32 * it;
      const bf16x8 ov = __builtin_bit_cast(bf16x8, ro[it]), dv = __builtin_bit_cast(bf16x8, rdo[it]);
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += (float)ov[j] * (float)dv[j];
      acc += __shfl_xor(acc, 1, 64);
      acc += __shfl_xor(acc, 2, 64);
      acc += __shfl_xor(acc, 4, 64);
      if ((tid & 7) == 0) del_s[row] = row < T ? acc : 0.f;
    }
    if (tid < TP32) { lse_s[tid] = lse_r; keep_s[tid] = keep_r; }
  }
  __syncthreads();
  const int nkt = (T + 15) >> 4;                           // live key / query tiles
  const float NEG = -__builtin_inff();

  // ---- phase A: this wave's key tiles; loop over query-tile pairs ----
  for (int kt = wave; kt < nkt; kt += 4) {
    f32x4 dvT[4], dkT[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dvT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int key = 16 * kt + li;
    const bool key_ok = keep_s[key] != 0.f;
    const bf16x8 kf0 = frag_row(Ks, key, g), kf1 = frag_row(Ks, key, 4 + g);
    const bf16x8 vf0 = frag_row(Vs, key, g), vf1 = frag_row(Vs, key, 4 + g);
    for (int ss = 0; ss < NKS; ++ss) {
      if (a.causal && 2 * ss + 1 < kt) continue;           // both query tiles entirely above the diagonal
      bf16x8 pf, dsf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * ss + half;
        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int qrow = 16 * qt + li;                     // A-operand row of this lane (row read)
        sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, g), kf0, sv);
        sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, 4 + g), kf1, sv);
        dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, g), vf0, dp);
        dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, 4 + g), vf1, dp);
        const float4 ls = *(const float4*)(lse_s + 16 * qt + 4 * g);
        const float4 de = *(const float4*)(del_s + 16 * qt + 4 * g);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = 16 * qt + 4 * g + r;
          const bool ok = key_ok && (!a.causal || key <= qi);
          const float pv = ok ? __expf(sv[r] * a.scale - lsv[r]) : 0.f;
          const float dsv = pv * (dp[r] - dev[r]) * a.scale;
          pf[4 * half + r] = (bf16)pv;
          dsf[4 * half + r] = (bf16)dsv;
          *(bf16*)(dSs + qi * DS_LD + key * 2) = (bf16)dsv;
        }
      }
      // contraction over the 32 queries of this pair: k index = (half, g, r) on both operands
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dvT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Os, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane), pf, dvT[dt]);
        dkT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Qs, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane), dsf, dkT[dt]);
      }
    }
    if (key < T) {
      bf16* dvp = a.dv + (row0 + key) * a.lddv + h * 64 + 4 * g;
      bf16* dkp = a.dk + (row0 + key) * a.lddk + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 x = {(bf16)dvT[dt][0], (bf16)dvT[dt][1], (bf16)dvT[dt][2], (bf16)dvT[dt][3]};
        bf16x4 y = {(bf16)dkT[dt][0], (bf16)dkT[dt][1], (bf16)dkT[dt][2], (bf16)dkT[dt][3]};
        *(bf16x4*)(dvp + 16 * dt) = x;
        *(bf16x4*)(dkp + 16 * dt) = y;
      }
    }
  }
  __syncthreads();
  // ---- phase B: dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]; this wave's query tiles ----
  for (int qt = wave; qt < nkt; qt += 4) {
    f32x4 dqT[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dqT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int qi = 16 * qt + li;
    for (int ss = 0; ss < NKS; ++ss) {
      if (a.causal && 32 * ss > 16 * qt + 15) continue;
      const bf16x8 dsf = *(const bf16x8*)(dSs + qi * DS_LD + (32 * ss + 8 * g) * 2);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        dqT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Ks, 32 * ss + 8 * g, 32 * ss + 8 * g + 4, dt, lane), dsf, dqT[dt]);
    }
    if (qi < T) {
      bf16* dqp = a.dq + (row0 + qi) * a.lddq + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 x = {(bf16)dqT[dt][0], (bf16)dqT[dt][1], (bf16)dqT[dt][2], (bf16)dqT[dt][3]};
        *(bf16x4*)(dqp + 16 * dt) = x;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Forward for sequences longer than 128 (ViT-B/16: 197, ViT-L/14: 257, ViT-L/14@336px: 577 tokens): same S^T = K Q^T /
// O^T = V^T P^T formulation, tiled over 64-key blocks with the online-softmax recurrence.  One workgroup per
// (batch, head, 128-query block); wave w owns the two 16-query tiles 32w.. of the block (every K / V fragment read from
// LDS feeds two MFMAs); K / V blocks are double-buffered in LDS and the next block's global loads are in flight (in
// registers) while the current block is multiplied - one barrier per block.  Causal workgroups stop at their diagonal.
#define ALQ 2                                             // query tiles per wave
__global__ __launch_bounds__(256) void attn_long_fwd_kernel(const AttnArgs a, int nqb) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 64 * 128];          // [buffer][K | V][64 rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int b = bh / a.H, h = bh % a.H;
  const int T = a.T;
  const long row0 = (long)b * T;
  const float NEG = -__builtin_inff();
  int qi[ALQ];
  bf16x8 qf0[ALQ], qf1[ALQ];
  float m[ALQ], l[ALQ];
  f32x4 o[ALQ][4];
#pragma unroll
  for (int t = 0; t < ALQ; ++t) {
    qi[t] = 128 * qb + 32 * wave + 16 * t + li;
    const int qrow = qi[t] < T ? qi[t] : T - 1;
    const bf16* qp = a.q + (row0 + qrow) * a.ldq + h * 64 + 8 * g;
    qf0[t] = *(const bf16x8*)qp; qf1[t] = *(const bf16x8*)(qp + 32);
    m[t] = NEG; l[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  int nkb = (T + 63) >> 6;
  if (a.causal) { const int last = (128 * qb + 127) >> 6; if (last + 1 < nkb) nkb = last + 1; }
  // this thread's share of a block: rows r0, r0 + 32 of K and of V, 16-byte chunk c
  const int r0 = tid >> 3, c = tid & 7;
  uint4 pk[2], pv[2];
  auto gload = [&](int kb) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int key = 64 * kb + r0 + 32 * i;
      pk[i] = make_uint4(0, 0, 0, 0); pv[i] = make_uint4(0, 0, 0, 0);
      if (key < T) {
        pk[i] = *(const uint4*)(a.k + (row0 + key) * a.ldk + h * 64 + c * 8);
        pv[i] = *(const uint4*)(a.v + (row0 + key) * a.ldv + h * 64 + c * 8);
      }
    }
  };
  auto lstore = [&](int buf) {
    char* Ks = smem + buf * (2 * 64 * 128);
    char* Vs = Ks + 64 * 128;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(uint4*)(Ks + at_off(r0 + 32 * i, c)) = pk[i];
      *(uint4*)(Vs + at_off(r0 + 32 * i, c)) = pv[i];
    }
  };
  gload(0);
  lstore(0);
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                   // block kb is in buffer kb & 1; everyone is done with buffer (kb+1) & 1
    if (kb + 1 < nkb) gload(kb + 1);                   // in flight during this block's MFMAs
    const char* Ks = smem + (kb & 1) * (2 * 64 * 128);
    const char* Vs = Ks + 64 * 128;
    f32x4 s[ALQ][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const bf16x8 k0 = frag_row(Ks, 16 * kt + li, g), k1 = frag_row(Ks, 16 * kt + li, 4 + g);
#pragma unroll
      for (int t = 0; t < ALQ; ++t) {
        s[t][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        s[t][kt] = CCLIP_MFMA_16x16x32(k0, qf0[t], s[t][kt]);
        s[t][kt] = CCLIP_MFMA_16x16x32(k1, qf1[t], s[t][kt]);
      }
    }
    // The softmax arithmetic, not the MFMAs, bounds this kernel at head_dim 64 (~10 VALU slots per score against 0.5 MFMA
    // slots): scores go to the log2 domain with ONE multiply (scale * log2 e), exponentials are bare v_exp_f32, and blocks
    // that need no masking (all but the last key block, no key padding, below the causal diagonal) skip the per-score tests.
    const float sl2 = a.scale * 1.4426950408889634f;
    const bool full = 64 * kb + 64 <= T && !a.keep && (!a.causal || 64 * kb + 63 <= 128 * qb + 32 * wave);
    float alpha[ALQ];
#pragma unroll
    for (int t = 0; t < ALQ; ++t) {
      float bm = NEG;
      if (full) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s[t][kt][r] *= sl2;
            bm = fmaxf(bm, s[t][kt][r]);
          }
      } else {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = 64 * kb + 16 * kt + 4 * g + r;
            bool ok = key < T && (!a.causal || key <= qi[t]);
            if (ok && a.keep) ok = a.keep[row0 + key] != 0.f;
            const float val = ok ? s[t][kt][r] * sl2 : NEG;
            s[t][kt][r] = val;
            bm = fmaxf(bm, val);
          }
      }
      bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
      bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
      const float mn = fmaxf(m[t], bm);                 // running maximum, log2 domain
      const float msafe = mn == NEG ? 0.f : mn;
      alpha[t] = __builtin_amdgcn_exp2f(m[t] - msafe);  // m = -inf (nothing seen yet) -> 0
      float bl = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv_ = __builtin_amdgcn_exp2f(s[t][kt][r] - msafe);
          s[t][kt][r] = pv_;
          bl += pv_;
        }
      bl += __shfl_xor(bl, 16, 64);
      bl += __shfl_xor(bl, 32, 64);
      l[t] = l[t] * alpha[t] + bl;
      m[t] = mn;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) o[t][dt][r] *= alpha[t];
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 pf[ALQ];
#pragma unroll
      for (int t = 0; t < ALQ; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) { pf[t][j] = (bf16)s[t][2 * ss][j]; pf[t][4 + j] = (bf16)s[t][2 * ss + 1][j]; }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = frag_tr(Vs, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane);
#pragma unroll
        for (int t = 0; t < ALQ; ++t) o[t][dt] = CCLIP_MFMA_16x16x32(vf, pf[t], o[t][dt]);
      }
    }
    if (kb + 1 < nkb) lstore((kb + 1) & 1);            // its readers (block kb-1) all passed this iteration's barrier
  }
#pragma unroll
  for (int t = 0; t < ALQ; ++t) {
    if (a.o8) {
      attn_store_mx(a, row0 + (qi[t] < T ? qi[t] : 0), h, g, o[t], l[t] > 0.f ? 1.0f / l[t] : 0.f, qi[t] < T);
    } else if (qi[t] < T) {
      const float inv = l[t] > 0.f ? 1.0f / l[t] : 0.f;
      bf16* op = a.o + (row0 + qi[t]) * a.ldo + h * 64 + 4 * g;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        bf16x4 ov = {(bf16)(o[t][dt][0] * inv), (bf16)(o[t][dt][1] * inv), (bf16)(o[t][dt][2] * inv), (bf16)(o[t][dt][3] * inv)};
        *(bf16x4*)(op + 16 * dt) = ov;
      }
      if (g == 0 && a.lse) a.lse[((long)b * a.H + h) * T + qi[t]] = (m[t] == NEG ? 0.f : m[t] * 0.6931471805599453f) + __logf(l[t]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Backward for sequences longer than 128: the single-workgroup kernel above, cut along the key axis (dK, dV) and along
// the query axis (dQ).  Both recompute S and dP from the saved log-sum-exp, as the short kernel does; delta = rowsum(dO*O)
// is recomputed per staged query block.  Same operand roles as attn_bwd_kernel: query on accumulator rows, so P and dS
// are B operands of dV^T = dO^T P and dK^T = Q^T dS without data movement; dS crosses LDS once for dQ^T = K^T dS^T.
//
// Staging of 64 rows [r0, r0+64) of a [T][64] head slice, split as in the short kernels: blk_load requests the thread's
// two 16-byte chunks (rows clamped: no predicated load), blk_store writes them to the swizzled image (zero rows >= T).
// Between the two a kernel keeps the NEXT block's chunks in registers while the current block is multiplied.
__device__ __forceinline__ void blk_load(const bf16* g, long ld, long row0, int r0, int T, uint4 (&r)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = r0 + (tid >> 3) + 32 * i;
    r[i] = *(const uint4*)(g + (row0 + (row < T ? row : T - 1)) * ld + (tid & 7) * 8);
  }
}
__device__ __forceinline__ void blk_store(char* img, int r0, int T, const uint4 (&r)[2], int tid) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 3) + 32 * i;
    *(uint4*)(img + at_off(row, tid & 7)) = r0 + row < T ? r[i] : make_uint4(0, 0, 0, 0);
  }
}
// one query block's Q, dO, O chunks and log-sum-exp in registers
struct QBlockRegs { uint4 q[2], d[2], o[2]; float lse; };
__device__ __forceinline__ void qblk_load(const AttnArgs& a, long row0, int b, int h, int q0, QBlockRegs& r, int tid) {
  blk_load(a.q + h * 64, a.ldq, row0, q0, a.T, r.q, tid);
  blk_load(a.dout + h * 64, a.lddo, row0, q0, a.T, r.d, tid);
  blk_load(a.o + h * 64, a.ldo, row0, q0, a.T, r.o, tid);
  r.lse = 1e30f;                                          // rows >= T: P = exp(s - 1e30) = 0
  if (tid < 64 && q0 + tid < a.T) r.lse = a.lse[((long)b * a.H + h) * a.T + q0 + tid];
}
// ... into LDS: Q and dO images, lse, and delta = rowsum(dO * O) from the chunks the 8 threads of a row hold
__device__ __forceinline__ void qblk_store(const QBlockRegs& r, int q0, int T, char* Qs, char* Os, float* lse_s, float* del_s, int tid) {
  blk_store(Qs, q0, T, r.q, tid);
  blk_store(Os, q0, T, r.d, tid);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (tid >> 3) + 32 * i;
    const bf16x8 ov = __builtin_bit_cast(bf16x8, r.o[i]), dv = __builtin_bit_cast(bf16x8, r.d[i]);
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += (float)ov[j] * (float)dv[j];
    acc += __shfl_xor(acc, 1, 64);
    acc += __shfl_xor(acc, 2, 64);
    acc += __shfl_xor(acc, 4, 64);
    if ((tid & 7) == 0) del_s[row] = q0 + row < T ? acc : 0.f;
  }
  if (tid < 64) lse_s[tid] = r.lse;
}

// dK, dV of one 64-key block: wave w owns key tile w; all query blocks stream through LDS
__global__ __launch_bounds__(256) void attn_long_bwd_dkv_kernel(const AttnArgs a, int nkb) {
  __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 128 + 2 * 64 * 4];
  char* Ks = smem;
  char* Vs = Ks + 64 * 128;
  char* Qs = Vs + 64 * 128;
  char* Os = Qs + 64 * 128;                               // dO
  float* lse_s = (float*)(Os + 64 * 128);
  float* del_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int kb = blockIdx.x % nkb, bh = blockIdx.x / nkb;
  const int b = bh / a.H, h = bh % a.H;
  const int T = a.T;
  const long row0 = (long)b * T;
  const int nqb = (T + 63) >> 6;
  const int qb0 = a.causal ? kb : 0;                      // causal: query blocks above the diagonal see none of these keys
  QBlockRegs qr;
  {
    uint4 rk[2], rv[2];
    blk_load(a.k + h * 64, a.ldk, row0, 64 * kb, T, rk, tid);
    blk_load(a.v + h * 64, a.ldv, row0, 64 * kb, T, rv, tid);
    qblk_load(a, row0, b, h, 64 * qb0, qr, tid);          // the first query block rides along
    blk_store(Ks, 64 * kb, T, rk, tid);
    blk_store(Vs, 64 * kb, T, rv, tid);
  }
  __syncthreads();
  const int keyl = 16 * wave + li, key = 64 * kb + keyl;
  bool key_ok = key < T;
  if (key_ok && a.keep) key_ok = a.keep[row0 + key] != 0.f;
  const bf16x8 kf0 = frag_row(Ks, keyl, g), kf1 = frag_row(Ks, keyl, 4 + g);
  const bf16x8 vf0 = frag_row(Vs, keyl, g), vf1 = frag_row(Vs, keyl, 4 + g);
  f32x4 dvT[4], dkT[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { dvT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dkT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int qb = qb0; qb < nqb; ++qb) {
    __syncthreads();                                      // the previous block's readers are done
    qblk_store(qr, 64 * qb, T, Qs, Os, lse_s, del_s, tid);
    __syncthreads();
    if (qb + 1 < nqb) qblk_load(a, row0, b, h, 64 * (qb + 1), qr, tid);   // in flight during this block's MFMAs
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      bf16x8 pf, dsf;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * ss + half;
        f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int qrow = 16 * qt + li;
        sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, g), kf0, sv);
        sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, 4 + g), kf1, sv);
        dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, g), vf0, dp);
        dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, 4 + g), vf1, dp);
        const float4 ls = *(const float4*)(lse_s + 16 * qt + 4 * g);
        const float4 de = *(const float4*)(del_s + 16 * qt + 4 * g);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qi = 64 * qb + 16 * qt + 4 * g + r;
          const bool ok = key_ok && (!a.causal || key <= qi);
          const float pv = ok ? __expf(sv[r] * a.scale - lsv[r]) : 0.f;
          pf[4 * half + r] = (bf16)pv;
          dsf[4 * half + r] = (bf16)(pv * (dp[r] - dev[r]) * a.scale);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dvT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Os, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane), pf, dvT[dt]);
        dkT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Qs, 32 * ss + 4 * g, 32 * ss + 16 + 4 * g, dt, lane), dsf, dkT[dt]);
      }
    }
  }
  if (key < T) {
    bf16* dvp = a.dv + (row0 + key) * a.lddv + h * 64 + 4 * g;
    bf16* dkp = a.dk + (row0 + key) * a.lddk + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 x = {(bf16)dvT[dt][0], (bf16)dvT[dt][1], (bf16)dvT[dt][2], (bf16)dvT[dt][3]};
      bf16x4 y = {(bf16)dkT[dt][0], (bf16)dkT[dt][1], (bf16)dkT[dt][2], (bf16)dkT[dt][3]};
      *(bf16x4*)(dvp + 16 * dt) = x;
      *(bf16x4*)(dkp + 16 * dt) = y;
    }
  }
}

// dQ of one 64-query block: wave w owns query tile w; all key blocks stream through LDS
__global__ __launch_bounds__(256) void attn_long_bwd_dq_kernel(const AttnArgs a, int nqb) {
  constexpr int DS_LD = 2 * 64 + 16;                      // padded dS row stride (bytes)
  __shared__ __attribute__((aligned(16))) char smem[4 * 64 * 128 + 64 * DS_LD + 2 * 64 * 4];
  char* Ks = smem;
  char* Vs = Ks + 64 * 128;
  char* Qs = Vs + 64 * 128;
  char* Os = Qs + 64 * 128;
  char* dSs = Os + 64 * 128;                              // [q][key] bf16
  float* lse_s = (float*)(dSs + 64 * DS_LD);
  float* del_s = lse_s + 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, g = lane >> 4;
  const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
  const int b = bh / a.H, h = bh % a.H;
  const int T = a.T;
  const long row0 = (long)b * T;
  uint4 rk[2], rv[2];
  {
    QBlockRegs qr;
    qblk_load(a, row0, b, h, 64 * qb, qr, tid);
    blk_load(a.k + h * 64, a.ldk, row0, 0, T, rk, tid);   // the first key block rides along
    blk_load(a.v + h * 64, a.ldv, row0, 0, T, rv, tid);
    qblk_store(qr, 64 * qb, T, Qs, Os, lse_s, del_s, tid);
  }
  f32x4 dqT[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dqT[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int nkb = (T + 63) >> 6;
  if (a.causal && qb + 1 < nkb) nkb = qb + 1;
  const int qrow = 16 * wave + li;                        // this lane's query row as an A-operand row (S, dP)
  for (int kb = 0; kb < nkb; ++kb) {
    __syncthreads();                                      // previous block's dS / K readers are done
    blk_store(Ks, 64 * kb, T, rk, tid);
    blk_store(Vs, 64 * kb, T, rv, tid);
    __syncthreads();
    if (kb + 1 < nkb) {                                   // in flight during this block's MFMAs
      blk_load(a.k + h * 64, a.ldk, row0, 64 * (kb + 1), T, rk, tid);
      blk_load(a.v + h * 64, a.ldv, row0, 64 * (kb + 1), T, rv, tid);
    }
    const float4 ls = *(const float4*)(lse_s + 16 * wave + 4 * g);
    const float4 de = *(const float4*)(del_s + 16 * wave + 4 * g);
    const float lsv[4] = {ls.x, ls.y, ls.z, ls.w}, dev[4] = {de.x, de.y, de.z, de.w};
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int keyl = 16 * kt + li, key = 64 * kb + keyl;
      bool key_ok = key < T;
      if (key_ok && a.keep) key_ok = a.keep[row0 + key] != 0.f;
      f32x4 sv = (f32x4){0.f, 0.f, 0.f, 0.f}, dp = (f32x4){0.f, 0.f, 0.f, 0.f};
      sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, g), frag_row(Ks, keyl, g), sv);
      sv = CCLIP_MFMA_16x16x32(frag_row(Qs, qrow, 4 + g), frag_row(Ks, keyl, 4 + g), sv);
      dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, g), frag_row(Vs, keyl, g), dp);
      dp = CCLIP_MFMA_16x16x32(frag_row(Os, qrow, 4 + g), frag_row(Vs, keyl, 4 + g), dp);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ql = 16 * wave + 4 * g + r, qi = 64 * qb + ql;
        const bool ok = key_ok && (!a.causal || key <= qi);
        const float pv = ok ? __expf(sv[r] * a.scale - lsv[r]) : 0.f;
        *(bf16*)(dSs + ql * DS_LD + keyl * 2) = (bf16)(pv * (dp[r] - dev[r]) * a.scale);
      }
    }
    __builtin_amdgcn_wave_barrier();                      // dS rows of this wave's query tile are written and read by the same wave
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 dsf = *(const bf16x8*)(dSs + qrow * DS_LD + (32 * ss + 8 * g) * 2);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        dqT[dt] = CCLIP_MFMA_16x16x32(frag_tr(Ks, 32 * ss + 8 * g, 32 * ss + 8 * g + 4, dt, lane), dsf, dqT[dt]);
    }
  }
  const int qi = 64 * qb + qrow;
  if (qi < T) {
    bf16* dqp = a.dq + (row0 + qi) * a.lddq + h * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      bf16x4 x = {(bf16)dqT[dt][0], (bf16)dqT[dt][1], (bf16)dqT[dt][2], (bf16)dqT[dt][3]};
      *(bf16x4*)(dqp + 16 * dt) = x;
    }
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static bool attn_args_ok(const cclip_attn_desc* d, bool bwd) {
  if (!d || !d->q || !d->k || !d->v) return false;
  if (d->o_fp8) {      // fp8 output (forward only): e4m3 rows + block scales replace o
    if (bwd || !d->o_block_scale || (d->ldo_fp8 & 3) || ((uintptr_t)d->o_fp8 & 3) || ((uintptr_t)d->o_block_scale & 1) ||
        (!d->cu_seqlens && d->ld_o_block_scale < 4 * (int64_t)d->B * d->T)) return false;   // (packed batch: the caller's row count)
  } else if (!d->o) return false;
  if (d->B <= 0 || d->H <= 0 || d->T <= 0 || d->T > 8192 || d->head_dim != 64) return false;
  if (d->cu_seqlens && d->T > 128) return false;            // packed batches: the single-workgroup kernels only
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7) || (d->ldo & 7)) return false;
  if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v | (uintptr_t)d->o) & 15) return false;
  if (bwd) {
    if (!d->dout || !d->dq || !d->dk || !d->dv || !d->lse) return false;
    if ((d->lddo & 7) || (d->lddq & 7) || (d->lddk & 7) || (d->lddv & 7)) return false;
    if (((uintptr_t)d->dout | (uintptr_t)d->dq | (uintptr_t)d->dk | (uintptr_t)d->dv) & 7) return false;
  }
  return true;
}

static AttnArgs attn_pack(const cclip_attn_desc* d) {
  AttnArgs a;
  a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v;
  a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
  a.o = (bf16*)d->o; a.ldo = d->ldo; a.lse = d->lse; a.keep = d->key_keep;
  a.B = d->B; a.T = d->T; a.H = d->H; a.causal = d->causal; a.scale = d->scale;
  a.dout = (const bf16*)d->dout; a.lddo = d->lddo;
  a.dq = (bf16*)d->dq; a.dk = (bf16*)d->dk; a.dv = (bf16*)d->dv;
  a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
  a.cu = d->cu_seqlens;
  a.o8 = (unsigned char*)d->o_fp8; a.ldo8 = d->ldo_fp8; a.omx = (unsigned char*)d->o_block_scale; a.ldomx = d->ld_o_block_scale;
  return a;
}

extern "C" int CCLIP_FN(cclip_attention_fwd)(const cclip_attn_desc* d, hipStream_t stream) {
  if (!attn_args_ok(d, false)) return CCLIP_ERR_ARG;
  const AttnArgs a = attn_pack(d);
  dim3 grid(d->B * d->H), block(256);
  const int nkt = (d->T + 15) / 16;
  if (d->T > 128) {
    const int nqb = (d->T + 127) / 128;
    hipLaunchKernelGGL(attn_long_fwd_kernel, dim3(d->B * d->H * nqb), block, 0, stream, a, nqb);
    return cclip_launch_status();
  }
  if (nkt <= 2) hipLaunchKernelGGL((attn_fwd_kernel<2>), grid, block, 0, stream, a);
  else if (nkt <= 4) hipLaunchKernelGGL((attn_fwd_kernel<4>), grid, block, 0, stream, a);
  else if (nkt <= 5) hipLaunchKernelGGL((attn_fwd_kernel<5>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<8>), grid, block, 0, stream, a);
  return cclip_launch_status();
}

extern "C" int CCLIP_FN(cclip_attention_bwd)(const cclip_attn_desc* d, hipStream_t stream) {
  if (!attn_args_ok(d, true)) return CCLIP_ERR_ARG;
  const AttnArgs a = attn_pack(d);
  dim3 grid(d->B * d->H), block(256);
  const int nkt = (d->T + 15) / 16;
  if (d->T > 128) {
    const int nb64 = (d->T + 63) / 64;
    hipLaunchKernelGGL(attn_long_bwd_dkv_kernel, dim3(d->B * d->H * nb64), block, 0, stream, a, nb64);
    hipLaunchKernelGGL(attn_long_bwd_dq_kernel, dim3(d->B * d->H * nb64), block, 0, stream, a, nb64);
    return cclip_launch_status();
  }
  if (nkt <= 2) hipLaunchKernelGGL((attn_bwd_kernel<2>), grid, block, 0, stream, a);
  else if (nkt <= 4) hipLaunchKernelGGL((attn_bwd_kernel<4>), grid, block, 0, stream, a);
  else if (nkt <= 6) hipLaunchKernelGGL((attn_bwd_kernel<6>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((attn_bwd_kernel<8>), grid, block, 0, stream, a);
  return cclip_launch_status();
}
