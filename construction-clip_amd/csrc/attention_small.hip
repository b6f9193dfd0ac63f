// Generic small attention (any head_dim <= 128 that is a multiple of 8, T <= 64, no mask) - forward and backward.
//
// Exists for the reference's TransformerMapper (/root/reference/CLIP_prefix_caption/train.py:141-171: 8 heads over
// dim_self 768 -> head_dim 96, sequence = clip_length + prefix_length = 40, einsum('bnhd,bmhd->bnmh') * scale,
// softmax over keys, einsum('bnmh,bmhd->bnhd')).  That path is <0.1 % of the caption step's FLOPs (1.2 GFLOP at
// batch 256), so this kernel is written for clarity, fp32 math on 16-bit I/O, everything of one (batch, head) in LDS;
// the MFMA attention in attention.hip stays specialised for head_dim 64.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

struct SmallAttnArgs {
  const bf16* q; const bf16* k; const bf16* v; long ldq, ldk, ldv;
  bf16* o; long ldo;
  float* lse;                    // [B, H, T]
  int B, T, H, DH;
  float scale;
  const bf16* dout; long lddo;
  bf16* dq; bf16* dk; bf16* dv; long lddq, lddk, lddv;
};

#define SA_MAXT 64
#define SA_MAXD 128
// LDS rows are padded to DH + 2 elements -> lanes reading different rows hit different banks; all LDS is sized
// from the actual (T, DH), so e.g. T = 40, DH = 96 takes 76 KiB in backward

__device__ __forceinline__ void sa_stage(const bf16* g, long ld, long row0, int T, int DH, float* dst, int tid) {
  const int SA_LD = DH + 2;
  for (int i = tid; i < T * (DH / 8); i += 256) {
    const int r = i / (DH / 8), c = i % (DH / 8);
    const bf16x8 x = *(const bf16x8*)(g + (row0 + r) * ld + c * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) dst[r * SA_LD + c * 8 + j] = (float)x[j];
  }
}

__global__ __launch_bounds__(256) void attn_small_fwd_kernel(const SmallAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, DH = a.DH, SA_LD = DH + 2;
  float* Ks = sm;
  float* Vs = Ks + T * SA_LD;
  float* Ps = Vs + T * SA_LD;                 // [4 waves][SA_MAXT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long row0 = (long)b * T;
  sa_stage(a.k + h * DH, a.ldk, row0, T, DH, Ks, tid);
  sa_stage(a.v + h * DH, a.ldv, row0, T, DH, Vs, tid);
  __syncthreads();
  for (int qi = wave; qi < T; qi += 4) {
    const bf16* qp = a.q + (row0 + qi) * a.ldq + h * DH;
    float s = -__builtin_inff();
    if (lane < T) {
      float acc = 0.f;
      for (int d = 0; d < DH; ++d) acc += (float)qp[d] * Ks[lane * SA_LD + d];
      s = acc * a.scale;
    }
    const float m = wave_max(s);
    const float p = lane < T ? __expf(s - m) : 0.f;
    const float l = wave_sum(p);
    Ps[wave * SA_MAXT + lane] = p;
    __builtin_amdgcn_wave_barrier();
    const float inv = 1.0f / l;
    for (int d = lane; d < DH; d += 64) {
      float acc = 0.f;
      for (int kk = 0; kk < T; ++kk) acc += Ps[wave * SA_MAXT + kk] * Vs[kk * SA_LD + d];
      a.o[(row0 + qi) * a.ldo + h * DH + d] = (bf16)(acc * inv);
    }
    if (lane == 0 && a.lse) a.lse[((long)b * a.H + h) * T + qi] = m + __logf(l);
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ __launch_bounds__(256) void attn_small_bwd_kernel(const SmallAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int T = a.T, DH = a.DH, SA_LD = DH + 2, PL = T + 1;
  float* Qs = sm;
  float* Ks = Qs + T * SA_LD;
  float* Vs = Ks + T * SA_LD;
  float* Os = Vs + T * SA_LD;                 // dO
  float* P = Os + T * SA_LD;                  // [T][T+1]
  float* dS = P + T * PL;
  float* delta = dS + T * PL;
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.H, h = blockIdx.x % a.H;
  const long row0 = (long)b * T;
  sa_stage(a.q + h * DH, a.ldq, row0, T, DH, Qs, tid);
  sa_stage(a.k + h * DH, a.ldk, row0, T, DH, Ks, tid);
  sa_stage(a.v + h * DH, a.ldv, row0, T, DH, Vs, tid);
  sa_stage(a.dout + h * DH, a.lddo, row0, T, DH, Os, tid);
  __syncthreads();
  // delta[q] = sum_d dO[q][d] * O[q][d]
  for (int qi = tid; qi < T; qi += 256) {
    const bf16* op = a.o + (row0 + qi) * a.ldo + h * DH;
    float acc = 0.f;
    for (int d = 0; d < DH; ++d) acc += Os[qi * SA_LD + d] * (float)op[d];
    delta[qi] = acc;
  }
  __syncthreads();
  // P[q][k] = exp(scale * q.k - lse[q]);  dS = P * (dO.V[k] - delta[q]) * scale
  for (int i = tid; i < T * T; i += 256) {
    const int qi = i / T, kk = i % T;
    float s = 0.f, dp = 0.f;
    for (int d = 0; d < DH; ++d) {
      s += Qs[qi * SA_LD + d] * Ks[kk * SA_LD + d];
      dp += Os[qi * SA_LD + d] * Vs[kk * SA_LD + d];
    }
    const float p = __expf(s * a.scale - a.lse[((long)b * a.H + h) * T + qi]);
    P[qi * PL + kk] = p;
    dS[qi * PL + kk] = p * (dp - delta[qi]) * a.scale;
  }
  __syncthreads();
  for (int i = tid; i < T * DH; i += 256) {
    const int r = i / DH, d = i % DH;      // r is a key row for dK/dV and a query row for dQ
    float dvv = 0.f, dkk = 0.f, dqq = 0.f;
    for (int t = 0; t < T; ++t) {
      dvv += P[t * PL + r] * Os[t * SA_LD + d];
      dkk += dS[t * PL + r] * Qs[t * SA_LD + d];
      dqq += dS[r * PL + t] * Ks[t * SA_LD + d];
    }
    a.dv[(row0 + r) * a.lddv + h * DH + d] = (bf16)dvv;
    a.dk[(row0 + r) * a.lddk + h * DH + d] = (bf16)dkk;
    a.dq[(row0 + r) * a.lddq + h * DH + d] = (bf16)dqq;
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static bool sa_ok(const cclip_attn_desc* d, bool bwd) {
  if (!d || !d->q || !d->k || !d->v || !d->o) return false;
  if (d->B <= 0 || d->H <= 0 || d->T <= 0 || d->T > SA_MAXT || d->head_dim <= 0 || d->head_dim > SA_MAXD || (d->head_dim & 7)) return false;
  if (d->causal || d->key_keep) return false;
  if ((d->ldq & 7) || (d->ldk & 7) || (d->ldv & 7)) return false;
  if (((uintptr_t)d->q | (uintptr_t)d->k | (uintptr_t)d->v) & 15) return false;
  if (bwd && (!d->dout || !d->dq || !d->dk || !d->dv || !d->lse || (d->lddo & 7) || ((uintptr_t)d->dout & 15))) return false;
  return true;
}

static SmallAttnArgs sa_pack(const cclip_attn_desc* d) {
  SmallAttnArgs a;
  a.q = (const bf16*)d->q; a.k = (const bf16*)d->k; a.v = (const bf16*)d->v; a.ldq = d->ldq; a.ldk = d->ldk; a.ldv = d->ldv;
  a.o = (bf16*)d->o; a.ldo = d->ldo; a.lse = d->lse; a.B = d->B; a.T = d->T; a.H = d->H; a.DH = d->head_dim; a.scale = d->scale;
  a.dout = (const bf16*)d->dout; a.lddo = d->lddo; a.dq = (bf16*)d->dq; a.dk = (bf16*)d->dk; a.dv = (bf16*)d->dv;
  a.lddq = d->lddq; a.lddk = d->lddk; a.lddv = d->lddv;
  return a;
}

extern "C" int CCLIP_FN(cclip_attention_small_fwd)(const cclip_attn_desc* d, hipStream_t stream) {
  if (!sa_ok(d, false)) return CCLIP_ERR_ARG;
  const size_t lds = (size_t)(2 * d->T * (d->head_dim + 2) + 4 * SA_MAXT) * sizeof(float);
  if (lds > 160 * 1024) return CCLIP_ERR_ARG;
  static size_t fwd_attr = 0;
  if (lds > fwd_attr) {
    hipFuncSetAttribute((const void*)attn_small_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    fwd_attr = lds;
  }
  hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(d->B * d->H), dim3(256), lds, stream, sa_pack(d));
  return cclip_launch_status();
}

extern "C" int CCLIP_FN(cclip_attention_small_bwd)(const cclip_attn_desc* d, hipStream_t stream) {
  if (!sa_ok(d, true)) return CCLIP_ERR_ARG;
  const size_t lds = (size_t)(4 * d->T * (d->head_dim + 2) + 2 * d->T * (d->T + 1) + d->T) * sizeof(float);
  if (lds > 160 * 1024) return CCLIP_ERR_ARG;     // e.g. T = 64 with head_dim 128 does not fit: not a shape this path issues
  static size_t bwd_attr = 0;
  if (lds > bwd_attr) {
    hipFuncSetAttribute((const void*)attn_small_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    bwd_attr = lds;
  }
  hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(d->B * d->H), dim3(256), lds, stream, sa_pack(d));
  return cclip_launch_status();
}
