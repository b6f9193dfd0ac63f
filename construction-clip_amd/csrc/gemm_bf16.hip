// bf16 MFMA GEMM with fused epilogues for gfx950 (MI355X).
//
//   C[m][n] = epilogue( alpha * sum_k A(m,k) * B(n,k) )
//
// This one kernel family carries >97 % of the FLOPs of the CLIP hot path:
// the QKV / out-proj / MLP projections of every ResidualAttentionBlock (the `nn.Linear` /
// `nn.MultiheadAttention` matmuls inside the third-party `clip` package that
// /root/reference/CLIP/train.py:161 reaches through `model(image, text)`), their dgrad/wgrad
// in backward, the prefix mapper (/root/reference/CLIP_prefix_caption/train.py:110-123) and the
// GPT-2 Conv1D projections + lm_head (train.py:268).
//
// Operand layouts (per operand, template parameter *_KC = "contraction index contiguous"):
//   A_KC=1: A(m,k) at A[m*lda + k]   (activations [M,K])
//   A_KC=0: A(m,k) at A[k*lda + m]   (wgrad: dY^T, contraction over tokens)
//   B_KC=1: B(n,k) at B[n*ldb + k]   (nn.Linear weight [N,K]; forward)
//   B_KC=0: B(n,k) at B[k*ldb + n]   (dgrad through an [N,K] weight; GPT-2 Conv1D [K,N]; wgrad X)
// so forward = (1,1), dgrad = (1,0), wgrad = (0,0): no transposed weight copies exist anywhere.
//
// Structure: 128x128x64 tile, 256 threads = 4 waves (2 x 2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16.  Operand tiles go HBM -> LDS by global_load_lds_dwordx4 (16 B/lane,
// no VGPR round trip) into two LDS buffers; one barrier per K-tile, the next tile's DMA is issued
// before the current tile's MFMAs.  LDS images are lane-linear (a DMA requirement) and
// XOR-swizzled through the *source* address so that fragment reads are bank-conflict free:
//   K-contiguous tile  [128 rows][8 chunks of 16 B]:  chunk' = chunk ^ (row & 7), ds_read_b128
//   K-strided tile     [64 k-rows][16 chunks]:        chunk' = chunk ^ ((kr&3)<<2 | (kr>>2)&3),
//                                                     read with ds_read_b64_tr_b16 (HW transpose)
// The MFMA is issued "swapped" (weights as the A operand): the accumulator then holds 4
// consecutive n per lane, and with the n-permutation P() below each lane owns two runs of 8
// consecutive output columns -> 16-byte stores, bias/residual/aux read as 16/32-byte vectors.
#include <stdlib.h>
#include "gemm_bf16_impl.h"

namespace CCLIP_NS {
bool cclip_gemm_launch_cfg1(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg2(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg3(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg4(int lay, int act, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg5(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg7(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg8(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a, int variant);
bool cclip_gemm_launch_cfg10(int lay, int act, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_cfg11(int lay, int act, dim3 grid, hipStream_t stream, const GemmArgs& a);
bool cclip_gemm_launch_skinny(int lay, int act, hipStream_t stream, const GemmArgs& a);

// ---- split-K combine: out = epilogue(sum_z ws[z]) ; 8 columns per thread ----
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, int M, int N,
                                                            float alpha, const float* __restrict__ residual, long ldr,
                                                            float* __restrict__ out_f32, bf16* __restrict__ out_bf16, long ldc,
                                                            const float* __restrict__ cs_ws, float* __restrict__ cs_dst, int cs_acc, int cs_len) {
  if (cs_dst) {                                            // fused bias gradient: sum the per-split row sums
    for (long m = blockIdx.x * 256L + threadIdx.x; m < cs_len; m += (long)gridDim.x * 256L) {
      const float prev = cs_acc ? cs_dst[m] : 0.f;
      float s = 0.f;
      int z = 0;
      for (; z + 4 <= splits; z += 4) {                    // loads first, adds in split order (see below)
        const float t0 = cs_ws[(long)(z + 0) * cs_len + m], t1 = cs_ws[(long)(z + 1) * cs_len + m];
        const float t2 = cs_ws[(long)(z + 2) * cs_len + m], t3 = cs_ws[(long)(z + 3) * cs_len + m];
        s += t0; s += t1; s += t2; s += t3;
      }
      for (; z < splits; ++z) s += cs_ws[(long)z * cs_len + m];
      cs_dst[m] = prev + s;
    }
  }
  const long total = (long)M * N / 4;
  const long zs = (long)M * N;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    // every partial (and the residual) is requested before the first add: a load -> wait -> add loop is `splits` serial
    // HBM round trips per thread, which was most of this kernel's 13 us.  Same summation order as the plain loop.
    const long m = (i * 4) / N, n = (i * 4) % N;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (residual) r = *(const float4*)(residual + m * ldr + n);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* wp = ws + i * 4;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
      const float4 t0 = *(const float4*)(wp + (z + 0) * zs), t1 = *(const float4*)(wp + (z + 1) * zs);
      const float4 t2 = *(const float4*)(wp + (z + 2) * zs), t3 = *(const float4*)(wp + (z + 3) * zs);
      s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
      s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
      s.x += t2.x; s.y += t2.y; s.z += t2.z; s.w += t2.w;
      s.x += t3.x; s.y += t3.y; s.z += t3.z; s.w += t3.w;
    }
    if (z + 2 <= splits) {
      const float4 t0 = *(const float4*)(wp + (z + 0) * zs), t1 = *(const float4*)(wp + (z + 1) * zs);
      s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
      s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
      z += 2;
    }
    if (z < splits) {
      const float4 t0 = *(const float4*)(wp + z * zs);
      s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
    }
    s.x *= alpha; s.y *= alpha; s.z *= alpha; s.w *= alpha;
    s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
    if (out_f32) *(float4*)(out_f32 + m * ldc + n) = s;
    if (out_bf16) {
      bf16x4 o = {(bf16)s.x, (bf16)s.y, (bf16)s.z, (bf16)s.w};
      *(bf16x4*)(out_bf16 + m * ldc + n) = o;
    }
  }
}


#ifndef CCLIP_F16
// (mean, rstd) of every row from its per-64-column (sum, sum of squares) partials (cclip_rowstats_combine): one thread per row,
// partials added in block order (deterministic)
__global__ __launch_bounds__(256) void rowstats_combine_kernel(const float* __restrict__ part, int nblk, int rows, int D, float eps,
                                                               float* __restrict__ stats) {
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= rows) return;
  float s1 = 0.f, s2 = 0.f;
  for (int j = 0; j < nblk; ++j) {
    const float2 v = *(const float2*)(part + ((long)j * rows + m) * 2);
    s1 += v.x; s2 += v.y;
  }
  const float mean = s1 / (float)D;
  const float var = fmaxf(s2 / (float)D - mean * mean, 0.f);
  *(float2*)(stats + (long)m * 2) = make_float2(mean, rsqrtf(var + eps));
}
#endif
}  // namespace CCLIP_NS
using namespace CCLIP_NS;

#ifndef CCLIP_F16
extern "C" int cclip_rowstats_combine(const float* partials, int32_t nblk, int32_t rows, int32_t D, float eps, float* stats, hipStream_t stream) {
  if (!partials || !stats || nblk <= 0 || rows <= 0 || D <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(rowstats_combine_kernel, dim3((rows + 255) / 256), dim3(256), 0, stream, partials, nblk, rows, D, eps, stats);
  return cclip_launch_status();
}
#endif


extern "C" int CCLIP_GEMM_FN(const cclip_gemm_desc* d, hipStream_t stream) {
  if (!d || !d->A || !d->B || d->M <= 0 || d->N <= 0 || d->K <= 0) return CCLIP_ERR_ARG;
  if ((d->lda & 7) || (d->ldb & 7) || (d->ldc & 7)) return CCLIP_ERR_ARG;
  if (!d->a_kcontig && d->b_kcontig) return CCLIP_ERR_ARG;   // (0,1) is not a layout this path uses
  if (!d->a_kcontig && !d->b_kcontig && (d->out_pre_bf16 || d->act != CCLIP_ACT_NONE || d->aux)) return CCLIP_ERR_ARG;   // wgrad epilogue forms
  if ((uintptr_t)d->A & 15 || (uintptr_t)d->B & 15) return CCLIP_ERR_ARG;
  if (d->residual && (d->ldr & 3)) return CCLIP_ERR_ARG;
  if (d->aux && (d->ldaux & 7)) return CCLIP_ERR_ARG;
  const bool is_bwd_act = d->act >= CCLIP_ACT_DQUICKGELU;
  if (is_bwd_act && !d->aux) return CCLIP_ERR_ARG;
  const int nkt = (d->K + BK - 1) / BK;
  int splits = d->split_k > 1 ? d->split_k : 1;
  if (splits > nkt) splits = nkt;
  if (splits > 1 && !d->split_ws) return CCLIP_ERR_ARG;
  if (splits > 1 && (d->bias || d->act != CCLIP_ACT_NONE || d->out_pre_bf16 || (d->N & 3))) return CCLIP_ERR_ARG;

  GemmArgs a;
  a.A = (const bf16*)d->A; a.B = (const bf16*)d->B; a.lda = d->lda; a.ldb = d->ldb;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.ktiles_per_split = (nkt + splits - 1) / splits;
  splits = (nkt + a.ktiles_per_split - 1) / a.ktiles_per_split;   // no empty split
  a.alpha = d->alpha; a.bias = d->bias; a.residual = d->residual; a.ldr = d->ldr;
  a.aux = (const bf16*)d->aux; a.ldaux = d->ldaux;
  a.out_f32 = d->out_f32; a.out_bf16 = (bf16*)d->out_bf16; a.out_pre = (bf16*)d->out_pre_bf16; a.ldc = d->ldc;
  a.act = d->act; a.split_ws = splits > 1 ? d->split_ws : nullptr;
  if (d->colsum_out) {
    if (d->a_kcontig || d->b_kcontig) return CCLIP_ERR_ARG;     // row sums of A ride on the wgrad layout only
    a.colsum_dst = d->colsum_out; a.colsum_acc = d->colsum_accumulate; a.colsum_b = d->colsum_of_b ? 1 : 0;
    a.colsum_ws = splits > 1 ? d->split_ws + (size_t)splits * d->M * d->N : nullptr;
  }

  { static const int dbg = getenv("CCLIP_GEMM_DBG") ? atoi(getenv("CCLIP_GEMM_DBG")) : 0; a.dbg = dbg; }
  a.ln_stats = d->ln_stats; a.ln_c1 = d->ln_c1; a.rowstats = d->rowstats_out;
  int cfg = d->tile_config & 255;
  a.group_n = (d->tile_config >> 8) & 255;          // column-group width of the tile order (tile_coords); 0 = row-major
  if (d->tile_config < 0 || (d->tile_config >> 16)) return CCLIP_ERR_ARG;
  if (a.group_n && (cfg == 0 || cfg == 4 || cfg == 10 || cfg == 11)) return CCLIP_ERR_ARG;   // those walk tiles their own way
  if (a.ln_stats || a.ln_c1 || a.rowstats) {        // the folded-LayerNorm forms: configuration 8, whole 256x256 tiles only
    if (cfg != 8 || (d->M & 255) || (d->N & 255) || splits > 1 || !d->a_kcontig || !d->b_kcontig) return CCLIP_ERR_ARG;
    if ((a.ln_stats == nullptr) != (a.ln_c1 == nullptr)) return CCLIP_ERR_ARG;
    if (a.ln_stats && !(d->out_bf16 && !d->out_f32 && !d->residual && !d->out_pre_bf16 && d->act < CCLIP_ACT_DQUICKGELU)) return CCLIP_ERR_ARG;
    if (a.rowstats && !(d->out_f32 && d->residual && d->out_bf16 && !d->out_pre_bf16 && d->act == CCLIP_ACT_NONE && d->ldr == d->ldc)) return CCLIP_ERR_ARG;
  }
  // M <= 8 against K-strided weights (the projections of a KV-cached decode step): weight-read-bound GEMV path
  if (cfg == 0 && splits == 1 && d->M <= 8 && !d->colsum_out && cclip_gemm_launch_skinny(d->a_kcontig * 2 + d->b_kcontig, d->act, stream, a))
    return cclip_launch_status();
  if (cfg == 4) {     // persistent streaming-epilogue kernel: forward layout, full tiles only - refused (not silently replaced) otherwise
    if (splits > 1 || !cclip_gemm_launch_cfg4(d->a_kcontig * 2 + d->b_kcontig, d->act, stream, a)) return CCLIP_ERR_ARG;
    return cclip_launch_status();
  }
  if (cfg == 8) {   // hand-scheduled 4-wave 256x256 kernel (gemm_bf16_cfg8.hip): forward layout, K % 64 == 0 - refused otherwise
    const int tiles8 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    if (splits > 1 || !cclip_gemm_launch_cfg8(d->a_kcontig * 2 + d->b_kcontig, d->act, dim3(tiles8, 1), stream, a, 0)) return CCLIP_ERR_ARG;
    return cclip_launch_status();
  }
  if (cfg == 10) {              // the same tile and K loop as a persistent kernel (cross-tile operand prefetch): K % 64 == 0, K >= 192
    if (splits > 1 || !cclip_gemm_launch_cfg10(d->a_kcontig * 2 + d->b_kcontig, d->act, stream, a)) return CCLIP_ERR_ARG;
    return cclip_launch_status();
  }
  if (cfg == 11) {              // hand-scheduled 4-wave weight-gradient kernel (gemm_bf16_cfg11.hip): layout (0,0), K % 64 == 0
    const int tiles11 = ((d->M + 255) / 256) * ((d->N + 255) / 256);
    if (!cclip_gemm_launch_cfg11(d->a_kcontig * 2 + d->b_kcontig, d->act, dim3(tiles11, splits), stream, a)) return CCLIP_ERR_ARG;
    int st11 = cclip_launch_status();
    if (st11 != CCLIP_OK || splits == 1) return st11;
    long total = (long)d->M * d->N / 4;
    int blocks = (int)((total + 255) / 256); if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, d->split_ws, splits, d->M, d->N,
                       d->alpha, d->residual, d->ldr, d->out_f32, (bf16*)d->out_bf16, d->ldc, a.colsum_ws, a.colsum_dst, a.colsum_acc, d->M);
    return cclip_launch_status();
  }
  if (cfg <= 0 || cfg > 7 || cfg == 6) cfg = (d->M >= 2048 && getenv("CCLIP_GEMM_CFG") ? atoi(getenv("CCLIP_GEMM_CFG")) : 1);
  if (cfg < 1 || cfg > 7 || cfg == 6) cfg = 1;
  const int bm = cfg == 1 ? 128 : cfg == 5 ? 192 : 256, bn = (cfg == 3 || cfg == 5 || cfg == 7) ? 256 : 128;
  const int tiles = ((d->M + bm - 1) / bm) * ((d->N + bn - 1) / bn);
  dim3 grid(tiles, splits);
  const int lay = d->a_kcontig * 2 + d->b_kcontig;   // 3 = fwd, 2 = dgrad/Conv1D, 0 = wgrad
  const bool launched = cfg == 7   ? cclip_gemm_launch_cfg7(lay, d->act, grid, stream, a)
                        : cfg == 5 ? cclip_gemm_launch_cfg5(lay, d->act, grid, stream, a)
                        : cfg == 3 ? cclip_gemm_launch_cfg3(lay, d->act, grid, stream, a)
                        : cfg == 2 ? cclip_gemm_launch_cfg2(lay, d->act, grid, stream, a)
                                   : cclip_gemm_launch_cfg1(lay, d->act, grid, stream, a);
  if (!launched) return CCLIP_ERR_ARG;
  int st = cclip_launch_status();
  if (st != CCLIP_OK) return st;
  if (splits > 1) {
    long total = (long)d->M * d->N / 4;
    int blocks = (int)((total + 255) / 256); if (blocks > (1 << 20)) blocks = 1 << 20;   // one float4 per thread
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, d->split_ws, splits, d->M, d->N,
                       d->alpha, d->residual, d->ldr, d->out_f32, (bf16*)d->out_bf16, d->ldc, a.colsum_ws, a.colsum_dst, a.colsum_acc, a.colsum_b ? d->N : d->M);
    st = cclip_launch_status();
  }
  return st;
}
