// Token / patch embedding plumbing around the towers (all HBM-bound, 16-byte accesses).
//
//   patchify        image fp32 NCHW -> bf16 im2col rows [B*T, 3*P*P]; row b*T+0 (class slot) is zero,
//                   row b*T+1+p is patch p in (c, ky, kx) order = conv1.weight.view(W, -1) order.
//                   The k=s=P Conv2d of VisionTransformer (reached from CLIP/train.py:161,
//                   parse_coco.py:43) then IS the bf16 MFMA GEMM against conv1.weight.
//   vit_embed_ln    x0 = patch_out + positional (+ class_embedding on slot 0); x = ln_pre(x0)
//   text_embed      x = token_embedding[text] + positional   (CLIP.encode_text head)
//   embed_scatter   dtoken_embedding[text[i]] += dx[i]       (fp32 atomics, 256 contiguous B / wave-instr)
//   colsum          out[c] (+)= sum_r in[r][c]               (bias grads, positional grads)
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B,
                                                       int R, int P, int G) {
  const int T = G * G + 1, KP = 3 * P * P, chunks = KP / 8;
  const long total = (long)B * T * chunks;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int ch = (int)(i % chunks);
    const long row = i / chunks;
    const int t = (int)(row % T);
    const long b = row / T;
    bf16x8 o;
    if (t == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
    } else {
      const int p = t - 1, py = p / G, px = p % G;
      const int k = ch * 8, c = k / (P * P), rem = k % (P * P), ky = rem / P, kx = rem % P;
      const float* src = img + (((b * 3 + c) * R) + (py * P + ky)) * (long)R + px * P + kx;
      const float4 a = *(const float4*)src, d = *(const float4*)(src + 4);
      o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
      o[4] = (bf16)d.x; o[5] = (bf16)d.y; o[6] = (bf16)d.z; o[7] = (bf16)d.w;
    }
    *(bf16x8*)(out + row * KP + ch * 8) = o;
  }
}

// any patch size (ViT-L/14: P = 14, 3*P*P = 588): rows are zero-padded to KPAD = round_up(3*P*P, 8) elements so that
// the im2col matrix keeps 16-byte rows for the GEMM's DMA; element-wise gather (4-byte reads)
__global__ __launch_bounds__(256) void patchify_any_kernel(const float* __restrict__ img, bf16* __restrict__ out, int B,
                                                           int R, int P, int G) {
  const int T = G * G + 1, KP = 3 * P * P, chunks = (KP + 7) / 8, KPAD = chunks * 8;
  const long total = (long)B * T * chunks;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int ch = (int)(i % chunks);
    const long row = i / chunks;
    const int t = (int)(row % T);
    const long b = row / T;
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)0.f;
    if (t != 0) {
      const int p = t - 1, py = p / G, px = p % G;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = ch * 8 + j;
        if (k < KP) {
          const int c = k / (P * P), rem = k % (P * P), ky = rem / P, kx = rem % P;
          o[j] = (bf16)img[(((b * 3 + c) * R) + (py * P + ky)) * (long)R + px * P + kx];
        }
      }
    }
    *(bf16x8*)(out + row * KPAD + ch * 8) = o;
  }
}

// one wave per token row: x0 = patch_out[row] + pos[t] (+ cls if t == 0); LayerNorm(x0) -> x
template <int NV>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const float* __restrict__ patch_out,
                                                           const float* __restrict__ cls,
                                                           const float* __restrict__ pos, int rows, int T, int D,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps,
                                                           float* __restrict__ x0, float* __restrict__ x,
                                                           float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)D;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const int t = r % T;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < D) {
        const float4 a = *(const float4*)(patch_out + (long)r * D + col);
        const float4 p = *(const float4*)(pos + (long)t * D + col);
        v[c] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
        if (t == 0) {
          const float4 cc = *(const float4*)(cls + col);
          v[c].x += cc.x; v[c].y += cc.y; v[c].z += cc.z; v[c].w += cc.w;
        }
        if (x0) *(float4*)(x0 + (long)r * D + col) = v[c];
      }
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
    if (lane == 0) {
      if (mean_out) mean_out[r] = mean;
      if (rstd_out) rstd_out[r] = rstd;
    }
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        const float4 g = *(const float4*)(gamma + col), b = *(const float4*)(beta + col);
        float4 y;
        y.x = (v[c].x - mean) * rstd * g.x + b.x;
        y.y = (v[c].y - mean) * rstd * g.y + b.y;
        y.z = (v[c].z - mean) * rstd * g.z + b.z;
        y.w = (v[c].w - mean) * rstd * g.w + b.w;
        *(float4*)(x + (long)r * D + col) = y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int* __restrict__ text, const float* __restrict__ emb,
                                                         const float* __restrict__ pos, int rows, int L, int D, int V,
                                                         float* __restrict__ x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    int tok = text[r];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const int t = r % L;
    for (int col = lane * 4; col < D; col += 256) {
      const float4 e = *(const float4*)(emb + (long)tok * D + col);
      float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pos) p = *(const float4*)(pos + (long)t * D + col);
      *(float4*)(x + (long)r * D + col) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
  }
}

// ids are [n_seq, L]; the gradient row of id (b, j) is dx row b*seq_stride + seq_off + j
__global__ __launch_bounds__(256) void embed_scatter_add_kernel(const int* __restrict__ text, const float* __restrict__ dx,
                                                                long lddx, int rows, int D, int V, float* __restrict__ demb,
                                                                int L, int seq_stride, int seq_off) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    int tok = text[r];
    tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
    const long src = (long)(r / L) * seq_stride + seq_off + (r % L);
    // rows whose gradient is exactly zero add nothing: in the causal text tower that is every position after the EOT token
    // (about half of a 77-token batch, all of them the same padding id 0 -> one hot row of contended atomics)
    bool any = false;
    for (int col = lane; col < D; col += 64) any |= dx[src * lddx + col] != 0.f;
    if (__ballot(any) == 0) continue;
    for (int col = lane; col < D; col += 64) atomicAdd(demb + (long)tok * D + col, dx[src * lddx + col]);
  }
}

// ---- deterministic embedding gradient: segment-owned sums over rows SORTED by token id ----
// The host sorts the row list by token id (stable: rows of one id keep their order) and marks, per list position p (integer
// index math of fixed size - no host synchronisation, cclip_hip/ops.py):
//   cend[p]  > 0: p starts a CHUNK (a piece of at most 64 rows of one id) that ends at cend[p];   0: not a chunk start
//   cidx[p]     : index of the chunk p belongs to (= its partial slot)
//   rlen[p]  > 0: p starts a RUN of equal ids made of rlen[p] chunks;                               0: not a run start
// Pass 1: one wave per chunk start sums its rows in list order; a run that is a single chunk is added straight into its
// embedding row (the wave owns it), the chunks of longer runs write partials.  Pass 2: one wave per multi-chunk run adds
// its partials in chunk order.  No atomics: the result is bit-identical from run to run.
__device__ __forceinline__ long seg_src_row(int r, int L, int seq_stride, int seq_off) {
  return (long)(r / L) * seq_stride + seq_off + (r % L);
}
__global__ __launch_bounds__(256) void embed_segsum_kernel(const int* __restrict__ order, const int* __restrict__ tok_sorted,
                                                           const int* __restrict__ cend, const int* __restrict__ cidx,
                                                           const int* __restrict__ rlen, int n, const float* __restrict__ dx,
                                                           long lddx, int D, float* __restrict__ demb, int L, int seq_stride,
                                                           int seq_off, float* __restrict__ partial) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p0 = blockIdx.x * 4 + wave; p0 < n; p0 += gridDim.x * 4) {
    const int p1 = cend[p0];
    if (p1 == 0) continue;                                              // not a chunk start (wave-uniform)
    const bool direct = rlen[p0] == 1;
    const int tok = tok_sorted[p0], slot = cidx[p0];
    for (int col = lane * 4; col < D; col += 256) {                     // 4 floats per lane per pass
      float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
      int p = p0;
      for (; p + 4 <= p1; p += 4) {                                     // four independent row loads in flight, added in order
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const float4*)(dx + seg_src_row(order[p + u], L, seq_stride, seq_off) * lddx + col);
#pragma unroll
        for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; p < p1; ++p) {
        const float4 v = *(const float4*)(dx + seg_src_row(order[p], L, seq_stride, seq_off) * lddx + col);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      if (direct) {
        float4* o = (float4*)(demb + (long)tok * D + col);
        float4 t = *o;
        t.x += s.x; t.y += s.y; t.z += s.z; t.w += s.w;
        *o = t;
      } else {
        *(float4*)(partial + (long)slot * D + col) = s;
      }
    }
  }
}
__global__ __launch_bounds__(256) void embed_segsum_finish_kernel(const int* __restrict__ tok_sorted, const int* __restrict__ cidx,
                                                                  const int* __restrict__ rlen, int n,
                                                                  const float* __restrict__ partial, int D,
                                                                  float* __restrict__ demb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p0 = blockIdx.x * 4 + wave; p0 < n; p0 += gridDim.x * 4) {
    const int k = rlen[p0];
    if (k <= 1) continue;                                               // not the start of a multi-chunk run
    const int tok = tok_sorted[p0], s0 = cidx[p0];
    for (int col = lane * 4; col < D; col += 256) {
      float4* o = (float4*)(demb + (long)tok * D + col);
      float4 t = *o;
      for (int i = 0; i < k; ++i) {
        const float4 v = *(const float4*)(partial + (long)(s0 + i) * D + col);
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      *o = t;
    }
  }
}

// The chunk / run tables embed_segsum_kernel walks, from the SORTED id list alone and without a scan: a position finds the ends of
// its run of equal ids by two binary searches of the (L2-resident) list, and a chunk's partial slot is id + (chunk start >> 6) -
// strictly increasing over chunk starts (ids are non-decreasing, chunk starts of one id are 64 apart), consecutive within a run,
// and below V + n/64 + 1.  Ids >= V (rows dropped up front: they sort to the end) form no chunk.
__global__ __launch_bounds__(256) void embed_tables_kernel(const int* __restrict__ st, int n, int V, int* __restrict__ cend,
                                                           int* __restrict__ cidx, int* __restrict__ rlen) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  const int t = st[p];
  int lo = 0, hi = p;                                  // first position holding t
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (st[mid] < t) lo = mid + 1; else hi = mid; }
  const int rs = lo;
  lo = p + 1; hi = n;                                  // first position past the run
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (st[mid] <= t) lo = mid + 1; else hi = mid; }
  const int re = lo;
  const bool valid = t < V;
  const bool chunk_start = valid && (((p - rs) & 63) == 0);
  cend[p] = chunk_start ? (p + 64 < re ? p + 64 : re) : 0;
  cidx[p] = valid ? t + (p >> 6) : 0;
  rlen[p] = (valid && p == rs) ? (re - rs + 63) >> 6 : 0;
}

// GPT-2 input rows for the prefix-caption model (CLIP_prefix_caption/train.py:258-263):
//   x[b, s] = (s < P ? prefix_proj[b, s] : wte[ids[b, s - P]]) + wpe[s]
__global__ __launch_bounds__(256) void caption_embed_kernel(const float* __restrict__ prefix_proj, const int* __restrict__ ids,
                                                            const float* __restrict__ wte, const float* __restrict__ wpe,
                                                            int B, int P, int Lt, int D, int V, float* __restrict__ x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int S = P + Lt, rows = B * S;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const int b = r / S, s = r % S;
    const float* src;
    if (s < P) src = prefix_proj + ((long)b * P + s) * D;
    else {
      int tok = ids[b * Lt + (s - P)];
      tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
      src = wte + (long)tok * D;
    }
    for (int col = lane * 4; col < D; col += 256) {
      const float4 e = *(const float4*)(src + col), p = *(const float4*)(wpe + (long)s * D + col);
      *(float4*)(x + (long)r * D + col) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
  }
}
// x = inputs_embeds + wpe[s]   (GPT2LMHeadModel(inputs_embeds=...), the generate loops' entry)
__global__ __launch_bounds__(256) void add_pos_kernel(const float* __restrict__ emb, const float* __restrict__ wpe, int rows,
                                                      int S, int D, float* __restrict__ x) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    const int s = r % S;
    for (int col = lane * 4; col < D; col += 256) {
      const float4 e = *(const float4*)(emb + (long)r * D + col), p = *(const float4*)(wpe + (long)s * D + col);
      *(float4*)(x + (long)r * D + col) = make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w);
    }
  }
}

// column sums, two deterministic passes: part[rs][c] then out[c].  8 columns (16 B of bf16) per lane.
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ in, long ld, int R, int C,
                                                             float* __restrict__ part) {
  __shared__ float red[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = blockIdx.x * 512 + lane * 8;
  const int rs = blockIdx.y, nrs = gridDim.y;
  const int r_per = (R + nrs - 1) / nrs;
  const int r0 = rs * r_per, r1 = (r0 + r_per < R) ? r0 + r_per : R;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (col < C) {
    const bool full = col + 8 <= C;                      // C % 4 == 0: a lane is full or half
    for (int r = r0 + wave; r < r1; r += 4) {
      if constexpr (sizeof(T) == 2) {
        const bf16* p = (const bf16*)in + (long)r * ld + col;
        if (full) {
          const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += (float)t[j];
        } else {
          const bf16x4 t = *(const bf16x4*)p;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] += (float)t[j];
        }
      } else {
        const float* p = (const float*)in + (long)r * ld + col;
        const float4 a = *(const float4*)p;
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        if (full) {
          const float4 b = *(const float4*)(p + 4);
          acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int c = blockIdx.x * 512 + i;
    if (c < C) part[(long)rs * C + c] = red[0][i] + red[1][i] + red[2][i] + red[3][i];
  }
}
// out[c] (+)= sum_rs part[rs][c]; block = 16 columns x 16 row groups (4 independent partial sums per thread keep
// several loads in flight), fixed summation order -> deterministic
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, int nrs, int C,
                                                           float* __restrict__ out, int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + c;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (col < C) {
    int r = rg;
    for (; r + 48 < nrs; r += 64) {
      s0 += part[(long)r * C + col];
      s1 += part[(long)(r + 16) * C + col];
      s2 += part[(long)(r + 32) * C + col];
      s3 += part[(long)(r + 48) * C + col];
    }
    for (; r < nrs; r += 16) s0 += part[(long)r * C + col];
  }
  red[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && col < C) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][c];
    out[col] = accumulate ? out[col] + t : t;
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static int grid_rows4(int rows) { int g = (rows + 3) / 4; return g > 4096 ? 4096 : (g < 1 ? 1 : g); }

extern "C" int CCLIP_FN(cclip_patchify)(const float* image, void* out_bf16, int32_t B, int32_t R, int32_t P, hipStream_t stream) {
  if (!image || !out_bf16 || B <= 0 || P <= 0 || R % P || ((uintptr_t)image & 15) || ((uintptr_t)out_bf16 & 15)) return CCLIP_ERR_ARG;
  const int G = R / P;
  const long total = (long)B * (G * G + 1) * ((3 * P * P + 7) / 8);
  long blocks = (total + 255) / 256; if (blocks > 8192) blocks = 8192;
  if (P & 7) hipLaunchKernelGGL(patchify_any_kernel, dim3((int)blocks), dim3(256), 0, stream, image, (bf16*)out_bf16, B, R, P, G);
  else hipLaunchKernelGGL(patchify_kernel, dim3((int)blocks), dim3(256), 0, stream, image, (bf16*)out_bf16, B, R, P, G);
  return cclip_launch_status();
}

#ifndef CCLIP_F16
extern "C" int cclip_vit_embed_ln(const float* patch_out, const float* cls, const float* pos, int32_t rows, int32_t T,
                                  int32_t D, const float* gamma, const float* beta, float eps, float* x0, float* x,
                                  float* mean, float* rstd, hipStream_t stream) {
  if (!patch_out || !cls || !pos || !gamma || !beta || !x || rows <= 0 || T <= 0 || (D & 3) || D > 1024) return CCLIP_ERR_ARG;
  dim3 grid(grid_rows4(rows)), block(256);
  const int nv = (D + 255) / 256;
#define VE(NV) hipLaunchKernelGGL((vit_embed_ln_kernel<NV>), grid, block, 0, stream, patch_out, cls, pos, rows, T, D, gamma, beta, eps, x0, x, mean, rstd)
  switch (nv) { case 1: VE(1); break; case 2: VE(2); break; case 3: VE(3); break; default: VE(4); }
#undef VE
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_text_embed(const int32_t* text, const float* emb, const float* pos, int32_t rows, int32_t L,
                                int32_t D, int32_t V, float* x, hipStream_t stream) {
  if (!text || !emb || !x || rows <= 0 || L <= 0 || (D & 3) || V <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(text_embed_kernel, dim3(grid_rows4(rows)), dim3(256), 0, stream, text, emb, pos, rows, L, D, V, x);
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_embed_scatter_add(const int32_t* text, const float* dx, int64_t lddx, int32_t rows, int32_t D,
                                       int32_t V, float* demb, int32_t L, int32_t seq_stride, int32_t seq_off,
                                       hipStream_t stream) {
  if (!text || !dx || !demb || rows <= 0 || D <= 0 || V <= 0 || L <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(embed_scatter_add_kernel, dim3(grid_rows4(rows)), dim3(256), 0, stream, text, dx, (long)lddx, rows, D, V,
                     demb, L, seq_stride, seq_off);
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_embed_segsum(const int32_t* order, const int32_t* tok_sorted, const int32_t* cend, const int32_t* cidx,
                                  const int32_t* rlen, int32_t n, const float* dx, int64_t lddx, int32_t D, float* demb,
                                  int32_t L, int32_t seq_stride, int32_t seq_off, float* partial, hipStream_t stream) {
  if (!order || !tok_sorted || !cend || !cidx || !rlen || !dx || !demb || !partial) return CCLIP_ERR_ARG;
  if (n <= 0 || D <= 0 || (D & 3) || (lddx & 3) || L <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(embed_segsum_kernel, dim3(grid_rows4(n)), dim3(256), 0, stream, order, tok_sorted, cend, cidx, rlen, n, dx,
                     (long)lddx, D, demb, L, seq_stride, seq_off, partial);
  hipLaunchKernelGGL(embed_segsum_finish_kernel, dim3(grid_rows4(n)), dim3(256), 0, stream, tok_sorted, cidx, rlen, n, partial, D, demb);
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_embed_tables(const int32_t* tok_sorted, int32_t n, int32_t V, int32_t* cend, int32_t* cidx, int32_t* rlen,
                                  hipStream_t stream) {
  if (!tok_sorted || !cend || !cidx || !rlen || n <= 0 || V <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(embed_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, tok_sorted, n, V, cend, cidx, rlen);
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_caption_embed(const float* prefix_proj, const int32_t* ids, const float* wte, const float* wpe,
                                   int32_t B, int32_t P, int32_t Lt, int32_t D, int32_t V, float* x, hipStream_t stream) {
  if (!wte || !wpe || !x || B <= 0 || P < 0 || Lt < 0 || P + Lt <= 0 || (D & 3) || V <= 0) return CCLIP_ERR_ARG;
  if ((P > 0 && !prefix_proj) || (Lt > 0 && !ids)) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(caption_embed_kernel, dim3(grid_rows4(B * (P + Lt))), dim3(256), 0, stream, prefix_proj, ids, wte, wpe, B,
                     P, Lt, D, V, x);
  return cclip_launch_status();
}
#endif

#ifndef CCLIP_F16
extern "C" int cclip_add_positional(const float* emb, const float* wpe, int32_t rows, int32_t S, int32_t D, float* x,
                                    hipStream_t stream) {
  if (!emb || !wpe || !x || rows <= 0 || S <= 0 || (D & 3)) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(add_pos_kernel, dim3(grid_rows4(rows)), dim3(256), 0, stream, emb, wpe, rows, S, D, x);
  return cclip_launch_status();
}
#endif

// row splits: enough blocks (~2048) to saturate HBM whatever the column count, >= 32 rows per split
static int colsum_splits(int R, int C) {
  const int cb = (C + 511) / 512;
  int s = (1536 + cb - 1) / cb;
  const int smax = (R + 63) / 64;
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  return s < 1 ? 1 : s;
}
#ifndef CCLIP_F16
extern "C" int cclip_colsum_ws_floats(int32_t R, int32_t C) { return colsum_splits(R, C) * C; }
#endif
extern "C" int CCLIP_FN(cclip_colsum)(const void* in, int32_t in_is_bf16, int64_t ld, int32_t R, int32_t C, float* out,
                            int32_t accumulate, float* ws, hipStream_t stream) {
  if (!in || !out || !ws || R <= 0 || C <= 0 || (C & 3) || (ld & 3)) return CCLIP_ERR_ARG;
  if (in_is_bf16 ? ((ld & 7) || ((uintptr_t)in & 15)) : ((uintptr_t)in & 15)) return CCLIP_ERR_ARG;
  const int nrs = colsum_splits(R, C);
  dim3 grid((C + 511) / 512, nrs), block(256);
  if (in_is_bf16) hipLaunchKernelGGL((colsum_partial_kernel<bf16>), grid, block, 0, stream, (const bf16*)in, (long)ld, R, C, ws);
  else hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, block, 0, stream, (const float*)in, (long)ld, R, C, ws);
  int st = cclip_launch_status();
  if (st != CCLIP_OK) return st;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((C + 15) / 16), dim3(256), 0, stream, ws, nrs, C, out, accumulate);
  return cclip_launch_status();
}
