// LayerNorm forward / backward for gfx950: one 64-lane wave per row, row held in registers,
// fp32 statistics (eps 1e-5), 16-byte loads and stores.  HBM-bound: 4 B/elem in, 2 B/elem out.
//
// Replaces the nn.LayerNorm calls of the `clip` package's ResidualAttentionBlock / ln_pre /
// ln_post / ln_final (reached from /root/reference/CLIP/train.py:161) and GPT-2's ln_1/ln_2/ln_f
// (/root/reference/CLIP_prefix_caption/train.py:268).  The optional row index turns it into the
// pooled form `ln_post(x[:, 0])` / `ln_final(x)[n, argmax(text[n])]`.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

#define LN_WAVES 4

// NV = number of 256-element column chunks a lane covers (D <= 256*NV); lane owns cols c*256 + 4*lane .. +3
template <int NV>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_fwd_kernel(const float* __restrict__ x, long ldx,
                                                               const int* __restrict__ row_index, int rows, int D,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps,
                                                               bf16* __restrict__ out_bf16, float* __restrict__ out_f32,
                                                               long ldo, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)D;
  for (int r = blockIdx.x * LN_WAVES + wave; r < rows; r += gridDim.x * LN_WAVES) {
    const long src = row_index ? (long)row_index[r] : (long)r;
    const float* xr = x + src * ldx;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      v[c] = col < D ? *(const float4*)(xr + col) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        if (out_f32) *(float4*)(out_f32 + (long)r * ldo + col) = y;
        if (out_bf16) {
          bf16x4 o = {(bf16)y.x, (bf16)y.y, (bf16)y.z, (bf16)y.w};
          *(bf16x4*)(out_bf16 + (long)r * ldo + col) = o;
        }
      }
    }
  }
}

// Backward.  dy: upstream grad of the LN output (bf16 or fp32), x: the forward input rows.
//   dxhat = dy*gamma ; dx = rstd*(dxhat - mean(dxhat) - xhat*mean(dxhat*xhat))
//   dx_out[src] = (dx_res ? dx_res[src] : 0) + dx          (fp32; optional bf16 copy for the next dgrad)
//   part[block][0][col] = sum_rows dy*xhat ; part[block][1][col] = sum_rows dy   (reduced by ln_bwd_reduce)
template <int NV, typename DY>
__global__ __launch_bounds__(64 * LN_WAVES) void ln_bwd_kernel(const DY* __restrict__ dy, long lddy,
                                                               const float* __restrict__ x, long ldx,
                                                               const int* __restrict__ row_index, int rows, int D,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in,
                                                               const float* __restrict__ dx_res,
                                                               float* __restrict__ dx_out, bf16* __restrict__ dx_out_bf16,
                                                               long lddx, float* __restrict__ part) {
  __shared__ float red[LN_WAVES][2][256 * NV];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float inv_d = 1.0f / (float)D;
  float4 dg[NV], db[NV], g[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    dg[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int col = c * 256 + lane * 4;
    g[c] = col < D ? *(const float4*)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int r = blockIdx.x * LN_WAVES + wave; r < rows; r += gridDim.x * LN_WAVES) {
    const long src = row_index ? (long)row_index[r] : (long)r;
    const float mean = mean_in[r], rstd = rstd_in[r];
    float4 xh[NV], d[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      xh[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      d[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < D) {
        const float4 xv = *(const float4*)(x + src * ldx + col);
        xh[c] = make_float4((xv.x - mean) * rstd, (xv.y - mean) * rstd, (xv.z - mean) * rstd, (xv.w - mean) * rstd);
        if constexpr (sizeof(DY) == 2) {
          const bf16x4 t = *(const bf16x4*)((const bf16*)dy + (long)r * lddy + col);
          d[c] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        } else {
          d[c] = *(const float4*)((const float*)dy + (long)r * lddy + col);
        }
        dg[c].x += d[c].x * xh[c].x; dg[c].y += d[c].y * xh[c].y; dg[c].z += d[c].z * xh[c].z; dg[c].w += d[c].w * xh[c].w;
        db[c].x += d[c].x; db[c].y += d[c].y; db[c].z += d[c].z; db[c].w += d[c].w;
        d[c].x *= g[c].x; d[c].y *= g[c].y; d[c].z *= g[c].z; d[c].w *= g[c].w;
        s1 += d[c].x + d[c].y + d[c].z + d[c].w;
        s2 += d[c].x * xh[c].x + d[c].y * xh[c].y + d[c].z * xh[c].z + d[c].w * xh[c].w;
      }
    }
    const float c1 = wave_sum(s1) * inv_d, c2 = wave_sum(s2) * inv_d;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const int col = c * 256 + lane * 4;
      if (col < D) {
        float4 o;
        o.x = rstd * (d[c].x - c1 - xh[c].x * c2);
        o.y = rstd * (d[c].y - c1 - xh[c].y * c2);
        o.z = rstd * (d[c].z - c1 - xh[c].z * c2);
        o.w = rstd * (d[c].w - c1 - xh[c].w * c2);
        if (dx_res) {
          const float4 rr = *(const float4*)(dx_res + src * lddx + col);
          o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        if (dx_out) *(float4*)(dx_out + src * lddx + col) = o;
        if (dx_out_bf16) {
          bf16x4 ob = {(bf16)o.x, (bf16)o.y, (bf16)o.z, (bf16)o.w};
          *(bf16x4*)(dx_out_bf16 + src * lddx + col) = ob;
        }
      }
    }
  }
  if (!part) return;
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    *(float4*)&red[wave][0][c * 256 + lane * 4] = dg[c];
    *(float4*)&red[wave][1][c * 256 + lane * 4] = db[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * D; i += 64 * LN_WAVES) {
    const int which = i / D, col = i % D;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < LN_WAVES; ++w) s += red[w][which][col];
    part[((long)blockIdx.x * 2 + which) * D + col] = s;
  }
}

// column sums of part[nblk][2][D] -> dgamma[D], dbeta[D].  Block = 16 columns x 16 row groups, 4 loads in
// flight per thread; fixed summation order -> deterministic.
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, int nblk, int D,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            int accumulate) {
  __shared__ float red[16][17];
  const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + c;                 // column in [0, 2D)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < 2 * D) {
    const int which = i / D, col = i % D;
    const float* base = part + (long)which * D + col;
    const long st = 2L * D;
    int b = rg;
    for (; b + 48 < nblk; b += 64) {
      s0 += base[(long)b * st];
      s1 += base[(long)(b + 16) * st];
      s2 += base[(long)(b + 32) * st];
      s3 += base[(long)(b + 48) * st];
    }
    for (; b < nblk; b += 16) s0 += base[(long)b * st];
  }
  red[rg][c] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && i < 2 * D) {
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][c];
    const int which = i / D, col = i % D;
    float* dst = which == 0 ? dgamma : dbeta;
    dst[col] = accumulate ? dst[col] + t : t;
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static int ln_grid(int rows) {
  int g = (rows + LN_WAVES - 1) / LN_WAVES;
  return g > 2048 ? 2048 : (g < 1 ? 1 : g);
}

extern "C" int CCLIP_FN(cclip_layernorm_fwd)(const float* x, int64_t ldx, const int32_t* row_index, int32_t rows, int32_t D,
                                   const float* gamma, const float* beta, float eps, void* out_bf16, float* out_f32,
                                   int64_t ldo, float* mean, float* rstd, hipStream_t stream) {
  if (!x || !gamma || !beta || rows <= 0 || D <= 0 || (D & 3) || D > 1024 || (ldx & 3) || (ldo & 3)) return CCLIP_ERR_ARG;
  if (!out_bf16 && !out_f32) return CCLIP_ERR_ARG;
  const int nv = (D + 255) / 256;
  dim3 grid(ln_grid(rows)), block(64 * LN_WAVES);
#define LNF(NV) hipLaunchKernelGGL((ln_fwd_kernel<NV>), grid, block, 0, stream, x, (long)ldx, row_index, rows, D, gamma, beta, eps, (bf16*)out_bf16, out_f32, (long)ldo, mean, rstd)
  switch (nv) { case 1: LNF(1); break; case 2: LNF(2); break; case 3: LNF(3); break; default: LNF(4); }
#undef LNF
  return cclip_launch_status();
}

static int ln_bwd_grid(int rows) { int g = ln_grid(rows); return g > 1024 ? 1024 : g; }
#ifndef CCLIP_F16
extern "C" int cclip_layernorm_bwd_ws_floats(int32_t rows, int32_t D) { return ln_bwd_grid(rows) * 2 * D; }
#endif

extern "C" int CCLIP_FN(cclip_layernorm_bwd)(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ldx,
                                   const int32_t* row_index, int32_t rows, int32_t D, const float* gamma,
                                   const float* mean, const float* rstd, const float* dx_res, float* dx_out,
                                   void* dx_out_bf16, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                                   float* ws, hipStream_t stream) {
  if (!dy || !x || !gamma || !mean || !rstd || rows <= 0 || D <= 0 || (D & 3) || D > 1024) return CCLIP_ERR_ARG;
  if ((ldx & 3) || (lddy & 3) || (lddx & 3)) return CCLIP_ERR_ARG;
  if ((dgamma || dbeta) && !(dgamma && dbeta && ws)) return CCLIP_ERR_ARG;
  const int nv = (D + 255) / 256;
  // parameter-grad partials: cap the grid so the partial buffer stays small (<= 2048 blocks x 2 x D)
  const int nblk = ln_bwd_grid(rows);
  dim3 grid(nblk), block(64 * LN_WAVES);
  float* part = dgamma ? ws : nullptr;
#define LNB(NV, T) hipLaunchKernelGGL((ln_bwd_kernel<NV, T>), grid, block, 0, stream, (const T*)dy, (long)lddy, x, (long)ldx, row_index, rows, D, gamma, mean, rstd, dx_res, dx_out, (bf16*)dx_out_bf16, (long)lddx, part)
  if (dy_is_bf16) { switch (nv) { case 1: LNB(1, bf16); break; case 2: LNB(2, bf16); break; case 3: LNB(3, bf16); break; default: LNB(4, bf16); } }
  else { switch (nv) { case 1: LNB(1, float); break; case 2: LNB(2, float); break; case 3: LNB(3, float); break; default: LNB(4, float); } }
#undef LNB
  int st = cclip_launch_status();
  if (st != CCLIP_OK || !dgamma) return st;
  hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3((2 * D + 15) / 16), dim3(256), 0, stream, ws, nblk, D, dgamma, dbeta, accumulate);
  return cclip_launch_status();
}
