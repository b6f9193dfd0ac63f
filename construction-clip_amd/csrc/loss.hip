// Loss-side kernels (fp32 throughout - this is where argmax parity is decided).
//
//   l2norm fwd/bwd : f / ||f||_2 per row          (CLIP.forward, before the similarity matmul)
//   xent_rows      : per-row softmax cross-entropy with fused gradient:
//                    loss_row = lse(row) - row[label]; pred = argmax(row);
//                    dlogits = (softmax(row) - onehot(label)) * grad_scale
//                    Used for both halves of the symmetric contrastive loss
//                    (/root/reference/CLIP/train.py:162-173: CE(logits_per_image, arange) and
//                    CE(logits_per_text, arange), accuracy = argmax == label) and for the caption
//                    LM loss with ignore_index (/root/reference/CLIP_prefix_caption/train.py:357).
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, long ldx, int rows, int D,
                                                         float* __restrict__ y, long ldy, float* __restrict__ inv_norm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) { const float v = x[(long)r * ldx + c]; s += v * v; }
    const float inv = rsqrtf(wave_sum(s));
    for (int c = lane; c < D; c += 64) y[(long)r * ldy + c] = x[(long)r * ldx + c] * inv;
    if (lane == 0 && inv_norm) inv_norm[r] = inv;
  }
}
// dx = (dy - y * dot(y, dy)) * inv_norm
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, long lddy,
                                                         const float* __restrict__ y, long ldy,
                                                         const float* __restrict__ inv_norm, int rows, int D,
                                                         float* __restrict__ dx, long lddx,
                                                         const float* __restrict__ mul_dev) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float mul = mul_dev ? *mul_dev : 1.f;
  for (int r = blockIdx.x * 4 + wave; r < rows; r += gridDim.x * 4) {
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += y[(long)r * ldy + c] * dy[(long)r * lddy + c];
    s = wave_sum(s);
    const float inv = inv_norm[r] * mul;
    for (int c = lane; c < D; c += 64) dx[(long)r * lddx + c] = (dy[(long)r * lddy + c] - y[(long)r * ldy + c] * s) * inv;
  }
}

template <typename DT>
// `dlogits` may alias `logits` (clip/loss.py overwrites the logits with their gradient in place), so neither carries
// __restrict__: every element is read by the lane that later writes it, and the label's logit is taken before any store.
__global__ __launch_bounds__(256) void xent_rows_kernel(const float* logits, long ld, int R, int C,
                                                        const int* __restrict__ labels, int ignore_index,
                                                        float grad_scale, float* __restrict__ loss_row,
                                                        int* __restrict__ pred, DT* dlogits, long ldd,
                                                        float* __restrict__ rowdot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
    const float* row = logits + (long)r * ld;
    const int label = labels[r];
    float m = -__builtin_inff(), s = 0.f;
    int arg = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = row[c];
      if (v > m) { s = s * __expf(m - v) + 1.f; m = v; arg = c; }
      else s += __expf(v - m);
    }
    // combine (m, s, arg) across lanes; ties -> smallest index (torch.argmax returns the first max)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
      const int a2 = __shfl_xor(arg, o, 64);
      const float mn = fmaxf(m, m2);
      s = s * (m == mn ? 1.f : __expf(m - mn)) + s2 * (m2 == mn ? 1.f : __expf(m2 - mn));
      if (m2 > m || (m2 == m && a2 < arg)) arg = a2;
      m = mn;
    }
    const float lse = m + __logf(s);
    const bool ignored = label == ignore_index || label < 0 || label >= C;
    const float at_label = ignored ? 0.f : row[label];
    if (lane == 0) {
      if (loss_row) loss_row[r] = ignored ? 0.f : lse - at_label;
      if (pred) pred[r] = arg;
    }
    if (dlogits) {
      DT* drow = dlogits + (long)r * ldd;
      const float gs = ignored ? 0.f : grad_scale;
      float dot = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float lv = row[c];                // read before the (possibly aliasing) write of the same element
        const float d = (__expf(lv - lse) - (c == label ? 1.f : 0.f)) * gs;
        dot += d * lv;
        drow[c] = (DT)d;
      }
      if (rowdot) { dot = wave_sum(dot); if (lane == 0) rowdot[r] = dot; }
    }
  }
}

// out (+)= alpha * (mul_dev ? *mul_dev : 1) * sum_i a[i] * (b ? b[i] : 1); one block, fixed order -> deterministic
__global__ __launch_bounds__(1024) void reduce_dot_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                          float alpha, const float* __restrict__ mul_dev,
                                                          float* __restrict__ out, int accumulate) {
  __shared__ float red[16];
  float s = 0.f;
  for (long i = threadIdx.x; i < n; i += 1024) s += b ? a[i] * b[i] : a[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += red[w];
    t *= alpha * (mul_dev ? *mul_dev : 1.f);
    *out = accumulate ? *out + t : t;
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static int grid_rows4(int rows) { int g = (rows + 3) / 4; return g > 4096 ? 4096 : (g < 1 ? 1 : g); }

#ifndef CCLIP_F16
extern "C" int cclip_l2norm_fwd(const float* x, int64_t ldx, int32_t rows, int32_t D, float* y, int64_t ldy,
                                float* inv_norm, hipStream_t stream) {
  if (!x || !y || rows <= 0 || D <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3(grid_rows4(rows)), dim3(256), 0, stream, x, (long)ldx, rows, D, y, (long)ldy, inv_norm);
  return cclip_launch_status();
}
#endif
#ifndef CCLIP_F16
extern "C" int cclip_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm,
                                int32_t rows, int32_t D, float* dx, int64_t lddx, const float* mul_dev,
                                hipStream_t stream) {
  if (!dy || !y || !inv_norm || !dx || rows <= 0 || D <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3(grid_rows4(rows)), dim3(256), 0, stream, dy, (long)lddy, y, (long)ldy, inv_norm, rows, D, dx, (long)lddx, mul_dev);
  return cclip_launch_status();
}
#endif
extern "C" int CCLIP_FN(cclip_xent_rows)(const float* logits, int64_t ld, int32_t R, int32_t C, const int32_t* labels,
                               int32_t ignore_index, float grad_scale, float* loss_row, int32_t* pred,
                               void* dlogits, int32_t dlogits_is_bf16, int64_t ldd, float* rowdot,
                               hipStream_t stream) {
  if (!logits || !labels || R <= 0 || C <= 0) return CCLIP_ERR_ARG;
  if (dlogits_is_bf16 && (const void*)dlogits == (const void*)logits) return CCLIP_ERR_ARG;
  dim3 grid(grid_rows4(R)), block(256);
  if (dlogits_is_bf16)
    hipLaunchKernelGGL((xent_rows_kernel<bf16>), grid, block, 0, stream, logits, (long)ld, R, C, labels, ignore_index, grad_scale, loss_row, pred, (bf16*)dlogits, (long)ldd, rowdot);
  else
    hipLaunchKernelGGL((xent_rows_kernel<float>), grid, block, 0, stream, logits, (long)ld, R, C, labels, ignore_index, grad_scale, loss_row, pred, (float*)dlogits, (long)ldd, rowdot);
  return cclip_launch_status();
}

#ifndef CCLIP_F16
extern "C" int cclip_reduce_dot(const float* a, const float* b, int64_t n, float alpha, const float* mul_dev, float* out,
                                int32_t accumulate, hipStream_t stream) {
  if (!a || !out || n <= 0) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(reduce_dot_kernel, dim3(1), dim3(1024), 0, stream, a, b, (long)n, alpha, mul_dev, out, accumulate);
  return cclip_launch_status();
}
#endif
