// Optimiser step + weight casting over the flat parameter arena (one launch for all 151 M
// parameters; HBM-bound: 16 B read + 14 B written per parameter).
//
// AdamW in the exact form the reference optimises with: `transformers.AdamW(lr, betas=(0.9,0.999),
// eps=1e-6, weight_decay=0.0, correct_bias=True)` at /root/reference/CLIP/train.py:143 and
// /root/reference/CLIP_prefix_caption/train.py:336 (class removed from transformers >= 5; its
// published update is restated in oracle/optim_oracle.py):
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v)+eps) ;
//   then p -= lr*wd*p.
// mode 1 is torch.optim.AdamW's form (decay first, eps added after the bias-corrected sqrt).
// The bf16 compute copy of the weights is refreshed in the same pass.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

// One thread-iteration = U float4 groups a grid stride apart: the 4 U loads go out before the first use (U = 2: eight read
// streams' worth of bytes in flight per thread; the single-group form sat at 4.4 TB/s of its 30 B per parameter).
__device__ __forceinline__ void adamw_update(float (&pa)[4], const float (&ga)[4], float (&ma)[4], float (&va)[4], float lr, float b1,
                                             float b2, float eps, float wd, float bc1, float sq2, float grad_scale, int mode) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float gr = ga[j] * grad_scale;
    ma[j] = b1 * ma[j] + (1.f - b1) * gr;
    va[j] = b2 * va[j] + (1.f - b2) * gr * gr;
    if (mode == 0) {
      pa[j] -= (lr * sq2 / bc1) * ma[j] / (sqrtf(va[j]) + eps);
      if (wd > 0.f) pa[j] -= lr * wd * pa[j];
    } else {
      pa[j] *= 1.f - lr * wd;
      pa[j] -= (lr / bc1) * ma[j] / (sqrtf(va[j]) / sq2 + eps);
    }
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, long n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2,
                                                    float grad_scale, int mode, bf16* __restrict__ shadow) {
  constexpr int U = 2;
  const long n4 = n >> 2;
  const float sq2 = sqrtf(bc2);
  const long stride = gridDim.x * 256L;
  for (long i0 = blockIdx.x * 256L + threadIdx.x; i0 < n4; i0 += U * stride) {
    float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < n4) { pp[u] = ((float4*)p)[i]; gg[u] = ((const float4*)g)[i]; mm[u] = ((float4*)m)[i]; vv[u] = ((float4*)v)[i]; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i >= n4) break;
      float pa[4] = {pp[u].x, pp[u].y, pp[u].z, pp[u].w}, ga[4] = {gg[u].x, gg[u].y, gg[u].z, gg[u].w};
      float ma[4] = {mm[u].x, mm[u].y, mm[u].z, mm[u].w}, va[4] = {vv[u].x, vv[u].y, vv[u].z, vv[u].w};
      adamw_update(pa, ga, ma, va, lr, b1, b2, eps, wd, bc1, sq2, grad_scale, mode);
      ((float4*)p)[i] = make_float4(pa[0], pa[1], pa[2], pa[3]);
      ((float4*)m)[i] = make_float4(ma[0], ma[1], ma[2], ma[3]);
      ((float4*)v)[i] = make_float4(va[0], va[1], va[2], va[3]);
      if (shadow) {
        bf16x4 sdw = {(bf16)pa[0], (bf16)pa[1], (bf16)pa[2], (bf16)pa[3]};
        ((bf16x4*)shadow)[i] = sdw;
      }
    }
  }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ in, bf16* __restrict__ out, long n) {
  const long n4 = n >> 2;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
    const float4 a = ((const float4*)in)[i];
    bf16x4 s = {(bf16)a.x, (bf16)a.y, (bf16)a.z, (bf16)a.w};
    ((bf16x4*)out)[i] = s;
  }
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

static int flat_grid(long n4) { long b = (n4 + 255) / 256; return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b)); }

extern "C" int CCLIP_FN(cclip_adamw_step)(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                                float beta1, float beta2, float eps, float weight_decay, int32_t step,
                                int32_t correct_bias, float grad_scale, int32_t mode, void* bf16_shadow,
                                hipStream_t stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || n <= 0 || (n & 3) || step < 1) return CCLIP_ERR_ARG;
  if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) return CCLIP_ERR_ARG;
  float bc1 = 1.f, bc2 = 1.f;
  if (correct_bias) {
    bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    bc2 = (float)(1.0 - pow((double)beta2, (double)step));
  }
  hipLaunchKernelGGL(adamw_kernel, dim3(flat_grid(n >> 2)), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq,
                     (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, mode, (bf16*)bf16_shadow);
  return cclip_launch_status();
}

extern "C" int CCLIP_CAST_FN(const float* in, void* out, int64_t n, hipStream_t stream) {
  if (!in || !out || n <= 0 || (n & 3) || ((uintptr_t)in & 15) || ((uintptr_t)out & 7)) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(flat_grid(n >> 2)), dim3(256), 0, stream, in, (bf16*)out, (long)n);
  return cclip_launch_status();
}
