// ABI version + small utilities of libcclip_hip.so.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

extern "C" int cclip_abi_version(void) { return CCLIP_ABI_VERSION; }

// x[i] *= alpha over a flat fp32 range (HBM-bound: 8 B per element).  Used to undo the static loss scale of the fp16
// operand mode on the gradient arena: powers of two, so the scaling itself is exact.
__global__ __launch_bounds__(256) void scale_f32_kernel(float* __restrict__ x, long n4, float alpha) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
    float4 v = ((float4*)x)[i];
    v.x *= alpha; v.y *= alpha; v.z *= alpha; v.w *= alpha;
    ((float4*)x)[i] = v;
  }
}

extern "C" int cclip_scale_f32(float* x, int64_t n, float alpha, hipStream_t stream) {
  if (!x || n <= 0 || (n & 3) || ((uintptr_t)x & 15)) return CCLIP_ERR_ARG;
  long b = ((n >> 2) + 255) / 256;
  hipLaunchKernelGGL(scale_f32_kernel, dim3((int)(b > 8192 ? 8192 : b)), dim3(256), 0, stream, x, (long)(n >> 2), alpha);
  return cclip_launch_status();
}
