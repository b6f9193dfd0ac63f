// Device-side `preprocess`: Resize(n, BICUBIC) -> CenterCrop(n) -> ToTensor -> Normalize of openai/CLIP's _transform
// (reached from /root/reference/CLIP/train.py:56, CLIP/predict.py:31, CLIP_prefix_caption/parse_coco.py:41), on an
// already decoded 8-bit RGB image.  The reference does this with PIL in DataLoader workers; at bs = 1024 per GPU that
// is the host-side bottleneck.  The arithmetic is PIL's 8-bit resampler restated:
//   * separable, horizontal pass first, the intermediate image is 8-bit;
//   * per output sample a window [xmin, xmin + xmax) of integer coefficients (the double-precision bicubic weights,
//     normalised, scaled by 2^22 and rounded half away from zero - computed on the host, cclip_hip/preprocess.py);
//   * ss = 2^21 + sum(pixel * k); result = clamp(ss >> 22, 0, 255).
// The vertical pass is fused with the crop and with (u8 / 255 - mean) / std in IEEE fp32 (same operation order as numpy),
// so the output is bit-identical to the PIL + numpy pipeline (tests/test_preprocess_gpu.py).
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

#define PRE_BITS 22

// out[r][xx][c] for xx in [0, out_w): input row r, horizontal window bounds[2*xx], bounds[2*xx+1]
__global__ __launch_bounds__(256) void resample_h_u8_kernel(const unsigned char* __restrict__ in, long in_ld, int rows,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                            int out_w, unsigned char* __restrict__ out, long out_ld) {
  const long total = (long)rows * out_w;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int xx = (int)(i % out_w);
    const long r = i / out_w;
    const int xmin = bounds[2 * xx], xn = bounds[2 * xx + 1];
    const int* k = kk + (long)xx * ksize;
    const unsigned char* p = in + r * in_ld + (long)xmin * 3;
    int s0 = 1 << (PRE_BITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < xn; ++x) {
      const int w = k[x];
      s0 += p[3 * x] * w; s1 += p[3 * x + 1] * w; s2 += p[3 * x + 2] * w;
    }
    unsigned char* o = out + r * out_ld + (long)xx * 3;
    s0 >>= PRE_BITS; s1 >>= PRE_BITS; s2 >>= PRE_BITS;
    o[0] = (unsigned char)(s0 < 0 ? 0 : (s0 > 255 ? 255 : s0));
    o[1] = (unsigned char)(s1 < 0 ? 0 : (s1 > 255 ? 255 : s1));
    o[2] = (unsigned char)(s2 < 0 ? 0 : (s2 > 255 ? 255 : s2));
  }
}

// out[c][yy][xx] (fp32 CHW, n x n) from the horizontally resampled rows tmp[(row - row0)][xx][c]
__global__ __launch_bounds__(256) void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, long tmp_ld, int row0,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                              int n, float m0, float m1, float m2, float d0, float d1, float d2,
                                                              float* __restrict__ out) {
  const int total = n * n;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int xx = i % n, yy = i / n;
    const int ymin = bounds[2 * yy], yn = bounds[2 * yy + 1];
    const int* k = kk + (long)yy * ksize;
    const unsigned char* p = tmp + (long)(ymin - row0) * tmp_ld + (long)xx * 3;
    int s0 = 1 << (PRE_BITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < yn; ++y) {
      const int w = k[y];
      s0 += p[0] * w; s1 += p[1] * w; s2 += p[2] * w;
      p += tmp_ld;
    }
    s0 >>= PRE_BITS; s1 >>= PRE_BITS; s2 >>= PRE_BITS;
    s0 = s0 < 0 ? 0 : (s0 > 255 ? 255 : s0);
    s1 = s1 < 0 ? 0 : (s1 > 255 ? 255 : s1);
    s2 = s2 < 0 ? 0 : (s2 > 255 ? 255 : s2);
    out[i] = ((float)s0 / 255.0f - m0) / d0;
    out[total + i] = ((float)s1 / 255.0f - m1) / d1;
    out[2 * total + i] = ((float)s2 / 255.0f - m2) / d2;
  }
}

#ifndef CCLIP_F16
extern "C" int cclip_resample_h_u8(const uint8_t* in, int64_t in_ld, int32_t rows, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                                   int32_t out_w, uint8_t* out, int64_t out_ld, hipStream_t stream) {
  if (!in || !bounds || !kk || !out || rows <= 0 || out_w <= 0 || ksize <= 0) return CCLIP_ERR_ARG;
  long blocks = ((long)rows * out_w + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(resample_h_u8_kernel, dim3((int)blocks), dim3(256), 0, stream, in, (long)in_ld, rows, bounds, kk, ksize, out_w, out, (long)out_ld);
  return cclip_launch_status();
}

extern "C" int cclip_resample_v_norm(const uint8_t* tmp, int64_t tmp_ld, int32_t row0, const int32_t* bounds, const int32_t* kk, int32_t ksize,
                                     int32_t n, const float* mean3, const float* std3, float* out, hipStream_t stream) {
  if (!tmp || !bounds || !kk || !out || !mean3 || !std3 || n <= 0 || ksize <= 0) return CCLIP_ERR_ARG;
  int blocks = (n * n + 255) / 256; if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(resample_v_norm_kernel, dim3(blocks), dim3(256), 0, stream, tmp, (long)tmp_ld, row0, bounds, kk, ksize, n, mean3[0], mean3[1],
                     mean3[2], std3[0], std3[1], std3[2], out);
  return cclip_launch_status();
}
#endif
