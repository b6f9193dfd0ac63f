// Decode-shaped attention: ONE new query per sequence against that sequence's cached keys / values (head_dim 64).
//
// The reference's generate_beam / generate2 (/root/reference/CLIP_prefix_caption/test.py:353-514, application.py:152-229)
// call `model.gpt(inputs_embeds=generated)` on the whole, growing sequence at every step and keep only the last
// position's logits - O(S^2) work per caption.  With a KV cache the step is one token wide; this kernel is its
// softmax(q K^T * scale) V.  The new token attends to every cached position (its own included), so there is no mask.
// Work per step is tiny (beams x heads workgroups, S <= 1024 keys) and bound by reading the cache once:
// one wave per (sequence, head), fp32 math, K rows read as 128-byte rows per lane, V rows as 16-byte chunks with four
// loads in flight per lane.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

namespace CCLIP_NS {

#define DEC_MAXS 2048

__global__ __launch_bounds__(64) void attn_decode_kernel(const bf16* __restrict__ q, long ldq, const bf16* __restrict__ kc,
                                                         const bf16* __restrict__ vc, long ld_pos, long ld_seq,
                                                         bf16* __restrict__ out, long ldo, int H, int S, float scale) {
  __shared__ float p[DEC_MAXS];
  __shared__ float qs[64];
  const int lane = threadIdx.x;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const bf16* kb = kc + (long)b * ld_seq + h * 64;
  const bf16* vb = vc + (long)b * ld_seq + h * 64;
  bf16x8 kv[8];                                      // lane's first key row, 128 bytes in 8 loads that go out together
  {
    const bf16* kr = kb + (long)(lane < S ? lane : S - 1) * ld_pos;
#pragma unroll
    for (int c = 0; c < 8; ++c) kv[c] = *(const bf16x8*)(kr + 8 * c);
  }
  // ... and the first 32 V rows of the PV phase (lane (kg, c): keys kg + 8u, dimensions 8c..8c+7), so that the
  // softmax runs under their latency
  const int c = lane & 7, kg = lane >> 3;
  bf16x8 vv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int key = 8 * u + kg;
    vv[u] = *(const bf16x8*)(vb + (long)(key < S ? key : S - 1) * ld_pos + 8 * c);
  }
  const bf16 qv = q[(long)b * ldq + h * 64 + lane];
  qs[lane] = (float)qv;
  __syncthreads();
  float m = -__builtin_inff();
  for (int key = lane; key < S; key += 64) {
    if (key != lane) {
      const bf16* kr = kb + (long)key * ld_pos;
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) kv[cc] = *(const bf16x8*)(kr + 8 * cc);
    }
    float acc = 0.f;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += qs[8 * cc + j] * (float)kv[cc][j];
    acc *= scale;
    p[key] = acc;
    m = fmaxf(m, acc);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int key = lane; key < S; key += 64) {
    const float e = __expf(p[key] - m);
    p[key] = e;
    l += e;
  }
  l = wave_sum(l);
  __syncthreads();
  // O = P V: lane (kg, c) owns keys kg, kg + 8, ... and the 8 output dimensions 8c..8c+7; four 16-byte V loads are in
  // flight per trip (a one-key-per-iteration loop is S serial memory round trips), then the 8 key groups are summed
  float o[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = 0.f;
  for (int k0 = 0; k0 < S; k0 += 32) {
    float w[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int key = k0 + 8 * u + kg;
      if (k0 != 0) vv[u] = *(const bf16x8*)(vb + (long)(key < S ? key : S - 1) * ld_pos + 8 * c);
      w[u] = key < S ? p[key] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += w[u] * (float)vv[u][j];
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
  if (kg == 0) *(bf16x8*)(out + (long)b * ldo + h * 64 + 8 * c) = ov;
}

}  // namespace CCLIP_NS
using namespace CCLIP_NS;

extern "C" int CCLIP_FN(cclip_attention_decode)(const void* q, int64_t ldq, const void* kcache, const void* vcache, int64_t ld_pos,
                                               int64_t ld_seq, void* out, int64_t ldo, int32_t B, int32_t H, int32_t S,
                                               float scale, hipStream_t stream) {
  if (!q || !kcache || !vcache || !out || B <= 0 || H <= 0 || S <= 0 || S > DEC_MAXS) return CCLIP_ERR_ARG;
  if ((ld_pos & 7) || (ld_seq & 7) || (ldo & 7) || (((uintptr_t)kcache | (uintptr_t)vcache | (uintptr_t)out) & 15)) return CCLIP_ERR_ARG;
  hipLaunchKernelGGL(attn_decode_kernel, dim3(B * H), dim3(64), 0, stream, (const bf16*)q, (long)ldq, (const bf16*)kcache,
                     (const bf16*)vcache, (long)ld_pos, (long)ld_seq, (bf16*)out, (long)ldo, H, S, scale);
  return cclip_launch_status();
}
