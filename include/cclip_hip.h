/* cclip_hip.h - C ABI of libcclip_hip.so: the MI355X (gfx950) kernels behind the CLIP hot path.
 *
 * The reference has no FFI/plugin boundary for this path: its scripts `import clip` and call
 * Python (`model(image, text)`, /root/reference/CLIP/train.py:161; `model.encode_image`,
 * /root/reference/CLIP_prefix_caption/parse_coco.py:43; `model.clip_project` + `model.gpt(...)`,
 * /root/reference/CLIP_prefix_caption/train.py:262,268).  The arithmetic behind those calls is
 * torch ATen kernels reached through the third-party `clip` / `transformers` packages
 * (SURVEY.md 2b).  Each entry point below replaces one group of those ATen launches; the
 * Python `clip` drop-in in construction-clip_amd/ binds them with ctypes.
 *
 * Conventions: every pointer is a DEVICE pointer unless stated; `stream` is a hipStream_t
 * (torch.cuda.current_stream().cuda_stream on ROCm); all launches are asynchronous and
 * graph-capturable (no allocation, no synchronisation inside); workspaces come from the caller.
 * Return value: 0 = CCLIP_OK, 1 = argument/shape/alignment contract violated (nothing was
 * launched), 2 = HIP launch error.  The Python shim maps non-zero to RuntimeError.
 * bf16 buffers are passed as `void*` (raw 16-bit storage, torch.bfloat16).
 */
#ifndef CCLIP_HIP_H
#define CCLIP_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define CCLIP_ABI_VERSION 1
int cclip_abi_version(void);

/* ---- epilogue activations (forward and their backward forms) ---- */
enum {
  CCLIP_ACT_NONE = 0,
  CCLIP_ACT_QUICKGELU = 1, /* x*sigmoid(1.702x): CLIP ResidualAttentionBlock MLP */
  CCLIP_ACT_TANH = 2,      /* prefix mapper, CLIP_prefix_caption/train.py:115-123 */
  CCLIP_ACT_GELU_NEW = 3,  /* GPT-2 MLP */
  CCLIP_ACT_RELU = 4,      /* TransformerMapper MLP, CLIP_prefix_caption/train.py:126-140 */
  /* backward forms: out = acc * act'(aux); aux = saved pre-activation (saved OUTPUT for tanh) */
  CCLIP_ACT_DQUICKGELU = 16,
  CCLIP_ACT_DTANH = 17,
  CCLIP_ACT_DGELU_NEW = 18,
  CCLIP_ACT_DRELU = 19
};

/* ---- bf16 MFMA GEMM:  C[m][n] = epi(alpha * sum_k A(m,k) B(n,k)) --------------------------
 * Replaces nn.Linear / nn.MultiheadAttention projections / Conv1D / lm_head matmuls and their
 * dgrad + wgrad.  a_kcontig: A(m,k) at A[m*lda+k] (1) or A[k*lda+m] (0); b_kcontig: B(n,k) at
 * B[n*ldb+k] (1) or B[k*ldb+n] (0).  Supported (a,b): (1,1) forward, (1,0) dgrad / Conv1D,
 * (0,0) wgrad.  Contract: K, N, lda, ldb, ldc multiples of 8; M multiple of 8 when a_kcontig=0;
 * A and B 16-byte aligned.  Epilogue order: *alpha, +bias[n], (store out_pre_bf16), act,
 * +residual[m][n] (fp32, may alias out_f32), store out_f32 and/or out_bf16.
 * split_k > 1: K is cut into split_k ranges whose fp32 partial tiles go to split_ws
 * (>= split_k*M*N floats) and are summed by a second launch (bias/act/out_pre must be unset). */
typedef struct cclip_gemm_desc {
  const void* A;
  const void* B;
  int32_t a_kcontig, b_kcontig;
  int64_t lda, ldb;
  int32_t M, N, K;
  float alpha;
  const float* bias;
  int32_t act;
  const void* aux;
  int64_t ldaux;
  const float* residual;
  int64_t ldr;
  float* out_f32;
  void* out_bf16;
  void* out_pre_bf16;
  int64_t ldc;
  int32_t split_k;
  float* split_ws;
} cclip_gemm_desc;
int cclip_gemm_bf16(const cclip_gemm_desc* d, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CCLIP_HIP_H */
