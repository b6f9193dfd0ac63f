// Native driver of one KV-cached GPT-2 decode step (see cclip_gpt2_decode_step in include/cclip_hip.h).
//
// The decode step is launch-bound: 12 layers x 9 launches of microsecond kernels.  Driven from Python each launch is a
// ctypes round trip (~15 us); here the whole sequence is issued from one C++ call through the library's own entry
// points.  Same arithmetic, kernel for kernel, as BlockStack.decode_step (cclip_hip/stack.py), which stays as the
// reference implementation and the parity check of this driver.
#include "gemm_bf16_impl.h"

extern "C" int CCLIP_FN(cclip_layernorm_fwd)(const float* x, int64_t ldx, const int32_t* row_index, int32_t rows, int32_t D,
                                            const float* gamma, const float* beta, float eps, void* out16, float* out_f32,
                                            int64_t ldo, float* mean, float* rstd, hipStream_t stream);
extern "C" int CCLIP_GEMM_FN(const cclip_gemm_desc* d, hipStream_t stream);
extern "C" int CCLIP_FN(cclip_attention_decode)(const void* q, int64_t ldq, const void* kcache, const void* vcache, int64_t ld_pos,
                                               int64_t ld_seq, void* out, int64_t ldo, int32_t B, int32_t H, int32_t S,
                                               float scale, hipStream_t stream);

namespace CCLIP_NS {
bool cclip_gemm_launch_skinny(int lay, int act, hipStream_t stream, const GemmArgs& a);
}
using namespace CCLIP_NS;

namespace {
// LayerNorm + skinny GEMM (+ KV-cache append) in one launch: the decode step's LN-fed projections (n_seq <= 8, Conv1D weights)
bool ln_gemv(const float* x, int64_t ldx, const float* gamma, const float* beta, const void* W, int64_t ldb, int M, int N, int K,
             const float* bias, int act, void* out16, int64_t ldc, void* kv_k, void* kv_v, int64_t kv_ld_seq, int kv_width,
             hipStream_t stream) {
  GemmArgs a;
  a.A = nullptr; a.B = (const bf16*)W; a.lda = 0; a.ldb = ldb; a.M = M; a.N = N; a.K = K; a.ktiles_per_split = 0;
  a.alpha = 1.0f; a.bias = bias; a.residual = nullptr; a.ldr = 0; a.aux = nullptr; a.ldaux = 0;
  a.out_f32 = nullptr; a.out_bf16 = (bf16*)out16; a.out_pre = nullptr; a.ldc = ldc; a.act = act; a.split_ws = nullptr;
  a.ln_x = x; a.ln_ldx = ldx; a.ln_gamma = gamma; a.ln_beta = beta;
  a.kv_k = (bf16*)kv_k; a.kv_v = (bf16*)kv_v; a.kv_ld_seq = kv_ld_seq; a.kv_width = kv_width;
  return cclip_gemm_launch_skinny(2, act, stream, a);
}

int gemm(const void* A, int64_t lda, const void* B, int b_kcontig, int64_t ldb, int M, int N, int K, const float* bias, int act,
         const float* residual, int64_t ldr, float* out_f32, void* out16, int64_t ldc, hipStream_t stream) {
  cclip_gemm_desc g = {};
  g.A = A; g.B = B; g.a_kcontig = 1; g.b_kcontig = b_kcontig; g.lda = lda; g.ldb = ldb;
  g.M = M; g.N = N; g.K = K; g.alpha = 1.0f; g.bias = bias; g.act = act;
  g.residual = residual; g.ldr = ldr; g.out_f32 = out_f32; g.out_bf16 = out16; g.ldc = ldc;
  g.split_k = 1; g.tile_config = 0;      // 0: the library picks (M <= 8 with Conv1D weights -> the skinny GEMV path)
  return CCLIP_GEMM_FN(&g, stream);
}
}  // namespace

#define TRY(call) do { const int st_ = (call); if (st_ != CCLIP_OK) return st_; } while (0)

extern "C" int CCLIP_FN(cclip_gpt2_decode_step)(const cclip_decode_desc* d, hipStream_t stream) {
  if (!d || !d->blocks || !d->x || !d->kcache || !d->vcache || !d->scratch16) return CCLIP_ERR_ARG;
  if (d->n_layer <= 0 || d->n_seq <= 0 || d->width <= 0 || d->heads <= 0 || d->width != d->heads * 64 || d->hidden <= 0 || d->pos < 0)
    return CCLIP_ERR_ARG;
  if ((d->width & 7) || (d->hidden & 7) || (d->ld_seq & 7) || (d->ld_layer & 7)) return CCLIP_ERR_ARG;
  const int D = d->width, Hd = d->hidden, nb = d->n_seq;
  const int64_t ldrow = 5 * (int64_t)D + Hd;                      // scratch row: xn | q k v | a | g
  char* sc = (char*)d->scratch16;
  void* xn = sc;
  void* qkv = sc + 2 * (size_t)D;
  void* a = sc + 2 * (size_t)(4 * D);
  void* g = sc + 2 * (size_t)(5 * D);
  const int lin = d->linear_layout ? 1 : 0;
  for (int l = 0; l < d->n_layer; ++l) {
    const cclip_block_ptrs& w = d->blocks[l];
    char* kc = (char*)d->kcache + 2 * (size_t)(l * d->ld_layer);
    char* vc = (char*)d->vcache + 2 * (size_t)(l * d->ld_layer);
    const bool fused = !lin && nb <= 8 && (D & 7) == 0 &&
                       ln_gemv(d->x, D, w.ln1_w, w.ln1_b, w.w_qkv, 3 * D, nb, 3 * D, D, w.b_qkv, CCLIP_ACT_NONE, qkv, ldrow,
                               kc + 2 * (size_t)d->pos * D, vc + 2 * (size_t)d->pos * D, d->ld_seq, D, stream);
    if (fused) {
      if (hipGetLastError() != hipSuccess) return CCLIP_ERR_LAUNCH;
    } else {
    TRY(CCLIP_FN(cclip_layernorm_fwd)(d->x, D, nullptr, nb, D, w.ln1_w, w.ln1_b, 1e-5f, xn, nullptr, ldrow, nullptr, nullptr, stream));
    TRY(gemm(xn, ldrow, w.w_qkv, lin, lin ? D : 3 * D, nb, 3 * D, D, w.b_qkv, CCLIP_ACT_NONE, nullptr, 0, nullptr, qkv, ldrow, stream));
    // cache append: this token's k and v rows -> position pos of every sequence
    if (hipMemcpy2DAsync(kc + 2 * (size_t)d->pos * D, 2 * (size_t)d->ld_seq, (char*)qkv + 2 * (size_t)D, 2 * (size_t)ldrow, 2 * (size_t)D, nb,
                         hipMemcpyDeviceToDevice, stream) != hipSuccess) return CCLIP_ERR_LAUNCH;
    if (hipMemcpy2DAsync(vc + 2 * (size_t)d->pos * D, 2 * (size_t)d->ld_seq, (char*)qkv + 2 * (size_t)(2 * D), 2 * (size_t)ldrow, 2 * (size_t)D, nb,
                         hipMemcpyDeviceToDevice, stream) != hipSuccess) return CCLIP_ERR_LAUNCH;
    }
    TRY(CCLIP_FN(cclip_attention_decode)(qkv, ldrow, kc, vc, D, d->ld_seq, a, ldrow, nb, d->heads, d->pos + 1, 0.125f, stream));
    TRY(gemm(a, ldrow, w.w_o, lin, D, nb, D, D, w.b_o, CCLIP_ACT_NONE, d->x, D, d->x, nullptr, D, stream));
    if (!lin && nb <= 8 && ln_gemv(d->x, D, w.ln2_w, w.ln2_b, w.w_fc, Hd, nb, Hd, D, w.b_fc, d->act, g, ldrow, nullptr, nullptr, 0, 0, stream)) {
      if (hipGetLastError() != hipSuccess) return CCLIP_ERR_LAUNCH;
    } else {
      TRY(CCLIP_FN(cclip_layernorm_fwd)(d->x, D, nullptr, nb, D, w.ln2_w, w.ln2_b, 1e-5f, xn, nullptr, ldrow, nullptr, nullptr, stream));
      TRY(gemm(xn, ldrow, w.w_fc, lin, lin ? D : Hd, nb, Hd, D, w.b_fc, d->act, nullptr, 0, nullptr, g, ldrow, stream));
    }
    TRY(gemm(g, ldrow, w.w_proj, lin, lin ? Hd : D, nb, D, Hd, w.b_proj, CCLIP_ACT_NONE, d->x, D, d->x, nullptr, D, stream));
  }
  if (d->logits) {
    if (!d->lnf_w || !d->lnf_b || !d->wte16 || d->vocab <= 0 || (d->ld_logits & 7)) return CCLIP_ERR_ARG;
    TRY(CCLIP_FN(cclip_layernorm_fwd)(d->x, D, nullptr, nb, D, d->lnf_w, d->lnf_b, 1e-5f, xn, nullptr, ldrow, nullptr, nullptr, stream));
    TRY(gemm(xn, ldrow, d->wte16, 1, D, nb, d->vocab, D, nullptr, CCLIP_ACT_NONE, nullptr, 0, d->logits, nullptr, d->ld_logits, stream));
  }
  return CCLIP_OK;
}
