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

#define CCLIP_ABI_VERSION 3
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
 * (0,0) wgrad.  Contract: lda, ldb, ldc (and ldr % 4, ldaux % 8) multiples of 8 elements and every row
 * readable up to the next multiple of 8 columns (pad the leading dimension; e.g. a vocabulary of
 * 50257 uses ld 50264) - M, N and K themselves are arbitrary; when K is not a multiple of 8 the pad
 * columns of a K-contiguous operand up to the next multiple of 8 must hold finite values (the other
 * operand's matching k-rows are zero-filled by the kernel);
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
  int32_t tile_config; /* 0 = auto (128x128; M <= 8 rows against K-strided weights takes the skinny GEMV path);
                        * 1 = 128x128 tile, 2 LDS stages; 2 = 256x128, 3 stages; 3 = 256x256, 2 stages;
                        * 4 = persistent 256x128 with the epilogue streamed under the next tile's K loop - forward layout,
                        *     M % 256 == 0, N % 128 == 0, N <= 4096, K >= 512 (640 with a residual), one of the three
                        *     epilogue forms {16-bit out | pre-activation + activation | fp32 out + residual}; status 1 otherwise.
                        * 5 = 192x256 (configuration 3's K loop with 96x64 per wave; K-contiguous A only): a tile-count
                        *     quantisation option - 9.39 rounds of 0.75-size tiles instead of 7.03 rounds of full ones.
                        * 7 = configuration 3 with the ROTATED K loop (k-step 1 of the previous K-tile is multiplied right after the
                        *     barrier while the new tile's fragments are read) - forward layout only; status 1 otherwise.
                        * 8 = 256x256 run by FOUR waves of 128x128 (one per SIMD, accumulators in AGPRs) with a hand-scheduled inline-asm
                        *     K loop (tools/gen_gemm_a4.py) and an LDS-staged epilogue that stores whole 256-byte row segments;
                        *     forward layout, K % 64 == 0, K >= 128, no split-K; status 1 otherwise.
                        * 10 = configuration 8 as a persistent kernel (one work-group per CU walks tiles, the next tile's first operands
                        *     are staged during the current tile's last iteration); K >= 192.  Not an autotuner candidate: next to a
                        *     second stream's kernels it loses to configuration 8 (it holds every CU for its whole life).
                        * 11 = the WEIGHT-GRADIENT layout (a_kcontig = b_kcontig = 0: both operands strided in K) on four 128x128 waves
                        *     with a hand-scheduled K loop (transposing LDS reads; tools/gen_gemm_a4.py), split-K, fused colsum_out
                        *     (of A; colsum_of_b is refused); M, N multiples of 8, K % 64 == 0 and at least two 64-deep K-tiles per
                        *     split (callers keep activation / gradient slabs on 64-row multiples with zero tails); status 1 otherwise.
                        * Bits 8..15 (configurations 1, 2, 3, 5, 7, 8): TILE ORDER - 0 = row-major over (row tile, column tile);
                        *     G > 0 = the column tiles in groups of G, every row tile of a group before the next group.  Work-items
                        *     that run together on one XCD then share G weight panels instead of all of them: a weight wider than
                        *     an XCD's L2 (N = 3072 at K = 768: 4.7 MB) stays resident (fc projection -8 %, 8192^3 -29 %).
                        *     Same tiles, same arithmetic per tile: bit-identical output.
                        * The host-side autotuner (cclip_hip/ops.py) times the configurations per shape. */
  /* wgrad layout (0,0) only: colsum_out[m] (+)= sum_k A(m,k) - the BIAS gradient of the layer whose weight gradient this
   * call computes (A = dY^T), taken off the operand tiles already in LDS by one extra MFMA per m-tile against an all-ones
   * fragment in the first column block of tiles.  With split_k > 1 split_ws must hold split_k*M*(N+1) floats. */
  float* colsum_out;
  int32_t colsum_accumulate;
  int32_t colsum_of_b;   /* 1: colsum_out[n] (+)= sum_k B(n,k) instead (size N; first ROW block of tiles) - the Conv1D weight layout,
                          * where dY is the B operand of the weight-gradient GEMM; workspace split_k*(M*N + max(M,N)) floats */
  /* ---- LayerNorm FOLDED into the projection that follows it (tile configuration 8; M % 256 == 0, N % 256 == 0) ----
   * LayerNorm(x) W^T + b  =  rstd_m * ( x W'^T - mean_m * c1_n ) + c2_n   with W' = gamma (.) W, c1_n = sum_k W'(n,k),
   * c2_n = b_n + sum_k beta_k W(n,k): A holds the UN-normalised rows as 16-bit values, B holds W', `bias` holds c2,
   * ln_stats = fp32 [M][2] (mean, rstd) of the rows, ln_c1 = fp32 [N].  Epilogue: v = rstd*(alpha*acc - mean*c1) + bias, then act. */
  const float* ln_stats;
  const float* ln_c1;
  /* The residual-stream form (out_f32 = alpha*acc + bias + residual) may emit what the NEXT folded projection needs: out_bf16 (a
   * 16-bit copy of the new rows; allowed next to out_f32 + residual for these configurations) and rowstats_out = fp32
   * [N/64][M][2]: (sum, sum of squares) of every new row over each 64-column block, combined by cclip_rowstats_combine. */
  float* rowstats_out;
} cclip_gemm_desc;
int cclip_gemm_bf16(const cclip_gemm_desc* d, hipStream_t stream);
/* stats[m] = (mean, rstd = rsqrt(var + eps)) of row m from nblk partial (sum, sum of squares) pairs: partials fp32 [nblk][rows][2] */
int cclip_rowstats_combine(const float* partials, int32_t nblk, int32_t rows, int32_t D, float eps, float* stats, hipStream_t stream);

/* ---- LayerNorm (fp32 statistics, eps as given) ---------------------------------------------
 * Replaces nn.LayerNorm (ln_pre / ln_1 / ln_2 / ln_post / ln_final; GPT-2 ln_1 / ln_2 / ln_f).
 * x: fp32 rows of D (row stride ldx); row r of the output normalises input row
 * (row_index ? row_index[r] : r) - the index form is the pooled `ln_post(x[:,0])` /
 * `ln_final(x)[n, argmax]`.  Outputs: bf16 and/or fp32 [rows, D] (row stride ldo), optional
 * mean[rows], rstd[rows] for backward.  D % 4 == 0, D <= 1024. */
int cclip_layernorm_fwd(const float* x, int64_t ldx, const int32_t* row_index, int32_t rows, int32_t D,
                        const float* gamma, const float* beta, float eps, void* out_bf16, float* out_f32,
                        int64_t ldo, float* mean, float* rstd, hipStream_t stream);
/* Backward.  dy: [rows, D] bf16 (dy_is_bf16) or fp32.  dx is written at the *input* row
 * (row_index applied): dx_out[src] = (dx_res ? dx_res[src] : 0) + dLN; optional bf16 copy.
 * dgamma/dbeta (both or neither): column sums over rows, `accumulate` adds to existing values;
 * ws must hold cclip_layernorm_bwd_ws_floats(rows, D) floats. */
int cclip_layernorm_bwd_ws_floats(int32_t rows, int32_t D);
int cclip_layernorm_bwd(const void* dy, int32_t dy_is_bf16, int64_t lddy, const float* x, int64_t ldx,
                        const int32_t* row_index, int32_t rows, int32_t D, const float* gamma,
                        const float* mean, const float* rstd, const float* dx_res, float* dx_out,
                        void* dx_out_bf16, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                        float* ws, hipStream_t stream);

/* ---- fused multi-head attention, head_dim 64, T <= 8192, forward and backward ------------------
 * (T > 128 - ViT-B/16 197, ViT-L/14 257, ViT-L/14@336px 577 tokens - runs the key-block-tiled
 * online-softmax forward and a two-launch backward: dK/dV per key block, dQ per query block.)  Replaces softmax(q k^T * scale + mask) v of nn.MultiheadAttention (CLIP towers; causal for the
 * text tower) and of GPT-2 (causal + key padding).  q/k/v/o/d*: bf16, row (b*T + t), head h at
 * column h*64 of the given base pointer (so a packed [B*T, 3D] qkv buffer is passed as three
 * offset pointers with the same row stride).  lse: fp32 [B,H,T] written by fwd, read by bwd.
 * key_keep: optional fp32 [B,T], 0 masks a key.  Backward needs o (for rowsum(dO*O)). */
typedef struct cclip_attn_desc {
  const void* q; const void* k; const void* v;
  int64_t ldq, ldk, ldv;
  void* o; int64_t ldo;
  float* lse;
  const float* key_keep;
  int32_t B, T, H, head_dim, causal;
  float scale;
  const void* dout; int64_t lddo;
  void* dq; void* dk; void* dv;
  int64_t lddq, lddk, lddv;
  /* cclip_attention_fwd only (round 2, fp8 inference path): o_fp8 != NULL -> the output rows are written as e4m3 [B*T, H*64]
   * (row stride ldo_fp8) + E8M0 block scales in cclip_quantize_mx_fp8's layout (plane stride ld_o_block_scale >= 4*B*T) - the
   * block-scaled A operand of the out-proj GEMM (cclip_gemm_fp8_ex), with no 16-bit round trip; o is then not written. */
  void* o_fp8; int64_t ldo_fp8; void* o_block_scale; int64_t ld_o_block_scale;
  /* packed (variable-length) batches (round 2; cclip_attention_fwd / _bwd, T <= 128): cu_seqlens != NULL (int32 [B+1], device) ->
   * sequence b occupies rows [cu[b], cu[b+1]) of q/k/v/o/d* and of key_keep, T = the longest length (lse keeps row stride T).
   * The text tower's captions end at their EOT token: rows after it never influence the pooled feature (causal attention), so
   * the tower runs on sum(len) rows instead of B*77. */
  const int32_t* cu_seqlens;
} cclip_attn_desc;
int cclip_attention_fwd(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_bwd(const cclip_attn_desc* d, hipStream_t stream);
/* Generic small attention for the reference's TransformerMapper (CLIP_prefix_caption/train.py:141-171: 8 heads of
 * 96 over 40 tokens): any head_dim <= 128 (multiple of 8), T <= 64, no mask; same descriptor; LDS must hold one
 * (batch, head): 4*T*(head_dim+2) + 2*T*(T+1) + T floats <= 160 KiB in backward. */
int cclip_attention_small_fwd(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_small_bwd(const cclip_attn_desc* d, hipStream_t stream);

/* Decode-shaped attention (head_dim 64): one new query per sequence against its KV cache - the step of the
 * KV-cached replacement for the reference's generate_beam / generate2 loops, which re-run GPT-2 on the whole
 * growing sequence every step (CLIP_prefix_caption/test.py:381,468; application.py:180).  q/out: [B, >= H*64]
 * 16-bit; kcache/vcache: position s of sequence b at element offset b*ld_seq + s*ld_pos, head h at +h*64;
 * S = number of cached positions INCLUDING the new token's own (<= 2048).  No mask (the newest token sees all).
 * ld_pos, ld_seq, ldo multiples of 8 elements; kcache, vcache, out 16-byte aligned (CCLIP_ERR_ARG otherwise). */
int cclip_attention_decode(const void* q, int64_t ldq, const void* kcache, const void* vcache, int64_t ld_pos,
                           int64_t ld_seq, void* out, int64_t ldo, int32_t B, int32_t H, int32_t S, float scale,
                           hipStream_t stream);

/* ---- exact fp32 GEMM (f32-input MFMA), generic strides ---------------------------------------
 * C[m*ldc+n] = alpha' * sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk] + beta * C[m*ldc+n], with
 * alpha' = alpha * (alpha_log_dev ? exp(*alpha_log_dev) : 1)  (logit_scale.exp() without a host sync).
 * Replaces `x @ visual.proj`, `x @ text_projection`, `logit_scale.exp() * I @ T.t()` of
 * CLIP.forward/encode_* and their backward products (tiny, accuracy critical). */
int cclip_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk,
                   int32_t M, int32_t N, int32_t K, float alpha, const float* alpha_log_dev, float beta,
                   float* C, int64_t ldc, hipStream_t stream);

/* ---- embeddings ------------------------------------------------------------------------------
 * cclip_patchify: image fp32 [B,3,R,R] -> bf16 im2col [B*T, 3*P*P], T = (R/P)^2 + 1, class slot
 *   row (t = 0) zero; column order (c, ky, kx) = conv1.weight.view(W,-1).  When 3*P*P is not a
 *   multiple of 8 (P = 14: 588) rows are zero-padded to round_up(3*P*P, 8) elements (592) so they stay
 *   16-byte aligned for the GEMM; pad the weight rows the same way.
 *   Replaces the input side of VisionTransformer.conv1 (k = s = P, no bias).
 * cclip_vit_embed_ln: x0 = patch_out + positional_embedding[t] (+ class_embedding at t = 0),
 *   x = ln_pre(x0); optional saves x0, mean, rstd.  All fp32 [rows = B*T, D].
 * cclip_text_embed: x[r] = token_embedding[text[r]] + positional_embedding[r % L] (pos may be NULL).
 * cclip_embed_scatter_add: ids are [n, L]; demb[text[r]] += dx[(r/L)*seq_stride + seq_off + r%L]  (fp32 atomics - hardware
 *   order, kept as the one-launch alternative; the default host path uses cclip_embed_segsum below;
 *   text tower: L = seq_stride = 77, seq_off = 0; caption model: the token part of a longer sequence).
 * cclip_caption_embed: x[b,s] = (s < P ? prefix_proj[b,s] : wte[ids[b,s-P]]) + wpe[s] - the
 *   `cat(clip_project(prefix), wte(cat(attribute,tokens)))` + GPT-2 position add of
 *   CLIP_prefix_caption/train.py:258-263,268.  cclip_add_positional: x = inputs_embeds + wpe[s].
 * cclip_colsum: out[c] (+)= sum_r in[r*ld + c]; in bf16 (ld % 8 == 0) or fp32 (ld % 4 == 0), 16-byte
 *   aligned; C % 4 == 0; deterministic;
 *   ws >= cclip_colsum_ws_floats(R, C) floats. */
int cclip_patchify(const float* image, void* out_bf16, int32_t B, int32_t R, int32_t P, hipStream_t stream);
int cclip_vit_embed_ln(const float* patch_out, const float* cls, const float* pos, int32_t rows, int32_t T,
                       int32_t D, const float* gamma, const float* beta, float eps, float* x0, float* x,
                       float* mean, float* rstd, hipStream_t stream);
int cclip_text_embed(const int32_t* text, const float* emb, const float* pos, int32_t rows, int32_t L,
                     int32_t D, int32_t V, float* x, hipStream_t stream);
int cclip_embed_scatter_add(const int32_t* text, const float* dx, int64_t lddx, int32_t rows, int32_t D,
                            int32_t V, float* demb, int32_t L, int32_t seq_stride, int32_t seq_off,
                            hipStream_t stream);
/* Deterministic form of the same sum (no atomics).  `order` lists the n rows sorted by token id (stable), tok_sorted their ids;
 * per list position p: cend[p] > 0 marks the start of a CHUNK (at most 64 rows of one id) ending at cend[p], cidx[p] is the
 * chunk's slot in `partial`, rlen[p] > 0 marks the start of a run of equal ids made of rlen[p] chunks.
 * One-chunk runs are added into demb[id] by the wave that owns them, longer runs through ordered partials (two launches).
 * All tables are fixed-size index vectors: the caller needs no host synchronisation.  Bit-identical from run to run.
 * cclip_embed_tables builds cend / cidx / rlen from tok_sorted alone (ids >= V = rows dropped up front, sorted to the end):
 * chunk slots are id + (chunk start >> 6), so `partial` holds (V + n/64 + 1) * D floats. */
int cclip_embed_tables(const int32_t* tok_sorted, int32_t n, int32_t V, int32_t* cend, int32_t* cidx, int32_t* rlen,
                       hipStream_t stream);
int cclip_embed_segsum(const int32_t* order, const int32_t* tok_sorted, const int32_t* cend, const int32_t* cidx,
                       const int32_t* rlen, int32_t n, const float* dx, int64_t lddx, int32_t D, float* demb, int32_t L,
                       int32_t seq_stride, int32_t seq_off, float* partial, hipStream_t stream);
int cclip_caption_embed(const float* prefix_proj, const int32_t* ids, const float* wte, const float* wpe,
                        int32_t B, int32_t P, int32_t Lt, int32_t D, int32_t V, float* x, hipStream_t stream);
int cclip_add_positional(const float* emb, const float* wpe, int32_t rows, int32_t S, int32_t D, float* x,
                         hipStream_t stream);
int cclip_colsum_ws_floats(int32_t R, int32_t C);
int cclip_colsum(const void* in, int32_t in_is_bf16, int64_t ld, int32_t R, int32_t C, float* out,
                 int32_t accumulate, float* ws, hipStream_t stream);

/* ---- loss side (fp32) ------------------------------------------------------------------------
 * cclip_l2norm_fwd/bwd: y = x / ||x||_2 per row (image_features / image_features.norm(dim=1)).
 * cclip_xent_rows: per row r with label labels[r]: loss_row = logsumexp(row) - row[label]
 *   (0 when label == ignore_index), pred = argmax(row) (first max), and, if dlogits != NULL,
 *   dlogits = (softmax(row) - onehot) * grad_scale (fp32, may alias logits; or bf16, must not);
 *   rowdot (optional) = sum_c dlogits[c] * logits[c]  (the d/d(logit_scale) contribution of the row);
 *   l2norm_bwd's mul_dev (optional) is a device scalar the result is multiplied by (upstream dloss).
 *   Replaces torch.nn.CrossEntropyLoss + torch.argmax of CLIP/train.py:162-173 and
 *   nnf.cross_entropy(ignore_index=0) of CLIP_prefix_caption/train.py:357. */
int cclip_l2norm_fwd(const float* x, int64_t ldx, int32_t rows, int32_t D, float* y, int64_t ldy,
                     float* inv_norm, hipStream_t stream);
int cclip_l2norm_bwd(const float* dy, int64_t lddy, const float* y, int64_t ldy, const float* inv_norm,
                     int32_t rows, int32_t D, float* dx, int64_t lddx, const float* mul_dev,
                     hipStream_t stream);
int cclip_xent_rows(const float* logits, int64_t ld, int32_t R, int32_t C, const int32_t* labels,
                    int32_t ignore_index, float grad_scale, float* loss_row, int32_t* pred,
                    void* dlogits, int32_t dlogits_is_bf16, int64_t ldd, float* rowdot, hipStream_t stream);
/* out (+)= alpha * (mul_dev ? *mul_dev : 1) * sum_i a[i] * (b ? b[i] : 1)   (single block, deterministic) */
int cclip_reduce_dot(const float* a, const float* b, int64_t n, float alpha, const float* mul_dev, float* out,
                     int32_t accumulate, hipStream_t stream);

/* ---- optimiser over the flat parameter arena -------------------------------------------------
 * mode 0: transformers.AdamW (CLIP/train.py:143): p -= lr*sqrt(bc2)/bc1 * m/(sqrt(v)+eps), then
 * p -= lr*wd*p.  mode 1: torch.optim.AdamW.  grad is multiplied by grad_scale first.  If
 * bf16_shadow != NULL the bf16 compute copy of the weights is rewritten in the same pass.
 * n % 4 == 0, buffers 16-byte aligned, step >= 1. */
int cclip_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                     float beta1, float beta2, float eps, float weight_decay, int32_t step,
                     int32_t correct_bias, float grad_scale, int32_t mode, void* bf16_shadow,
                     hipStream_t stream);
int cclip_cast_f32_to_bf16(const float* in, void* out, int64_t n, hipStream_t stream);
/* Batched out-of-place transpose of 16-bit matrices (bf16 or fp16: elements are moved, not interpreted): for i < n_matrices,
 * dst_base[table[i].dst_off ...] as [cols, rows] = transpose of src_base[table[i].src_off ...] as [rows, cols];
 * table_dev = n_matrices x {src_off, dst_off, rows, cols} int64 in DEVICE memory (element offsets), max_tiles = the largest
 * ceil(rows/64) * ceil(cols/64) of the batch.  Rebuilds the transposed weight shadows the dgrad GEMMs read (one launch per
 * optimiser step), so that dX = dY . W runs in the forward operand layout. */
int cclip_transpose16_batched(const void* src_base, void* dst_base, const int64_t* table_dev, int32_t n_matrices,
                              int32_t max_tiles, hipStream_t stream);
/* x[i] *= alpha over n fp32 elements (n % 4 == 0, 16-byte aligned).  The fp16 operand mode runs its backward under a static
 * power-of-two loss scale (the 16-bit gradient stream of a mean loss over thousands of rows would otherwise sink into
 * fp16 subnormals); this undoes it on the flat gradient arena. */
int cclip_scale_f32(float* x, int64_t n, float alpha, hipStream_t stream);

/* ---- device-side preprocess --------------------------------------------------------------------
 * openai/CLIP's _transform(n) = Resize(n, BICUBIC) -> CenterCrop(n) -> ToTensor -> Normalize on a decoded 8-bit RGB image
 * (HWC, 3 bytes per pixel), bit-identical to the PIL + numpy pipeline the reference runs in DataLoader workers
 * (CLIP/train.py:56).  Two launches per image: a horizontal 8-bit resampling pass (PIL's integer resampler: windows and
 * 2^22-scaled integer coefficients per output sample, computed on the host) over the rows the vertical pass needs and the
 * columns that survive the crop, then the vertical pass fused with crop + (u8/255 - mean)/std -> fp32 [3, n, n].
 * mean3 / std3 are HOST pointers to 3 floats. */
int cclip_resample_h_u8(const uint8_t* in, int64_t in_ld, int32_t rows, const int32_t* bounds, const int32_t* kk,
                        int32_t ksize, int32_t out_w, uint8_t* out, int64_t out_ld, hipStream_t stream);
int cclip_resample_v_norm(const uint8_t* tmp, int64_t tmp_ld, int32_t row0, const int32_t* bounds, const int32_t* kk,
                          int32_t ksize, int32_t n, const float* mean3, const float* std3, float* out, hipStream_t stream);

/* ---- fp8 (OCP e4m3) inference projections --------------------------------------------------------
 * BASELINE.json configs[4] (ViT-L/14@336px encode_image, fp8 MFMA path).  No reference behaviour exists for fp8
 * (the reference runs fp16 / fp32): parity of this path is unpinned and bounded against the fp32 oracle by test.
 * cclip_quantize_rows_fp8: x16 [rows, cols] (cols % 8 == 0) -> e4m3 bytes [rows, cols] and scale[r] = amax_r / 448
 *   (1 for an all-zero row), x ~= scale[r] * fp8.  Per token for activations, per output channel for weights.
 * cclip_gemm_fp8: out16[m][n] = act(scale_a[m] * scale_b[n] * sum_k A8[m][k] * B8[n][k] + bias[n]); A8 [M,K], B8 [N,K]
 *   K-contiguous e4m3, K % 16 == 0, leading dimensions % 16 == 0, 16-byte aligned; act NONE or QUICKGELU.
 *   v_mfma_scale_f32_16x16x128_f8f6f4 (the form that runs at 2x the bf16 rate on gfx950), block scales fixed at 1. */
int cclip_quantize_rows_fp8(const void* x_bf16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                            float* scale, hipStream_t stream);
/* LayerNorm (as cclip_layernorm_fwd) whose output is written as e4m3 rows + per-row scale: the A operand of the LN-fed
 * projections (qkv, fc) with no separate quantisation pass. */
int cclip_layernorm_fwd_fp8(const float* x, int64_t ldx, int32_t rows, int32_t D, const float* gamma, const float* beta,
                            float eps, void* out_fp8, int64_t ldo, float* scale, hipStream_t stream);
int cclip_gemm_fp8(const void* A8, int64_t lda, const float* scale_a, const void* B8, int64_t ldb, const float* scale_b,
                   int32_t M, int32_t N, int32_t K, const float* bias, int32_t act, void* out_bf16, int64_t ldc,
                   hipStream_t stream);
/* Round 2: block-scaled (MX) A operands, so that the projections whose input no single workgroup sees a whole row of
 * (out-proj <- attention, c_proj <- the fc GEMM's epilogue) can run in e4m3 too.
 * cclip_quantize_mx_fp8: x16 [rows, cols] (cols % 32 == 0) -> e4m3 [rows, cols] + one E8M0 byte per (row, 32 columns):
 *   x ~= 2^(e - 127) * fp8 with e the smallest exponent that keeps the block's amax within +-448.
 *   Block-scale layout (both functions): K-tile major [cols/128][rows][4] bytes - the scale of (row r, block b) is byte
 *   (b >> 2) * ld + 4 r + (b & 3), ld >= 4 * rows = the byte distance between the planes of consecutive 128-column groups.
 * cclip_gemm_fp8_ex: cclip_gemm_fp8 in descriptor form with
 *   - block_scale_a != NULL: A carries E8M0 block scales (K % 128 == 0, layout above) applied by the MFMA itself
 *     (v_mfma_scale_f32_16x16x128_f8f6f4 takes one scale byte per lane = per (row, 32-deep k block)); scale_a is not read;
 *   - exactly one output: out16 (16-bit), out_fp8 + out_block_scale (e4m3 + E8M0 per 32 output columns, N % 64 == 0: the
 *     block-scaled A operand of the NEXT GEMM written straight from the epilogue), or out_f32 = residual + result (fp32
 *     residual stream, shared leading dimension ldf; may alias).
 *   Kernel set: act NONE with any combination above except (block scales, fp8 out); act QUICKGELU with row-scaled A and
 *   out16 or out_fp8; anything else returns status 1. */
int cclip_quantize_mx_fp8(const void* x_bf16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                          void* block_scale, int64_t ld_block_scale, hipStream_t stream);
typedef struct cclip_fp8_gemm_desc {
  const void* A; int64_t lda; const float* scale_a;
  const void* block_scale_a; int64_t ld_block_scale_a;
  const void* B; int64_t ldb; const float* scale_b;
  int32_t M, N, K;
  const float* bias;
  int32_t act;
  void* out16; int64_t ldc;
  void* out_fp8; int64_t ld_out_fp8; void* out_block_scale; int64_t ld_out_block_scale;
  float* out_f32; const float* residual; int64_t ldf;
} cclip_fp8_gemm_desc;
int cclip_gemm_fp8_ex(const cclip_fp8_gemm_desc* d, hipStream_t stream);

/* ---- native driver of one KV-cached GPT-2 decode step ------------------------------------------
 * One call = the whole per-token launch sequence (per layer: ln_1, qkv GEMM, cache append, decode attention,
 * out-proj GEMM + residual, ln_2, fc GEMM + activation, proj GEMM + residual; then optionally ln_f + tied lm_head)
 * for n_seq sequences - the body of the reference's generate_beam / generate2 loops
 * (CLIP_prefix_caption/test.py:381,468), which re-run `model.gpt(inputs_embeds=generated)` on the whole sequence.
 * Issued from C++ so that a step costs ~100 kernel launches, not ~100 Python -> ctypes round trips.
 * x: fp32 [n_seq, width], in = token embedding + position embedding of position `pos`, out = final hidden state.
 * kcache / vcache: 16-bit, element (layer l, sequence b, position s, column c) at l*ld_layer + b*ld_seq + s*width + c;
 * positions [0, pos) must be filled; position pos is written.  scratch16: n_seq * (5*width + hidden) 16-bit elements.
 * linear_layout: 1 = nn.Linear [out,in] weights, 0 = GPT-2 Conv1D [in,out].  logits == NULL skips the head. */
typedef struct cclip_block_ptrs {
  const float* ln1_w; const float* ln1_b; const void* w_qkv; const float* b_qkv; const void* w_o; const float* b_o;
  const float* ln2_w; const float* ln2_b; const void* w_fc; const float* b_fc; const void* w_proj; const float* b_proj;
} cclip_block_ptrs;
typedef struct cclip_decode_desc {
  int32_t n_layer, n_seq, width, heads, hidden, act, linear_layout, pos;
  const cclip_block_ptrs* blocks;
  float* x;
  void* kcache; void* vcache;
  int64_t ld_layer, ld_seq;
  void* scratch16;
  const float* lnf_w; const float* lnf_b; const void* wte16; int32_t vocab; float* logits; int64_t ld_logits;
} cclip_decode_desc;
int cclip_gpt2_decode_step(const cclip_decode_desc* d, hipStream_t stream);

/* ---- persistent KV-cached beam search -------------------------------------------------------------
 * Replaces the body of the reference's generate_beam loop (CLIP_prefix_caption/test.py:380-434, application.py:176-222:
 * `model.gpt(inputs_embeds=generated)` on the growing sequence, temperature, softmax(-1).log(), the stopped-beam rule,
 * length-normalised top-k over [beams x vocab], token append, beam reorder, next-token embedding) with ONE kernel launch for
 * `n_steps` decode steps: one workgroup per CU, phases separated by a grid barrier (csrc/decode_persist.hip).
 * step: as for cclip_gpt2_decode_step (Conv1D weight layout only, n_seq = beam count <= 8, n_layer <= 24, head_dim 64);
 *   step.pos = position of the first token to be decoded (= length of the prefilled prefix); step.x is written by the
 *   selection (token embedding + position embedding of the chosen tokens) and need not be initialised when first = 1;
 *   step.logits (optional, fp32 [n_seq, ld_logits]) receives each step's logits.
 * first = 1: `first_logits` [vocab] are the prefill's last-position logits and the first selection is the one-sequence form
 *   (test.py:396-405: top-k of one row, scores = its log-probabilities); the key / value rows of the prefix must be in cache
 *   slot 0 and `slot_of` zero.  first = 0 continues a search from the state the previous call left.
 * slot_of: int32 [max_len][8]: cache slot holding beam b's key / value of position t (beam reorder permutes this table; the
 *   cache itself is never copied).  tokens: int32 [n_seq][ld_tokens], state[4] columns valid on entry (prompt tokens of row 0
 *   when first = 1) - one more per selection.  scores / seq_lengths / is_stopped: [n_seq] (written when first = 1).
 * state: int32 [8]: [0] barrier counter and [1] error flag (cleared by every call; error = a grid barrier timed out),
 *   [2] set once every beam has stopped, [3] number of selections made by then (the reference's loop breaks there),
 *   [4] token columns.  Zero [2..4] (or set [4] to the prompt length) before the first call.
 * select_ws: fp32 [256 * 8 * 20].  grid_cap: 0, or a cap on the number of workgroups (tests).
 * After every beam has stopped the remaining steps are skipped (the kernel exits); results are those of the loop's break. */
typedef struct cclip_beam_desc {
  cclip_decode_desc step;
  int32_t n_steps, first, stop_token, ld_tokens, max_len, grid_cap;
  float temperature;
  const float* first_logits;
  const float* wte_f32; const float* wpe_f32;
  int32_t* slot_of; int32_t* tokens;
  float* scores; float* seq_lengths; int32_t* is_stopped;
  int32_t* state; float* select_ws;
} cclip_beam_desc;
int cclip_gpt2_beam_search(const cclip_beam_desc* d, hipStream_t stream);

/* ---- IEEE fp16 twins ---------------------------------------------------------------------------
 * Every entry point above whose 16-bit buffers are bf16 has a twin with the identical signature that
 * treats them as IEEE fp16 (same MFMA rate on gfx950; 3 more mantissa bits - the reference's own CUDA
 * dtype, and the mode in which the towers meet the <= 1e-3 parity target; use bf16 for training range). */
int cclip_gemm_f16(const cclip_gemm_desc* d, hipStream_t stream);
int cclip_layernorm_fwd_f16(const float* x, int64_t ldx, const int32_t* row_index, int32_t rows, int32_t D,
                            const float* gamma, const float* beta, float eps, void* out_f16, float* out_f32,
                            int64_t ldo, float* mean, float* rstd, hipStream_t stream);
int cclip_layernorm_bwd_f16(const void* dy, int32_t dy_is_f16, int64_t lddy, const float* x, int64_t ldx,
                            const int32_t* row_index, int32_t rows, int32_t D, const float* gamma,
                            const float* mean, const float* rstd, const float* dx_res, float* dx_out,
                            void* dx_out_f16, int64_t lddx, float* dgamma, float* dbeta, int32_t accumulate,
                            float* ws, hipStream_t stream);
int cclip_attention_fwd_f16(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_bwd_f16(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_small_fwd_f16(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_small_bwd_f16(const cclip_attn_desc* d, hipStream_t stream);
int cclip_attention_decode_f16(const void* q, int64_t ldq, const void* kcache, const void* vcache, int64_t ld_pos,
                               int64_t ld_seq, void* out, int64_t ldo, int32_t B, int32_t H, int32_t S, float scale,
                               hipStream_t stream);
int cclip_patchify_f16(const float* image, void* out_f16, int32_t B, int32_t R, int32_t P, hipStream_t stream);
int cclip_colsum_f16(const void* in, int32_t in_is_f16, int64_t ld, int32_t R, int32_t C, float* out,
                     int32_t accumulate, float* ws, hipStream_t stream);
int cclip_xent_rows_f16(const float* logits, int64_t ld, int32_t R, int32_t C, const int32_t* labels,
                        int32_t ignore_index, float grad_scale, float* loss_row, int32_t* pred,
                        void* dlogits, int32_t dlogits_is_f16, int64_t ldd, float* rowdot, hipStream_t stream);
int cclip_gpt2_decode_step_f16(const cclip_decode_desc* d, hipStream_t stream);
int cclip_gpt2_beam_search_f16(const cclip_beam_desc* d, hipStream_t stream);
int cclip_quantize_rows_fp8_f16(const void* x_f16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                                float* scale, hipStream_t stream);
int cclip_gemm_fp8_f16(const void* A8, int64_t lda, const float* scale_a, const void* B8, int64_t ldb, const float* scale_b,
                       int32_t M, int32_t N, int32_t K, const float* bias, int32_t act, void* out_f16, int64_t ldc,
                       hipStream_t stream);
int cclip_quantize_mx_fp8_f16(const void* x_f16, int64_t ldx, int32_t rows, int32_t cols, void* out_fp8, int64_t ldo,
                              void* block_scale, int64_t ld_block_scale, hipStream_t stream);
int cclip_gemm_fp8_ex_f16(const cclip_fp8_gemm_desc* d, hipStream_t stream);
int cclip_adamw_step_f16(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int32_t step,
                         int32_t correct_bias, float grad_scale, int32_t mode, void* f16_shadow,
                         hipStream_t stream);
int cclip_cast_f32_to_f16(const float* in, void* out, int64_t n, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CCLIP_HIP_H */
