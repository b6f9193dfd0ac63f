"""Pre-LN transformer block stack executed as an explicit sequence of HIP launches, with a
hand-written backward (no per-op autograd).  Shared by the CLIP image tower, the CLIP text tower
and GPT-2: they differ only in weight layout (nn.Linear [out,in] vs Conv1D [in,out]), activation
(QuickGELU vs gelu_new) and attention mask.

Forward of one block (the `clip` package's ResidualAttentionBlock, reached from
/root/reference/CLIP/train.py:161; HF GPT2Block behind /root/reference/CLIP_prefix_caption/train.py:268):
    xn1 = LN1(x)              -> bf16          [layernorm.hip]
    qkv = xn1 Wqkv^T + b      -> bf16          [gemm_bf16.hip, bias epilogue]
    a   = softmax(q k^T) v    -> bf16, lse     [attention.hip]
    x'  = x + a Wo^T + b      -> fp32          [gemm_bf16.hip, bias + residual epilogue]
    xn2 = LN2(x')             -> bf16
    g   = act(xn2 Wfc^T + b)  -> bf16 (+ pre-activation h saved for backward, same launch)
    x'' = x' + g Wproj^T + b  -> fp32
The residual stream stays fp32; GEMM operands are bf16 with fp32 MFMA accumulation.
Everything backward needs is kept resident in HBM (~27 KB per token per layer for ViT-B/32):
288 GB makes recomputation pointless.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import duet, ops


@dataclass
class BlockWeights:
    """bf16 compute copies of the matrices + fp32 vectors of one block, and where their grads go."""
    ln1_w: torch.Tensor; ln1_b: torch.Tensor
    w_qkv: torch.Tensor; b_qkv: Optional[torch.Tensor]
    w_o: torch.Tensor; b_o: torch.Tensor
    ln2_w: torch.Tensor; ln2_b: torch.Tensor
    w_fc: torch.Tensor; b_fc: torch.Tensor
    w_proj: torch.Tensor; b_proj: torch.Tensor
    grads: Optional[Dict[str, torch.Tensor]] = None    # same field names -> fp32 grad views (None = frozen)
    wt: Optional[Dict[str, torch.Tensor]] = None       # nn.Linear layout: transposed ([in, out]) 16-bit copies of the four
                                                       # matrices for the dgrad GEMMs (ParamArena.register_transposed)


@dataclass
class StackGeometry:
    width: int
    heads: int
    tokens: int            # sequence length T
    linear_layout: bool    # True: weights [out, in] (nn.Linear); False: [in, out] (GPT-2 Conv1D)
    act: int               # ops.ACT_QUICKGELU / ops.ACT_GELU_NEW / ops.ACT_RELU
    causal: bool
    head_dim: int = 64     # 64 -> MFMA attention (attention.hip); anything else -> attention_small.hip (no mask)
    hidden: int = 0        # MLP hidden size; 0 = 4 * width


_DACT = {ops.ACT_QUICKGELU: ops.ACT_DQUICKGELU, ops.ACT_GELU_NEW: ops.ACT_DGELU_NEW, ops.ACT_RELU: ops.ACT_DRELU}


def wgrad_splits(n_out: int, k_in: int, tokens: int) -> int:
    """Split the token contraction so that (output tiles x splits) fills the 256 CUs a few times over."""
    tiles = ((n_out + 127) // 128) * ((k_in + 127) // 128)
    nkt = (tokens + 63) // 64
    s = max(1, min(32, round(768 / max(tiles, 1))))
    return max(1, min(s, nkt // 4 if nkt >= 8 else 1))


def wgrad_candidates(n_out: int, k_in: int, tokens: int):
    """(tile_config, split_k) candidates for the wgrad autotuner: aim the grid at 1x / 2x / 3x the machine's
    concurrent workgroups (512 for the 128x128 tile, 256 for the 256x128, 256x256 and 192x256 tiles), >= 4 K-tiles per split."""
    nkt = (tokens + 63) // 64
    cands = []
    for cfg, em, en, slots in ((1, 128, 128, 512), (2, 256, 128, 256), (3, 256, 256, 256), (5, 192, 256, 256), (11, 256, 256, 256)):
        tiles = ((n_out + em - 1) // em) * ((k_in + en - 1) // en)
        for mult in (1, 2, 3):
            s = max(1, min(64, (slots * mult) // max(tiles, 1)))
            s = max(1, min(s, nkt // 4 if nkt >= 8 else 1))
            if (cfg, s) not in cands:
                cands.append((cfg, s))
    return cands


class Scratch:
    """Grow-only device scratch shared by the launches of one stream (split-K slabs, LN/colsum partials)."""

    def __init__(self, device):
        self.device = device
        self._buf: Optional[torch.Tensor] = None

    def floats(self, n: int) -> torch.Tensor:
        if self._buf is None or self._buf.numel() < n:
            self._buf = torch.empty(max(n, 1 << 20), device=self.device, dtype=torch.float32)
        return self._buf


class BlockStack:
    def __init__(self, geo: StackGeometry, blocks: List[BlockWeights], scratch: Scratch,
                 dtype: torch.dtype = torch.bfloat16):
        self.geo, self.blocks, self.scratch, self.dtype = geo, blocks, scratch, dtype
        # fp8 (e4m3) inference projections (BASELINE config 5): the LN-fed GEMMs (qkv, fc) take their A operand as e4m3
        # rows straight from a LayerNorm with fused quantisation, weights quantised per output channel once
        self.fp8 = False
        # round 2: out-proj and c_proj too, their A operands block-scaled (one E8M0 exponent per 32 k: the attention output is
        # quantised by one extra pass, the MLP hidden leaves the fc GEMM's epilogue as e4m3 + block scales directly)
        self.fp8_wide = True
        self._fp8_weights = None
        # inference: LayerNorm folded into the projection behind it (forward(), `fold`); the gamma-scaled weight copies and their
        # column sums are rebuilt when `weights_version` (the arena's parameter stamp) moves
        self._fold_w = None
        self._fold_version = None

    def fold_enabled(self) -> bool:
        import os
        return os.environ.get("CCLIP_LN_FOLD", "0") == "1" and self.geo.linear_layout and not self.fp8

    def _fold_weights(self, version):
        """Per block: W' = gamma (.) W as 16-bit operands, c1 = row sums of W' (fp32, of the ROUNDED values: the folded epilogue's
        `mean * c1` then removes exactly the mean term the MFMAs added), c2 = b + W beta.  ln_1 -> qkv, ln_2 -> fc."""
        if self._fold_w is None or self._fold_version != version:
            out = []
            for w in self.blocks:
                ent = {}
                for tag, wm, bias, gam, bet in (("q", w.w_qkv, w.b_qkv, w.ln1_w, w.ln1_b), ("f", w.w_fc, w.b_fc, w.ln2_w, w.ln2_b)):
                    wf = wm.float()
                    ws = (wf * gam.float()[None, :]).to(wm.dtype).contiguous()
                    c1 = ws.float().sum(dim=1).contiguous()
                    c2 = (wf @ bet.float()).contiguous()
                    if bias is not None:
                        c2 = (c2 + bias.float()).contiguous()
                    ent[tag] = (ws, c1, c2)
                out.append(ent)
            self._fold_w, self._fold_version = out, version
        return self._fold_w

    def quantise_weights_fp8(self) -> None:
        """(Re)build the e4m3 copies + per-channel scales of every block's qkv / fc weights from the 16-bit shadows."""
        assert self.geo.linear_layout, "fp8 projections: nn.Linear weight layout only"
        out = []
        for w in self.blocks:
            ent = {}
            for name in ("w_qkv", "w_fc", "w_o", "w_proj"):
                m = getattr(w, name)
                q = torch.empty(m.shape, device=m.device, dtype=torch.uint8)
                sc = torch.empty(m.shape[0], device=m.device, dtype=torch.float32)
                ops.quantize_rows_fp8(m, q, sc)
                ent[name] = (q, sc)
            out.append(ent)
        self._fp8_weights = out

    # ------------------------------------------------------------------ forward
    def alloc_saved(self, B: int, device, T: Optional[int] = None, M: Optional[int] = None) -> dict:
        """Activation store for one training forward.  The caller writes the stack input (fp32 [B*T, D])
        into saved["xs"][0, 0] and passes that view as `x`.  M: row count of a PACKED batch (sequences of different lengths
        back to back, forward(..., cu=...)); default B*T."""
        D, H = self.geo.width, self.geo.heads
        Hd = self.geo.hidden or 4 * D
        T = T or self.geo.tokens
        M, L = (M or B * T), len(self.blocks)
        # The 16-bit activation slab is allocated on whole 64-row K-tiles with its tail rows ZERO: the weight-gradient GEMMs contract
        # over tokens, and the hand-scheduled kernel (tile configuration 11) only takes whole K-tiles - a packed text batch has an
        # arbitrary row count.  Rows [M, Mp) are never written (every launch is told M), so they stay zero and add exact zeros.
        Mp = (M + 63) // 64 * 64
        bf = torch.empty(L, Mp, 6 * D + 2 * Hd, device=device, dtype=self.dtype)   # xn1 | qkv | a | xn2 | h | g
        if Mp != M:
            bf[:, M:].zero_()
        return dict(T=T, M=M, Mp=Mp, cu=None, bf=bf,
            xs=torch.empty(L, 2, M, D, device=device, dtype=torch.float32),      # x_in, x_mid
            st=torch.empty(L, 4, M, device=device, dtype=torch.float32),         # mean1 rstd1 mean2 rstd2
            lse=torch.empty(L, B, H, T, device=device, dtype=torch.float32),
            out=torch.empty(M, D, device=device, dtype=torch.float32), B=B, key_keep=None)

    def forward(self, x: torch.Tensor, B: int, *, saved: Optional[dict] = None,
                key_keep: Optional[torch.Tensor] = None, T: Optional[int] = None, kv_out=None,
                cu: Optional[torch.Tensor] = None, tail_rows: Optional[torch.Tensor] = None,
                weights_version=None) -> torch.Tensor:
        """x: fp32 [B*T, D] residual stream entering block 0; returns the stream leaving the last block.
        saved=None (inference) keeps nothing and updates x in place; otherwise x must be saved["xs"][0,0].
        kv_out = (kcache, vcache), each [L, B, Smax, D] 16-bit: every layer's keys / values of positions [0, T) are
        kept there (prefill of the KV-cached decode, see decode_step)."""
        geo = self.geo
        D, H = geo.width, geo.heads
        Hd = geo.hidden or 4 * D
        T = T or geo.tokens
        # tail_rows (int64 [R]): the caller only reads these rows of the stack's output (CLIP pools ONE token per sequence: the
        # class token / the EOT token).  The last block's attention still needs every token's keys and values, but its out-proj,
        # LayerNorm, fc and c_proj - 3/4 of a block's FLOPs - are then row-local work for R rows instead of M: they run on a
        # compact [R, D] copy and the compact stream [R, D] is RETURNED instead of [M, D] (backward takes the compact gradient).
        # cu (int32 [B+1]): PACKED batch - sequence b is rows [cu[b], cu[b+1]) of x, T = the longest length; every row-wise
        # kernel simply sees M = x.shape[0] rows, only the attention needs the row ranges
        M = x.shape[0] if cu is not None else B * T
        assert cu is None or (geo.head_dim == 64 and T <= 128 and kv_out is None)
        L = len(self.blocks)
        dev = x.device
        kc = geo.linear_layout
        train = saved is not None
        if train:
            assert x.data_ptr() == saved["xs"][0, 0].data_ptr() and saved["T"] == T and saved["M"] == M
            bf, xs, st, lse = saved["bf"], saved["xs"], saved["st"], saved["lse"]
            saved["key_keep"] = key_keep
            saved["cu"] = cu
            saved["tail"] = None
        else:
            bf = torch.empty(M, 6 * D + Hd, device=dev, dtype=self.dtype)  # xn | qkv | a | - | g  (reused per layer)
        # INFERENCE, LayerNorm folded away (SURVEY 2b "LayerNorm ... fused into the following GEMM"): the residual-writing GEMMs
        # (out-proj, c_proj) also emit a 16-bit copy of the new stream rows and per-row (sum, sum of squares) partials; the next
        # projection (fc, qkv of the next block) multiplies the RAW rows by W' = gamma (.) W and applies mean / rstd in its epilogue:
        # LN(x) W^T + b = rstd (x W'^T - mean colsum(W')) + (W beta + b).  24 of a ViT-B/32 tower's 25 standalone LayerNorm passes
        # (a read of the fp32 stream + a 16-bit write each) disappear; block 0's ln_1 and the pooled final LayerNorm stay.
        # Needs whole 256-row / 256-column tiles (tile configuration 8).  OFF by default (CCLIP_LN_FOLD=1 enables it): measured on
        # ViT-B/32 bs 1024 (profiles/r03_ln_fold_ab.txt) it is a wash with two lanes and 2 % SLOWER on one stream - the standalone
        # LayerNorm kernels run at the HBM roofline and overlap the other lane's GEMMs, while the bytes they save come back as extra
        # store instructions in the residual GEMMs' epilogue, which is store-issue bound and serial on its CU.
        import os
        fold = (not train and weights_version is not None and kc and cu is None and kv_out is None and not self.fp8
                and os.environ.get("CCLIP_LN_FOLD", "0") == "1" and geo.head_dim == 64 and geo.act in (ops.ACT_NONE, ops.ACT_QUICKGELU)
                and M % 256 == 0 and D % 256 == 0 and Hd % 256 == 0 and D >= 128)
        if fold:
            fw = self._fold_weights(weights_version)
            nblk = D // 64
            part = torch.empty(nblk * M * 2 + M * 2, device=dev, dtype=torch.float32)     # (per call: lanes run this stack concurrently)
            partials, stats = part[:nblk * M * 2].view(nblk, M, 2), part[nblk * M * 2:].view(M, 2)
            xb = torch.empty(M, D, device=dev, dtype=self.dtype)   # 16-bit copy of the current stream rows (same row stride as the fp32 stream: one ldc per GEMM)
        have_stats = False                                     # stats / xb describe the CURRENT x_in (ln_1's input)?
        for l, w in enumerate(self.blocks):
            duet.interleave_point()                            # (two towers launching side by side take turns block by block)
            if train:
                row = bf[l]
                xn1, qkv, a = row[:, 0:D], row[:, D:4 * D], row[:, 4 * D:5 * D]
                xn2, h, g = row[:, 5 * D:6 * D], row[:, 6 * D:6 * D + Hd], row[:, 6 * D + Hd:6 * D + 2 * Hd]
                x_in, x_mid = xs[l, 0], xs[l, 1]
                x_out = xs[l + 1, 0] if l + 1 < L else saved["out"]
                m1, r1, m2, r2 = st[l, 0], st[l, 1], st[l, 2], st[l, 3]
                lse_l = lse[l]
            else:
                xn1, qkv, a, g = bf[:, 0:D], bf[:, D:4 * D], bf[:, 4 * D:5 * D], bf[:, 6 * D:6 * D + Hd]
                xn2, h = xn1, None
                x_in = x_mid = x_out = x
                m1 = r1 = m2 = r2 = lse_l = None
            f8 = self.fp8 and not train and kc and D % 16 == 0
            if f8:
                if self._fp8_weights is None:
                    self.quantise_weights_fp8()
                if l == 0:
                    x8 = torch.empty(M, D, device=dev, dtype=torch.uint8)
                    sx = torch.empty(M, device=dev, dtype=torch.float32)
                    wide = self.fp8_wide and D % 128 == 0 and Hd % 128 == 0 and geo.act in (ops.ACT_NONE, ops.ACT_QUICKGELU)
                    if wide:
                        xmx = ops.mx_scale_buffer(M, D, dev)                              # block scales of the attention output
                        g8 = torch.empty(M, Hd, device=dev, dtype=torch.uint8)
                        gmx = ops.mx_scale_buffer(M, Hd, dev)
                wq8, swq = self._fp8_weights[l]["w_qkv"]
                ops.layernorm_fwd_fp8(x_in, w.ln1_w, w.ln1_b, x8, sx, rows=M)
                ops.gemm_fp8(x8, sx, wq8, swq, qkv, bias=w.b_qkv, M=M)
            elif fold and have_stats:
                wsq, c1q, c2q = fw[l]["q"]
                ops.gemm_bf16(xb, wsq, bias=c2q, out_bf16=qkv, M=M, ln_stats=stats, ln_c1=c1q)
            else:
                ops.layernorm_fwd(x_in, w.ln1_w, w.ln1_b, rows=M, out_bf16=xn1, mean=m1, rstd=r1)
                ops.gemm_bf16(xn1, w.w_qkv, b_kcontig=kc, bias=w.b_qkv, out_bf16=qkv, M=M)
            if kv_out is not None:       # data movement only
                kv_out[0][l, :B, :T].copy_(qkv[:M, D:2 * D].view(B, T, D))
                kv_out[1][l, :B, :T].copy_(qkv[:M, 2 * D:3 * D].view(B, T, D))
            if geo.head_dim == 64:
                a_mx = f8 and wide and H % 2 == 0 and not (tail_rows is not None and l == L - 1)   # the attention writes the out-proj's block-scaled e4m3 operand itself
                ops.attention_fwd(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], a, B=B, T=T, H=H, causal=geo.causal,
                                  key_keep=key_keep, lse=lse_l, out_mx=(x8, xmx) if a_mx else None, cu=cu)
            else:
                a_mx = False
                assert not geo.causal and key_keep is None
                ops.attention_small_fwd(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], a, B=B, T=T, H=H,
                                        head_dim=geo.head_dim, lse=lse_l)
            if tail_rows is not None and l == L - 1:
                R = tail_rows.numel()
                assert not a_mx
                a_c = a.index_select(0, tail_rows)
                xmid_c = torch.empty(R, D, device=dev, dtype=torch.float32)
                ops.gemm_bf16(a_c, w.w_o, b_kcontig=kc, bias=w.b_o, residual=x_in.index_select(0, tail_rows), out_f32=xmid_c, M=R, tile_config=1)
                tb = torch.empty(R, D + 2 * Hd, device=dev, dtype=self.dtype)            # xn2 | h | g of the kept rows
                xn2_c, h_c, g_c = tb[:, 0:D], tb[:, D:D + Hd], tb[:, D + Hd:D + 2 * Hd]
                st_c = torch.empty(2, R, device=dev, dtype=torch.float32)
                ops.layernorm_fwd(xmid_c, w.ln2_w, w.ln2_b, rows=R, out_bf16=xn2_c, mean=st_c[0], rstd=st_c[1])
                ops.gemm_bf16(xn2_c, w.w_fc, b_kcontig=kc, bias=w.b_fc, act=geo.act, out_bf16=g_c, out_pre=h_c if train else None, M=R, tile_config=1)
                xout_c = torch.empty(R, D, device=dev, dtype=torch.float32)
                ops.gemm_bf16(g_c, w.w_proj, b_kcontig=kc, bias=w.b_proj, residual=xmid_c, out_f32=xout_c, M=R, tile_config=1)
                if train:
                    saved["tail"] = dict(rows=tail_rows, a=a_c, xmid=xmid_c, xn2=xn2_c, h=h_c, g=g_c, st=st_c)
                return xout_c
            if f8 and wide:
                wo8, swo = self._fp8_weights[l]["w_o"]
                if not a_mx:
                    ops.quantize_mx_fp8(a, x8, xmx, rows=M)
                ops.gemm_fp8(x8, None, wo8, swo, bias=w.b_o, M=M, block_scale_a=xmx, out_f32=x_mid, residual=x_in, half=self.dtype)
            elif fold:
                ops.gemm_bf16(a, w.w_o, bias=w.b_o, residual=x_in, out_f32=x_mid, out_bf16=xb, M=M, rowstats_out=partials)
                ops.rowstats_combine(partials, stats, rows=M, D=D)
            else:
                ops.gemm_bf16(a, w.w_o, b_kcontig=kc, bias=w.b_o, residual=x_in, out_f32=x_mid, M=M)
            if f8 and geo.act in (ops.ACT_NONE, ops.ACT_QUICKGELU):
                wf8, swf = self._fp8_weights[l]["w_fc"]
                ops.layernorm_fwd_fp8(x_mid, w.ln2_w, w.ln2_b, x8, sx, rows=M)
                if wide:
                    ops.gemm_fp8(x8, sx, wf8, swf, bias=w.b_fc, act=geo.act, M=M, out_mx=(g8, gmx), half=self.dtype)
                else:
                    ops.gemm_fp8(x8, sx, wf8, swf, g, bias=w.b_fc, act=geo.act, M=M)
            elif fold:
                wsf, c1f, c2f = fw[l]["f"]
                ops.gemm_bf16(xb, wsf, bias=c2f, act=geo.act, out_bf16=g, M=M, ln_stats=stats, ln_c1=c1f)
            else:
                ops.layernorm_fwd(x_mid, w.ln2_w, w.ln2_b, rows=M, out_bf16=xn2, mean=m2, rstd=r2)
                ops.gemm_bf16(xn2, w.w_fc, b_kcontig=kc, bias=w.b_fc, act=geo.act, out_bf16=g, out_pre=h, M=M)
            if f8 and wide:
                wp8, swp = self._fp8_weights[l]["w_proj"]
                ops.gemm_fp8(g8, None, wp8, swp, bias=w.b_proj, M=M, block_scale_a=gmx, out_f32=x_out, residual=x_mid, half=self.dtype)
            elif fold and l + 1 < L:
                ops.gemm_bf16(g, w.w_proj, bias=w.b_proj, residual=x_mid, out_f32=x_out, out_bf16=xb, M=M, rowstats_out=partials)
                ops.rowstats_combine(partials, stats, rows=M, D=D)
                have_stats = True
            else:
                ops.gemm_bf16(g, w.w_proj, b_kcontig=kc, bias=w.b_proj, residual=x_mid, out_f32=x_out, M=M)
            x = x_out
        return x

    # ------------------------------------------------------------------ KV-cached decode
    def decode_step(self, x: torch.Tensor, kcache: torch.Tensor, vcache: torch.Tensor, pos: int) -> torch.Tensor:
        """One token per sequence: x fp32 [nb, D] is the residual stream of the token at position `pos` (updated in place and
        returned); kcache / vcache [L, nb, Smax, D] hold positions [0, pos) and receive this token's keys / values.
        Same arithmetic as forward() restricted to the last row of a causal sequence (no key padding)."""
        geo = self.geo
        D, H = geo.width, geo.heads
        Hd = geo.hidden or 4 * D
        assert geo.head_dim == 64 and geo.causal, "decode_step: GPT-2-shaped stacks only"
        nb = x.shape[0]
        kc = geo.linear_layout
        dev = x.device
        bf = torch.empty(nb, 5 * D + Hd, device=dev, dtype=self.dtype)          # xn | qkv | a | g
        xn, qkv, a, g = bf[:, 0:D], bf[:, D:4 * D], bf[:, 4 * D:5 * D], bf[:, 5 * D:5 * D + Hd]
        for l, w in enumerate(self.blocks):
            ops.layernorm_fwd(x, w.ln1_w, w.ln1_b, rows=nb, out_bf16=xn)
            ops.gemm_bf16(xn, w.w_qkv, b_kcontig=kc, bias=w.b_qkv, out_bf16=qkv, M=nb)
            kcache[l, :nb, pos].copy_(qkv[:, D:2 * D])
            vcache[l, :nb, pos].copy_(qkv[:, 2 * D:3 * D])
            ops.attention_decode(qkv[:, 0:D], kcache[l, :nb], vcache[l, :nb], a, H=H, S=pos + 1)
            ops.gemm_bf16(a, w.w_o, b_kcontig=kc, bias=w.b_o, residual=x, out_f32=x, M=nb)
            ops.layernorm_fwd(x, w.ln2_w, w.ln2_b, rows=nb, out_bf16=xn)
            ops.gemm_bf16(xn, w.w_fc, b_kcontig=kc, bias=w.b_fc, act=geo.act, out_bf16=g, M=nb)
            ops.gemm_bf16(g, w.w_proj, b_kcontig=kc, bias=w.b_proj, residual=x, out_f32=x, M=nb)
        return x

    # ------------------------------------------------------------------ backward
    def _wgrad(self, dy: torch.Tensor, xin: torch.Tensor, gw: torch.Tensor, M: int, acc: bool, scratch=None,
               gb: Optional[torch.Tensor] = None, acc_b: bool = False, fixed: bool = False, pad: bool = False):
        """gw (+)= dy^T xin for nn.Linear layout [out,in]; xin^T dy for Conv1D layout [in,out].
        gb: the layer's bias gradient (+)= column sums of dy: dy is an operand of this GEMM (A for nn.Linear weights, B for
        Conv1D ones), so the sums ride on the same launch - one extra MFMA per tile row / column against an all-ones fragment
        in the first block of tiles."""
        lin = self.geo.linear_layout
        a, b = (dy, xin) if lin else (xin, dy)
        n_out, k_in = gw.shape
        if fixed:
            # the compact tail of the last block (a batch's worth of rows): 128x128 tiles and the formula split, no table lookup and
            # no timing - these GEMMs are microseconds, and their row count is the batch size, which the table need not know
            sp = wgrad_splits(n_out, k_in, M)
            ws = (scratch or self.scratch).floats(sp * (n_out * k_in + max(n_out, k_in))) if sp > 1 else None
            ops.gemm_bf16(a[:M], b[:M], a_kcontig=False, b_kcontig=False, residual=gw if acc else None, out_f32=gw, tile_config=1,
                          split_k=sp, split_ws=ws, colsum_out=gb, colsum_accumulate=acc_b, colsum_of_b=not lin)
            return
        # both operands from this stack's zero-tailed slabs (alloc_saved / backward): contract over whole 64-row K-tiles
        Mk = (M + 63) // 64 * 64
        if not (pad and a.shape[0] >= Mk and b.shape[0] >= Mk):
            Mk = M
        ops.gemm_bf16(a[:Mk], b[:Mk], a_kcontig=False, b_kcontig=False, residual=gw if acc else None, out_f32=gw,
                      split_candidates=wgrad_candidates(n_out, k_in, Mk), scratch=(scratch or self.scratch).floats,
                      colsum_out=gb, colsum_accumulate=acc_b, colsum_of_b=not lin)

    def _bgrad(self, dy: torch.Tensor, gb: torch.Tensor, M: int, acc: bool, scratch=None):
        C = gb.numel()
        ws = (scratch or self.scratch).floats(ops.colsum_ws_floats(M, C))
        ops.colsum(dy, gb, ws, R=M, C=C, ld=dy.stride(0), accumulate=acc)

    def backward(self, dx: torch.Tensor, dxb: torch.Tensor, saved: dict, acc: Dict[int, bool]) -> torch.Tensor:
        """dx (fp32) / dxb (16-bit copy): gradient w.r.t. the stack output, [B*T, D] - or [R, D] on the kept rows when the forward
        ran with tail_rows.  dx is updated in place layer by layer; the fp32 gradient w.r.t. the stack input is left in
        saved["dx_in"] (the argument itself unless the tail was compact) and the matching 16-bit copy is RETURNED.
        acc[id(grad_tensor)] says whether that grad buffer already holds a gradient to add to.

        The dgrad chain (GEMM -> LN / attention backward -> GEMM ...) is latency-critical and half memory-bound; the
        weight/bias gradients hang off it as leaves.  They are therefore issued on a SIDE stream (every dY / dX
        copy gets its own buffer - 288 GB makes that free - so nothing is overwritten under a running wgrad): the
        long MFMA-bound wgrad launches fill the CUs while the chain's LN / attention / epilogue phases wait on HBM.
        CCLIP_WGRAD_STREAM=0 keeps everything on one stream."""
        import os
        geo = self.geo
        D, H = geo.width, geo.heads
        Hd = geo.hidden or 4 * D
        B, T = saved["B"], saved["T"]
        M = saved.get("M") or B * T
        Mp = saved.get("Mp") or M          # rows of the zero-tailed 16-bit slabs (whole 64-row K-tiles for the weight gradients)
        L = len(self.blocks)
        dev = dx.device
        kc = geo.linear_layout
        dact = _DACT[geo.act]
        bf, xs, st, lse = saved["bf"], saved["xs"], saved["st"], saved["lse"]
        trainable = any(b.grads is not None for b in self.blocks)
        if getattr(self, "refresh_transposed", None) is not None:
            self.refresh_transposed()            # rebuilt once per optimiser step (no-op while the shadows are unchanged)
        side = None
        if trainable and os.environ.get("CCLIP_WGRAD_STREAM", "1") == "1" and dev.type == "cuda":
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=dev)       # (default priority: raising either side was measured +8..15 %)
                self._side_scratch = Scratch(dev)
            side = self._side
        cur = torch.cuda.current_stream() if dev.type == "cuda" else None
        if side is not None:
            side.wait_stream(cur)
            dh_all = torch.empty(L, Mp, Hd, device=dev, dtype=self.dtype)
            dqkv_all = torch.empty(L, Mp, 3 * D, device=dev, dtype=self.dtype)
            dxb_all = torch.empty(2 * L, Mp, D, device=dev, dtype=self.dtype)
            tmp = torch.empty(M, D, device=dev, dtype=self.dtype)
            if Mp != M:                          # zero tail rows: see alloc_saved (operands of the weight-gradient GEMMs)
                dh_all[:, M:].zero_(); dqkv_all[:, M:].zero_(); dxb_all[:, M:].zero_()
        else:
            tmp8 = torch.empty(Mp, 4 * D + Hd, device=dev, dtype=self.dtype)      # dh | dqkv | dxn / da
            if Mp != M:
                tmp8[M:].zero_()
        ln_ws = self.scratch  # partial sums live in scratch; sized per call

        def leaf(fn):
            """run a parameter-gradient launch group where it belongs: after what the main stream has produced so far"""
            if side is None:
                fn(self.scratch)
                return
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                fn(self._side_scratch)

        for l in range(L - 1, -1, -1):
            duet.interleave_point()
            w = self.blocks[l]
            gr = w.grads
            row = bf[l]
            xn1, qkv, a = row[:, 0:D], row[:, D:4 * D], row[:, 4 * D:5 * D]
            xn2, h, g = row[:, 5 * D:6 * D], row[:, 6 * D:6 * D + Hd], row[:, 6 * D + Hd:6 * D + 2 * Hd]
            x_in, x_mid = xs[l, 0], xs[l, 1]
            m1, r1, m2, r2 = st[l, 0], st[l, 1], st[l, 2], st[l, 3]
            if side is not None:
                dh, dqkv, dsm = dh_all[l], dqkv_all[l], tmp
                dxb_mid, dxb_in = dxb_all[2 * l + 1], dxb_all[2 * l]
            else:
                dh, dqkv, dsm = tmp8[:, 0:Hd], tmp8[:, Hd:Hd + 3 * D], tmp8[:, Hd + 3 * D:Hd + 4 * D]
                dxb_mid = dxb_in = dxb

            def A(name, gr=gr):
                return acc.get(id(gr[name]), False)

            def wd(name, w=w):
                """B operand of a dgrad GEMM dX = dY . W: the transposed shadow (K-contiguous, forward layout) when the arena
                keeps one, else the weight itself read K-strided ([out, in]) / K-contiguously (Conv1D [in, out])"""
                if w.wt is not None:
                    return w.wt[name], True
                return getattr(w, name), not kc

            tail = saved.get("tail") if l == L - 1 else None
            if tail is not None:
                # the last block's MLP and out-proj ran on the R kept rows only (forward(tail_rows=...)): dx / dxb arrive compact
                # [R, D]; their backward is compact too, and the gradient re-enters the full-width stream at the attention output
                # (d a) and at the residual stream (d x_in) of the kept rows
                R = tail["rows"].numel()
                a_c, xmid_c, xn2_c, h_c, g_c, st_c = tail["a"], tail["xmid"], tail["xn2"], tail["h"], tail["g"], tail["st"]
                dh_c = torch.empty(R, Hd, device=dev, dtype=self.dtype)
                ds_c = torch.empty(R, D, device=dev, dtype=self.dtype)
                dxb_mid_c = torch.empty(R, D, device=dev, dtype=self.dtype)
                if gr is not None:
                    def t1(sc, dxb=dxb, g_c=g_c, gr=gr):
                        self._wgrad(dxb, g_c, gr["w_proj"], R, A("w_proj", gr), sc, gr["b_proj"], A("b_proj", gr), fixed=True)
                    leaf(t1)
                ops.gemm_bf16(dxb, wd("w_proj")[0], b_kcontig=wd("w_proj")[1], act=dact, aux=h_c, out_bf16=dh_c, M=R, tile_config=1)
                if gr is not None:
                    def t2(sc, dh_c=dh_c, xn2_c=xn2_c, gr=gr):
                        self._wgrad(dh_c, xn2_c, gr["w_fc"], R, A("w_fc", gr), sc, gr["b_fc"], A("b_fc", gr), fixed=True)
                    leaf(t2)
                ops.gemm_bf16(dh_c, wd("w_fc")[0], b_kcontig=wd("w_fc")[1], out_bf16=ds_c, M=R, tile_config=1)
                ws = ln_ws.floats(ops.layernorm_bwd_ws_floats(R, D)) if gr is not None else None
                ops.layernorm_bwd(ds_c, xmid_c, w.ln2_w, st_c[0], st_c[1], rows=R, dx_res=dx, dx_out=dx, dx_out_bf16=dxb_mid_c,
                                  dgamma=gr["ln2_w"] if gr is not None else None, dbeta=gr["ln2_b"] if gr is not None else None,
                                  accumulate=A("ln2_w") if gr is not None else False, ws=ws)
                if gr is not None:
                    def t3(sc, dxb_mid_c=dxb_mid_c, a_c=a_c, gr=gr):
                        self._wgrad(dxb_mid_c, a_c, gr["w_o"], R, A("w_o", gr), sc, gr["b_o"], A("b_o", gr), fixed=True)
                    leaf(t3)
                ops.gemm_bf16(dxb_mid_c, wd("w_o")[0], b_kcontig=wd("w_o")[1], out_bf16=ds_c, M=R, tile_config=1)      # d a of the kept rows
                dsm.zero_()
                dsm.index_copy_(0, tail["rows"], ds_c)
                dx_c = dx                                                 # d x_mid of the kept rows = their share of d x_in
                dx = torch.zeros(M, D, device=dev, dtype=torch.float32)
                dx.index_copy_(0, tail["rows"], dx_c)
                if side is None:
                    dxb_in = torch.empty(Mp, D, device=dev, dtype=self.dtype)   # (one-stream mode reuses the caller's dxb, which is compact here)
                    if Mp != M:
                        dxb_in[M:].zero_()
            else:
                # ---- MLP branch ----
                if gr is not None:
                    dxb_is_slab = l != L - 1        # (the stack output's gradient copy belongs to the caller: no zero tail promised)

                    def f1(sc, dxb=dxb, g=g, gr=gr, dxb_is_slab=dxb_is_slab):
                        self._wgrad(dxb, g, gr["w_proj"], M, A("w_proj", gr), sc, gr["b_proj"], A("b_proj", gr), pad=dxb_is_slab)
                    leaf(f1)
                ops.gemm_bf16(dxb, wd("w_proj")[0], b_kcontig=wd("w_proj")[1], act=dact, aux=h, out_bf16=dh, M=M)
                if gr is not None:
                    def f2(sc, dh=dh, xn2=xn2, gr=gr):
                        self._wgrad(dh, xn2, gr["w_fc"], M, A("w_fc", gr), sc, gr["b_fc"], A("b_fc", gr), pad=True)
                    leaf(f2)
                ops.gemm_bf16(dh, wd("w_fc")[0], b_kcontig=wd("w_fc")[1], out_bf16=dsm, M=M)
                ws = ln_ws.floats(ops.layernorm_bwd_ws_floats(M, D)) if gr is not None else None
                ops.layernorm_bwd(dsm, x_mid, w.ln2_w, m2, r2, rows=M, dx_res=dx, dx_out=dx, dx_out_bf16=dxb_mid,
                                  dgamma=gr["ln2_w"] if gr is not None else None, dbeta=gr["ln2_b"] if gr is not None else None,
                                  accumulate=A("ln2_w") if gr is not None else False, ws=ws)
                dxb = dxb_mid
                # ---- attention branch ----
                if gr is not None:
                    def f3(sc, dxb=dxb, a=a, gr=gr):
                        self._wgrad(dxb, a, gr["w_o"], M, A("w_o", gr), sc, gr["b_o"], A("b_o", gr), pad=True)
                    leaf(f3)
                ops.gemm_bf16(dxb, wd("w_o")[0], b_kcontig=wd("w_o")[1], out_bf16=dsm, M=M)
            if geo.head_dim == 64:
                ops.attention_bwd(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], a, lse[l], dsm, dqkv[:, 0:D],
                                  dqkv[:, D:2 * D], dqkv[:, 2 * D:3 * D], B=B, T=T, H=H, causal=geo.causal,
                                  key_keep=saved["key_keep"], cu=saved.get("cu"))
            else:
                ops.attention_small_bwd(qkv[:, 0:D], qkv[:, D:2 * D], qkv[:, 2 * D:3 * D], a, lse[l], dsm, dqkv[:, 0:D],
                                        dqkv[:, D:2 * D], dqkv[:, 2 * D:3 * D], B=B, T=T, H=H, head_dim=geo.head_dim)
            if gr is not None:
                def f4(sc, dqkv=dqkv, xn1=xn1, gr=gr):
                    gbq = gr.get("b_qkv")                    # TransformerMapper's q / kv projections have no bias
                    self._wgrad(dqkv, xn1, gr["w_qkv"], M, A("w_qkv", gr), sc, gbq, A("b_qkv", gr) if gbq is not None else False, pad=True)
                leaf(f4)
            ops.gemm_bf16(dqkv, wd("w_qkv")[0], b_kcontig=wd("w_qkv")[1], out_bf16=dsm, M=M)
            ws = ln_ws.floats(ops.layernorm_bwd_ws_floats(M, D)) if gr is not None else None
            ops.layernorm_bwd(dsm, x_in, w.ln1_w, m1, r1, rows=M, dx_res=dx, dx_out=dx, dx_out_bf16=dxb_in,
                              dgamma=gr["ln1_w"] if gr is not None else None, dbeta=gr["ln1_b"] if gr is not None else None,
                              accumulate=A("ln1_w") if gr is not None else False, ws=ws)
            dxb = dxb_in
            hook = getattr(self, "grad_hook", None) if getattr(self, "grad_hook_enabled", True) else None
            if hook is not None and gr is not None:
                # this layer's last gradient kernels are enqueued (weights on the side stream, LayerNorm affines on the main
                # one): a data-parallel reducer may start this layer's all-reduce now, under the remaining layers' backward
                hook([t for t in gr.values() if t is not None], [s_ for s_ in (cur, side) if s_ is not None])
        saved["dx_in"] = dx              # (with a compact tail the full-width fp32 gradient is a new tensor, not the argument)
        if side is not None:
            cur.wait_stream(side)        # every parameter gradient is complete before anyone downstream looks at it
        return dxb[:M] if dxb.shape[0] != M and dxb.shape[0] == Mp else dxb   # (a zero-tailed slab row: hand back its M live rows)
