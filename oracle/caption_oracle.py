"""CPU fp32 ORACLE for the prefix-caption path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.

Restates
  /root/reference/CLIP_prefix_caption/train.py:110-123   MLP mapper (Linear-Tanh-Linear)
  /root/reference/CLIP_prefix_caption/train.py:126-248   TransformerMapper (+MultiHeadAttention, TransformerLayer)
  /root/reference/CLIP_prefix_caption/train.py:256-269   ClipCaptionModel.forward
  /root/reference/CLIP_prefix_caption/train.py:356-357   logits slice + CE(ignore_index=0)
and the GPT-2 arithmetic that `self.model = GPT2LMHeadModel.from_pretrained(...)`
(train.py:275) contributes, which lives in the un-pinned third-party
`transformers` package: restated here from its published algorithm and pinned
against the local `transformers.GPT2LMHeadModel(GPT2Config(...))` (config-only,
random init, no fetch) in oracle/hf_crosscheck.py.  HF:<file>:<line> cites are
transformers 5.x models/gpt2/modeling_gpt2.py.

State-dict key layout = what `torch.save(model.state_dict())` at train.py:371-381
produces: `clip_project.model.{0,2}.{weight,bias}` + `model.transformer.*` + `model.lm_head.weight`.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


def gelu_new(x: Tensor) -> Tensor:
    # HF:activations.py NewGELUActivation (GPT-2 default activation_function="gelu_new")
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def mlp_mapper(sd: SD, prefix: Tensor, p: str = "clip_project.") -> Tensor:
    """train.py:110-123 with sizes (512, 768*P/2, 768*P): Linear -> Tanh -> Linear."""
    h = torch.tanh(prefix.float() @ sd[p + "model.0.weight"].float().t() + sd[p + "model.0.bias"].float())
    return h @ sd[p + "model.2.weight"].float().t() + sd[p + "model.2.bias"].float()


def transformer_mapper(sd: SD, prefix: Tensor, clip_length: int, num_heads: int = 8,
                       p: str = "clip_project.") -> Tensor:
    """train.py:233-248 TransformerMapper.forward: linear -> view [B, clip_length, D];
    cat learned prefix_const; pre-LN layers (bias-free q / kv projections, ReLU MLP ratio 2,
    train.py:141-207); return the last prefix_length tokens."""
    x = prefix.float() @ sd[p + "linear.weight"].float().t() + sd[p + "linear.bias"].float()
    b = x.shape[0]
    x = x.view(b, clip_length, -1)
    const = sd[p + "prefix_const"].float()
    x = torch.cat((x, const.unsqueeze(0).expand(b, *const.shape)), dim=1)
    d = x.shape[-1]
    n_layers = len({k.split(".")[3] for k in sd if k.startswith(p + "transformer.layers.")})
    for i in range(n_layers):
        q = f"{p}transformer.layers.{i}."
        h = F.layer_norm(x, (d,), sd[q + "norm1.weight"].float(), sd[q + "norm1.bias"].float(), 1e-5)
        n = h.shape[1]
        dh = d // num_heads
        queries = (h @ sd[q + "attn.to_queries.weight"].float().t()).reshape(b, n, num_heads, dh)
        kv = (h @ sd[q + "attn.to_keys_values.weight"].float().t()).reshape(b, n, 2, num_heads, dh)
        keys, values = kv[:, :, 0], kv[:, :, 1]
        att = torch.einsum("bnhd,bmhd->bnmh", queries, keys) * (dh ** -0.5)   # train.py:164
        att = att.softmax(dim=2)
        o = torch.einsum("bnmh,bmhd->bnhd", att, values).reshape(b, n, d)      # train.py:170
        o = o @ sd[q + "attn.project.weight"].float().t() + sd[q + "attn.project.bias"].float()
        x = x + o
        h = F.layer_norm(x, (d,), sd[q + "norm2.weight"].float(), sd[q + "norm2.bias"].float(), 1e-5)
        h = torch.relu(h @ sd[q + "mlp.fc1.weight"].float().t() + sd[q + "mlp.fc1.bias"].float())
        h = h @ sd[q + "mlp.fc2.weight"].float().t() + sd[q + "mlp.fc2.bias"].float()
        x = x + h
    return x[:, clip_length:]


def gpt2_forward(sd: SD, inputs_embeds: Tensor, attention_mask: Optional[Tensor], n_head: int,
                 p: str = "model.") -> Tensor:
    """GPT2LMHeadModel(inputs_embeds=..., attention_mask=...).logits (call: train.py:268).
    pre-LN blocks, Conv1D weights stored [in, out] (HF:pytorch_utils.py:95-117), packed c_attn,
    causal + additive key-padding mask (HF:modeling_gpt2.py:103-107), gelu_new, tied lm_head."""
    x = inputs_embeds.float()
    b, s, d = x.shape
    dh = d // n_head
    x = x + sd[p + "transformer.wpe.weight"].float()[:s]
    causal = torch.full((s, s), float("-inf")).triu_(1)
    if attention_mask is not None:
        pad = (1.0 - attention_mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    else:
        pad = None
    n_layer = len({k.split(".")[3] for k in sd if k.startswith(p + "transformer.h.")})
    for i in range(n_layer):
        q = f"{p}transformer.h.{i}."
        h = F.layer_norm(x, (d,), sd[q + "ln_1.weight"].float(), sd[q + "ln_1.bias"].float(), 1e-5)
        qkv = h @ sd[q + "attn.c_attn.weight"].float() + sd[q + "attn.c_attn.bias"].float()
        qq, kk, vv = qkv.split(d, dim=-1)
        qq = qq.view(b, s, n_head, dh).transpose(1, 2)
        kk = kk.view(b, s, n_head, dh).transpose(1, 2)
        vv = vv.view(b, s, n_head, dh).transpose(1, 2)
        sc = (qq @ kk.transpose(-1, -2)) / math.sqrt(dh) + causal
        if pad is not None:
            sc = sc + pad
        a = (torch.softmax(sc, dim=-1) @ vv).transpose(1, 2).reshape(b, s, d)
        x = x + a @ sd[q + "attn.c_proj.weight"].float() + sd[q + "attn.c_proj.bias"].float()
        h = F.layer_norm(x, (d,), sd[q + "ln_2.weight"].float(), sd[q + "ln_2.bias"].float(), 1e-5)
        h = gelu_new(h @ sd[q + "mlp.c_fc.weight"].float() + sd[q + "mlp.c_fc.bias"].float())
        x = x + h @ sd[q + "mlp.c_proj.weight"].float() + sd[q + "mlp.c_proj.bias"].float()
    x = F.layer_norm(x, (d,), sd[p + "transformer.ln_f.weight"].float(), sd[p + "transformer.ln_f.bias"].float(), 1e-5)
    return x @ sd[p + "transformer.wte.weight"].float().t()


def caption_forward(sd: SD, tokens: Tensor, prefix: Tensor, attribute: Tensor, mask: Optional[Tensor],
                    prefix_length: int, n_head: int = 12, clip_length: Optional[int] = None) -> Tensor:
    """ClipCaptionModel.forward (train.py:256-269): returns logits [B, P+A+L, V].  The mapper is the MLP unless the
    state_dict holds a TransformerMapper (then clip_length must be given)."""
    wte = sd["model.transformer.wte.weight"].float()
    emb_text = wte[torch.cat((attribute, tokens), dim=1).long()]
    if "clip_project.prefix_const" in sd:
        pre = transformer_mapper(sd, prefix, clip_length)
    else:
        pre = mlp_mapper(sd, prefix).view(-1, prefix_length, wte.shape[1])
    return gpt2_forward(sd, torch.cat((pre, emb_text), dim=1), mask, n_head)


def caption_loss(logits: Tensor, tokens: Tensor, prefix_length: int, attribute_length: int) -> Tensor:
    """train.py:356-357: logits[:, P+A-1:-1] vs tokens, CE with ignore_index=0."""
    lg = logits[:, prefix_length + attribute_length - 1: -1]
    return F.cross_entropy(lg.reshape(-1, lg.shape[-1]), tokens.flatten().long(), ignore_index=0)


# ---- decoding (test.py:353-514), restated with a FULL forward per step exactly as the reference runs it -----------
def _last_logits(sd: SD, generated: Tensor, n_head: int) -> Tensor:
    return gpt2_forward(sd, generated, None, n_head)[:, -1, :]


@torch.no_grad()
def generate_beam_tokens(sd: SD, embed: Tensor, n_head: int, beam_size: int = 3, entry_length: int = 100,
                         temperature: float = 0.5, stop_token: int = 102):
    """generate_beam (test.py:353-441) up to the token / length / score tensors (no tokenizer).  Returns
    (tokens [beams, steps], seq_lengths, scores / seq_lengths, per-step last-position logits of beam 0)."""
    wte = sd["model.transformer.wte.weight"].float()
    tokens, scores = None, None
    seq_lengths = torch.ones(beam_size)
    is_stopped = torch.zeros(beam_size, dtype=torch.bool)
    generated = embed.float()
    trace = []
    for _ in range(entry_length):
        raw = _last_logits(sd, generated, n_head)
        trace.append(raw.clone())
        logits = raw / (temperature if temperature > 0 else 1.0)
        logits = logits.softmax(-1).log()
        if scores is None:
            scores, next_tokens = logits.topk(beam_size, -1)
            generated = generated.expand(beam_size, *generated.shape[1:])
            next_tokens, scores = next_tokens.permute(1, 0), scores.squeeze(0)
            tokens = next_tokens
        else:
            logits[is_stopped] = -float("inf")
            logits[is_stopped, 0] = 0
            scores_sum = scores[:, None] + logits
            seq_lengths[~is_stopped] += 1
            scores_sum_average = scores_sum / seq_lengths[:, None]
            scores_sum_average, next_tokens = scores_sum_average.view(-1).topk(beam_size, -1)
            next_tokens_source = next_tokens // scores_sum.shape[1]
            seq_lengths = seq_lengths[next_tokens_source]
            next_tokens = (next_tokens % scores_sum.shape[1]).unsqueeze(1)
            tokens = torch.cat((tokens[next_tokens_source], next_tokens), dim=1)
            generated = generated[next_tokens_source]
            scores = scores_sum_average * seq_lengths
            is_stopped = is_stopped[next_tokens_source]
        generated = torch.cat((generated, wte[next_tokens.squeeze(1)].view(generated.shape[0], 1, -1)), dim=1)
        is_stopped = is_stopped + next_tokens.eq(stop_token).squeeze(1)
        if is_stopped.all():
            break
    return tokens, seq_lengths, scores / seq_lengths, trace


@torch.no_grad()
def generate2_tokens(sd: SD, embed: Tensor, n_head: int, entry_length: int = 67, top_p: float = 0.8,
                     temperature: float = 1.0, stop_token: int = 102):
    """generate2 (test.py:443-514): nucleus filter, then arg-max.  Returns (tokens [1, steps], per-step logits)."""
    wte = sd["model.transformer.wte.weight"].float()
    generated = embed.float()
    tokens, trace = None, []
    for _ in range(entry_length):
        raw = _last_logits(sd, generated, n_head)
        trace.append(raw.clone())
        logits = raw / (temperature if temperature > 0 else 1.0)
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cumulative_probs = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cumulative_probs > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[:, sorted_indices[remove]] = -float("inf")
        next_token = torch.argmax(logits, -1).unsqueeze(0)
        tokens = next_token if tokens is None else torch.cat((tokens, next_token), dim=1)
        generated = torch.cat((generated, wte[next_token]), dim=1)
        if stop_token == next_token.item():
            break
    return tokens, trace
