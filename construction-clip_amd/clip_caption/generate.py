"""Caption decoding with a KV cache: `generate_beam` and `generate2` with the reference's signatures and token-for-token
semantics (/root/reference/CLIP_prefix_caption/test.py:353-514, application.py:152-229, predict.py:164-300).

The reference calls `model.gpt(inputs_embeds=generated)` on the whole growing sequence at every step and keeps the last
position's logits.  Here the prefix is run once (prefill) and every later step feeds ONE token per beam through
`BlockStack.decode_step` against the cached keys / values; beam reordering gathers the cache.  The selection
rule on the [beams, V] logits (temperature, softmax-log, length-normalised top-k; a nucleus filter that cannot move the
arg-max) is the reference's.  For GPT-2 geometry `generate_beam` / `generate2` run all of it - decode steps AND selection - inside
one persistent kernel launch (ClipCaptionModel.beam_search_native); the host-side search below (class _Beams) covers what that
kernel does not.  The reference's loops themselves are restated only in oracle/caption_oracle.py, the checker.

Not carried over: the attention-map dump that the reference's test.py copy of generate_beam interleaves with decoding
(`output_attentions=True`, test.py:381-390, `attention_map(...)` :438) - visualisation, SURVEY.md section 8 out of scope.
"""
from __future__ import annotations

from typing import List, Optional

import torch


def _step_logits(model, embeds: torch.Tensor, cache):
    out = model.gpt(inputs_embeds=embeds, past_key_values=cache, use_cache=True)
    return out.logits[:, -1, :], out.past_key_values


@torch.no_grad()
def generate_beam(model, tokenizer, beam_size: int = 3, prompt=None, embed=None, entry_length: int = 100,
                  temperature: float = 0.5, stop_token: int = 102, return_tokens: bool = False):
    """Beam search over length-normalised log-probabilities (test.py:353-441).  Returns the decoded texts best-first;
    with return_tokens also (token tensor [beams, steps], lengths, scores) in beam order."""
    model.eval()
    device = next(model.parameters()).device
    tokens = None
    if getattr(model, "beam_native_ok", None) is not None and model.beam_native_ok(beam_size) and entry_length >= 1:
        # prefill, then ONE persistent kernel for every decode step and every selection (csrc/decode_persist.hip); the
        # host loop below is the same arithmetic op by op and stays as its parity reference (CCLIP_BEAM_NATIVE=0)
        if embed is not None:
            generated = embed
        else:
            tokens = torch.tensor(tokenizer.encode(prompt)).unsqueeze(0).to(device)
            generated = model.gpt.transformer.wte(tokens)
        tokens, seq_lengths, scores = model.beam_search_native(generated, beam_size, entry_length, temperature, stop_token,
                                                               prompt_tokens=tokens)
        return _beam_outputs(tokenizer, tokens, seq_lengths, scores, return_tokens)
    # Host-side search (nn.Linear-layout stacks, > 8 beams, CPU stubs in the tests, CCLIP_BEAM_NATIVE=0): one KV-cached decode
    # step per position, the beam bookkeeping kept in a _Beams record.  Same selection rule as the reference (a stopped beam
    # may only extend by token 0 at no cost; candidates ranked by total log-probability / length), written as one masked
    # candidate table per step instead of the reference's in-place edits.
    if embed is not None:
        prefix = embed
    else:
        tokens = torch.tensor(tokenizer.encode(prompt)).unsqueeze(0).to(device)
        prefix = model.gpt.transformer.wte(tokens)
    inv_t = 1.0 / (temperature if temperature > 0 else 1.0)
    logp, cache = _step_logits(model, prefix, None)                     # prefill: the whole prefix once
    logp = (logp * inv_t).softmax(-1).log()                             # (softmax-then-log, as the device kernels do)
    beams = _Beams.start(logp, beam_size, tokens)
    cache = cache.expand(beam_size)
    for step in range(1, entry_length):
        step_in = model.gpt.transformer.wte(beams.last_token()).view(beam_size, 1, -1)
        logp, cache = _step_logits(model, step_in, cache)
        logp = (logp * inv_t).softmax(-1).log()
        parent = beams.extend(logp, stop_token)
        cache = cache.reorder(parent)
        # `all stopped` is a device -> host sync; a stopped beam only ever appends token 0 at score 0 and keeps its length, so
        # looking every 4th step (and on the last) returns the same texts, lengths and scores while the host runs ahead
        if ((step & 3) == 3 or step == entry_length - 1) and bool((beams.stopped | beams.last_token().eq(stop_token)).all()):
            break
    return _beam_outputs(tokenizer, beams.tokens, beams.lengths, beams.total, return_tokens)


class _Beams:
    """Beam bookkeeping of the host-side search: token rows, lengths (floats, as the scores divide by them), total
    log-probabilities and the stopped flags, all [beams]-shaped device tensors."""

    def __init__(self, tokens, lengths, total, stopped):
        self.tokens, self.lengths, self.total, self.stopped = tokens, lengths, total, stopped

    @classmethod
    def start(cls, logp, k: int, prompt_tokens):
        """first position: the k most probable tokens of the single prefix row open the beams"""
        total, first = logp[0].topk(k)
        col = first.unsqueeze(1)
        toks = col if prompt_tokens is None else torch.cat((prompt_tokens.expand(k, -1), col), dim=1)
        b = cls(toks, torch.ones(k, device=logp.device), total, torch.zeros(k, dtype=torch.bool, device=logp.device))
        return b

    def last_token(self):
        return self.tokens[:, -1]

    def extend(self, logp, stop_token: int):
        """one position further: returns each new beam's parent row (for the KV-cache reorder)"""
        k, V = logp.shape
        # a beam stops on the position AFTER it emitted the stop token
        self.stopped = self.stopped | self.tokens[:, -1].eq(stop_token)
        live = ~self.stopped
        # candidate table: live rows extend by any token; a stopped row offers only token 0, free of charge
        step_cost = torch.where(live[:, None], logp, torch.full_like(logp, -float("inf")))
        step_cost[self.stopped, 0] = 0.0
        new_len = self.lengths + live.to(self.lengths.dtype)
        ranked = (self.total[:, None] + step_cost) / new_len[:, None]
        best, flat = ranked.reshape(-1).topk(k)
        parent = torch.div(flat, V, rounding_mode="floor")
        token = flat - parent * V
        self.lengths = new_len[parent]
        self.total = best * self.lengths
        self.stopped = self.stopped[parent]
        self.tokens = torch.cat((self.tokens[parent], token.unsqueeze(1)), dim=1)
        return parent


def _beam_outputs(tokenizer, tokens, seq_lengths, scores, return_tokens: bool):
    """test.py:435-441: length-normalise, order best first, decode each beam up to its length"""
    scores = scores / seq_lengths
    order = scores.argsort(descending=True)
    output_list = tokens.cpu().numpy()
    texts = [tokenizer.decode(output[: int(length)]) for output, length in zip(output_list, seq_lengths)]
    texts = [texts[i] for i in order]
    if return_tokens:
        return texts, tokens, seq_lengths, scores
    return texts


@torch.no_grad()
def generate2(model, tokenizer, tokens=None, prompt=None, embed=None, entry_count: int = 1, entry_length: int = 67,
              top_p: float = 0.8, temperature: float = 1.0, stop_token: int = 102, return_tokens: bool = False):
    """Nucleus-filtered greedy decoding (test.py:443-514): tokens outside the top-p mass are removed, the arg-max of the
    rest is taken.  As in the reference's live code the sequence always starts from `embed` (test.py:472)."""
    model.eval()
    device = next(model.parameters()).device
    generated_list: List[str] = []
    out_tokens: Optional[torch.Tensor] = tokens
    for _ in range(entry_count):
        if embed is None:
            if out_tokens is None:
                out_tokens = torch.tensor(tokenizer.encode(prompt)).unsqueeze(0).to(device)
            step_in = model.gpt.transformer.wte(out_tokens)
        else:
            step_in = embed
        if getattr(model, "beam_native_ok", None) is not None and model.beam_native_ok(1) and entry_length >= 1 and top_p > 0:
            # The nucleus filter never removes the most probable token (`remove[..., 0] = 0`, test.py:499), so the arg-max of the
            # filtered logits IS the arg-max of the logits: this loop is a greedy search = the persistent beam kernel with one
            # beam, which stops on the stop token exactly as the loop's `break` does (the stop token is the last one kept).
            prev = out_tokens
            new_tokens, _, _ = model.beam_search_native(step_in, 1, entry_length, temperature, stop_token,
                                                        prompt_tokens=prev if embed is None else None)
            # (from a prompt the kernel's token row starts with the prompt; from `embed` the running token list is kept in front)
            out_tokens = new_tokens if embed is None or prev is None else torch.cat((prev.to(new_tokens.device), new_tokens), dim=1)
            generated_list.append(tokenizer.decode(list(out_tokens.squeeze(0).cpu().numpy())))
            continue
        # host-side loop (stacks the persistent kernel does not cover, CPU stubs): the same observation makes the sort / cumulative
        # sum of the nucleus filter unnecessary for the arg-max - one KV-cached step and one arg-max per position
        cache = None
        for _ in range(entry_length):
            logits, cache = _step_logits(model, step_in, cache)
            pick = logits.argmax(dim=-1, keepdim=True)                  # [1, 1]; the temperature does not move an arg-max
            out_tokens = pick if out_tokens is None else torch.cat((out_tokens, pick), dim=1)
            if int(pick) == stop_token:
                break
            step_in = model.gpt.transformer.wte(pick)
        generated_list.append(tokenizer.decode(list(out_tokens.squeeze(0).cpu().numpy())))
    if return_tokens:
        return generated_list[0], out_tokens
    return generated_list[0]
