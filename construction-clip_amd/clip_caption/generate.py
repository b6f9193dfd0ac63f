"""Caption decoding with a KV cache: `generate_beam` and `generate2` with the reference's signatures and token-for-token
semantics (/root/reference/CLIP_prefix_caption/test.py:353-514, application.py:152-229, predict.py:164-300).

The reference calls `model.gpt(inputs_embeds=generated)` on the whole growing sequence at every step and keeps the last
position's logits.  Here the prefix is run once (prefill) and every later step feeds ONE token per beam through
`BlockStack.decode_step` against the cached keys / values; beam reordering gathers the cache.  The selection
arithmetic on the [beams, V] logits (temperature, softmax-log, length-normalised top-k, nucleus filter) is the
reference's, restated.  For GPT-2 geometry `generate_beam` runs all of it - decode steps AND selection - inside one persistent
kernel launch (ClipCaptionModel.beam_search_native); the torch loop below is the same arithmetic and its parity reference.

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
    scores = None
    seq_lengths = torch.ones(beam_size, device=device)
    is_stopped = torch.zeros(beam_size, device=device, dtype=torch.bool)
    if embed is not None:
        generated = embed
    else:
        tokens = torch.tensor(tokenizer.encode(prompt)).unsqueeze(0).to(device)
        generated = model.gpt.transformer.wte(tokens)
    cache = None
    step_in = generated                                   # prefill: the whole prefix; afterwards one token per beam
    for step in range(entry_length):
        logits, cache = _step_logits(model, step_in, cache)
        logits = logits / (temperature if temperature > 0 else 1.0)
        logits = logits.softmax(-1).log()
        if scores is None:
            scores, next_tokens = logits.topk(beam_size, -1)
            cache = cache.expand(beam_size)
            next_tokens, scores = next_tokens.permute(1, 0), scores.squeeze(0)
            if tokens is None:
                tokens = next_tokens
            else:
                tokens = torch.cat((tokens.expand(beam_size, *tokens.shape[1:]), next_tokens), dim=1)
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
            cache = cache.reorder(next_tokens_source)
            scores = scores_sum_average * seq_lengths
            is_stopped = is_stopped[next_tokens_source]
        step_in = model.gpt.transformer.wte(next_tokens.squeeze(1)).view(beam_size, 1, -1)
        is_stopped = is_stopped + next_tokens.eq(stop_token).squeeze(1)
        # `is_stopped.all()` is a device -> host sync; a stopped beam only ever appends token 0 at score 0 and keeps its length,
        # so looking every 4th step (and on the last) returns the same texts, lengths and scores while the host runs ahead
        if (step & 3) == 3 or step == entry_length - 1:
            if is_stopped.all():
                break
    return _beam_outputs(tokenizer, tokens, seq_lengths, scores, return_tokens)


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
        cache = None
        for _ in range(entry_length):
            logits, cache = _step_logits(model, step_in, cache)
            logits = logits / (temperature if temperature > 0 else 1.0)
            sorted_logits, sorted_indices = torch.sort(logits, descending=True)
            cumulative_probs = torch.cumsum(torch.softmax(sorted_logits, dim=-1), dim=-1)
            remove = cumulative_probs > top_p
            remove[..., 1:] = remove[..., :-1].clone()
            remove[..., 0] = 0
            logits[:, sorted_indices[remove]] = -float("inf")
            next_token = torch.argmax(logits, -1).unsqueeze(0)
            out_tokens = next_token if out_tokens is None else torch.cat((out_tokens, next_token), dim=1)
            step_in = model.gpt.transformer.wte(next_token)
            if stop_token == next_token.item():
                break
        generated_list.append(tokenizer.decode(list(out_tokens.squeeze(0).cpu().numpy())))
    if return_tokens:
        return generated_list[0], out_tokens
    return generated_list[0]
