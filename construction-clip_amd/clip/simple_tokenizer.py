"""Byte-level BPE tokenizer with the behaviour of openai/CLIP's `SimpleTokenizer` (used through
`clip.tokenize` at /root/reference/CLIP/train.py:60, /root/reference/CLIP/predict.py:40,
/root/reference/CLIP_prefix_caption/parse_coco.py:29-30, and directly at /root/reference/attention.py:114).

The merges file (`bpe_simple_vocab_16e6.txt.gz`, 48 894 merges used) ships with the `clip` package, which is
not present offline: point CCLIP_BPE_PATH (or the `bpe_path` argument) at a copy.
The algorithm below is the published one: bytes -> printable unicode, words split by the CLIP
regex, greedy lowest-rank pair merging with an `</w>` end-of-word marker, vocab =
256 byte symbols + 256 `</w>` variants + merges + <|startoftext|> + <|endoftext|>.
Parity vs the reference's own tokenizer output is unpinned (no vocab file, no fixtures on disk);
tests exercise the mechanics on a small synthetic merges file.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Dict, List, Tuple

import regex as re


def default_bpe() -> str:
    return os.environ.get("CCLIP_BPE_PATH",
                          os.path.join(os.path.dirname(os.path.abspath(__file__)), "bpe_simple_vocab_16e6.txt.gz"))


@lru_cache()
def bytes_to_unicode() -> Dict[int, str]:
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(2 ** 8):
        if b not in bs:
            bs.append(b)
            cs.append(2 ** 8 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def get_pairs(word: Tuple[str, ...]):
    return set(zip(word[:-1], word[1:]))


def basic_clean(text: str) -> str:
    # openai/CLIP runs ftfy.fix_text first; ftfy is not installed here, so only the html unescape is applied
    try:
        import ftfy  # type: ignore
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    return html.unescape(html.unescape(text)).strip()


def whitespace_clean(text: str) -> str:
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None, max_merges: int = 49152 - 256 - 2):
        bpe_path = bpe_path or default_bpe()
        if not os.path.exists(bpe_path):
            raise FileNotFoundError(
                f"BPE merges file not found at {bpe_path}. It ships with openai/CLIP "
                "(clip/bpe_simple_vocab_16e6.txt.gz); copy it there or set CCLIP_BPE_PATH.")
        self.byte_encoder = bytes_to_unicode()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        opener = gzip.open if bpe_path.endswith(".gz") else open
        with opener(bpe_path, "rb") as f:
            lines = f.read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:max_merges + 1] if len(m.split()) == 2]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab]
        vocab.extend("".join(m) for m in merges)
        vocab.extend(["<|startoftext|>", "<|endoftext|>"])
        self.encoder = dict(zip(vocab, range(len(vocab))))
        self.decoder = {v: k for k, v in self.encoder.items()}
        self.bpe_ranks = dict(zip(merges, range(len(merges))))
        self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
        self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                              re.IGNORECASE)

    def bpe(self, token: str) -> str:
        if token in self.cache:
            return self.cache[token]
        word = tuple(token[:-1]) + (token[-1] + "</w>",)
        pairs = get_pairs(word)
        if not pairs:
            return token + "</w>"
        while True:
            bigram = min(pairs, key=lambda pair: self.bpe_ranks.get(pair, float("inf")))
            if bigram not in self.bpe_ranks:
                break
            first, second = bigram
            new_word: List[str] = []
            i = 0
            while i < len(word):
                try:
                    j = word.index(first, i)
                except ValueError:
                    new_word.extend(word[i:])
                    break
                new_word.extend(word[i:j])
                i = j
                if word[i] == first and i < len(word) - 1 and word[i + 1] == second:
                    new_word.append(first + second)
                    i += 2
                else:
                    new_word.append(word[i])
                    i += 1
            word = tuple(new_word)
            if len(word) == 1:
                break
            pairs = get_pairs(word)
        out = " ".join(word)
        self.cache[token] = out
        return out

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        text = whitespace_clean(basic_clean(text)).lower()
        for token in re.findall(self.pat, text):
            token = "".join(self.byte_encoder[b] for b in token.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self.bpe(token).split(" "))
        return ids

    def decode(self, tokens) -> str:
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")
