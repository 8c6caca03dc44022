"""Byte-level BPE tokenizer for CLIP prompts (the algorithm of clip/simple_tokenizer.py in the
reference: GPT-2 style byte->unicode alphabet, ranked merges, '</w>' word-end marker, lower-cased
whitespace-normalised text; 49 408 ids with <|startoftext|> = 49406, <|endoftext|> = 49407).

Host-side, one-off per model construction; the kernels only ever see token ids.  The merges file
(`bpe_simple_vocab_16e6.txt.gz`, 1.3 MB) is data of the reference checkout and is NOT shipped here:
point ``HGR_BPE_VOCAB`` at it (or pass ``bpe_path``).
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache

import regex as re

_PAT = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", re.IGNORECASE)


def default_bpe_path() -> str:
    for cand in (os.environ.get("HGR_BPE_VOCAB"), "clip/bpe_simple_vocab_16e6.txt.gz"):      # env, or a reference checkout as cwd
        if cand and os.path.isfile(cand):
            return cand
    raise FileNotFoundError("BPE merges file not found: set HGR_BPE_VOCAB to the reference's clip/bpe_simple_vocab_16e6.txt.gz")


@lru_cache()
def byte_alphabet():
    """Reversible byte -> printable unicode map: printable latin-1 bytes map to themselves, the rest to 256+."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    chars = keep[:]
    extra = 0
    for b in range(256):
        if b not in keep:
            keep.append(b)
            chars.append(256 + extra)
            extra += 1
    return dict(zip(keep, (chr(c) for c in chars)))


def _clean(text: str) -> str:
    try:
        import ftfy
        text = ftfy.fix_text(text)
    except ImportError:
        pass
    text = html.unescape(html.unescape(text)).strip()
    return re.sub(r"\s+", " ", text).strip()


class SimpleTokenizer:
    def __init__(self, bpe_path: str = None):
        self.byte_encoder = byte_alphabet()
        self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
        lines = gzip.open(bpe_path or default_bpe_path()).read().decode("utf-8").split("\n")
        merges = [tuple(m.split()) for m in lines[1:49152 - 256 - 2 + 1]]
        vocab = list(self.byte_encoder.values())
        vocab = vocab + [v + "</w>" for v in vocab] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self._cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}

    def _bpe(self, token: str) -> str:
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        word = list(token[:-1]) + [token[-1] + "</w>"]
        while len(word) > 1:
            ranked = [(self.rank.get((a, b), 1 << 60), i) for i, (a, b) in enumerate(zip(word, word[1:]))]
            best, _ = min(ranked)
            if best == 1 << 60:
                break
            first, second = next((a, b) for (a, b) in zip(word, word[1:]) if self.rank.get((a, b)) == best)
            merged, i = [], 0
            while i < len(word):                      # merge every occurrence of the best pair, left to right
                if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
                    merged.append(first + second)
                    i += 2
                else:
                    merged.append(word[i])
                    i += 1
            word = merged
        out = " ".join(word)
        self._cache[token] = out
        return out

    def encode(self, text: str):
        ids = []
        for tok in re.findall(_PAT, _clean(text).lower()):
            tok = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            ids.extend(self.encoder[t] for t in self._bpe(tok).split(" "))
        return ids

    def decode(self, tokens) -> str:
        text = "".join(self.decoder[t] for t in tokens)
        return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")
