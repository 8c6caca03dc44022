#!/usr/bin/env python3
"""Generates tests/golden/tokenizer_ids.json by RUNNING the reference's BPE tokenizer (clip/simple_tokenizer.py:62-132,
with its own merges file bpe_simple_vocab_16e6.txt.gz) in this container on a fixed list of texts.  `ftfy` is absent
here; the stand-in is the identity, which is what ftfy.fix_text does on these inputs (no mojibake in them).  Nothing of
the reference travels: the fixture holds the input strings and the ids they encode to (SOT / EOT / padding are added by
clip.tokenize and are not part of it)."""
import importlib.util
import json
import os
import sys
from pathlib import Path

sys.dont_write_bytecode = True
REPO = Path(__file__).resolve().parent.parent
REF = Path(os.environ.get("HGR_REFERENCE", "/root/reference"))
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tools"))

from make_golden import install_stubs

TEXTS = ["a photo of a great white shark.", "a photo of a kind12 thing3.", "Hello,   World!! it's 1999 &amp; co.",
         "a photo of a zebra-finch's nest (árbol).", "naïve café \U0001F600 emoji test", "a photo of a jack o' lantern.",
         "THE QUICK brown_fox jumps"]


def main():
    install_stubs()
    spec = importlib.util.spec_from_file_location("ref_simple_tokenizer", REF / "clip" / "simple_tokenizer.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tok = mod.SimpleTokenizer()
    out = {"texts": TEXTS, "ids": [tok.encode(t) for t in TEXTS]}
    path = REPO / "tests" / "golden" / "tokenizer_ids.json"
    old = json.load(open(path)) if path.exists() else None
    json.dump(out, open(path, "w"))
    print("written", path, "| identical to the committed fixture:", old == out)


if __name__ == "__main__":
    main()
