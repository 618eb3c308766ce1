"""Desk fuzz (CPU): the CPU twin of the device algorithm (tests/twin: tiled speculative pre-tokenizer, exact-key lookup, lane merge)
against the compiled reference on batches of nasty documents — mixed scripts, white-space runs of every kind, contractions,
digits, emoji sequences, long runs — for one of the pattern families.
    python tools/fuzz_twin_vs_reference.py <llama4|tekken|cl100k|gpt2> <seed> <iterations>
Round 3: 360 000 documents over the four families, no difference."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
which = sys.argv[1]; seed = int(sys.argv[2]); iters = int(sys.argv[3])
tw = getattr(H, "twin_" + which)()
R = {"llama4": H.ref_tokenizer, "tekken": H.ref_tokenizer_tekken, "cl100k": H.ref_tokenizer_cl100k, "gpt2": H.ref_tokenizer_gpt2}[which]()
rng = random.Random(seed)
ALPH = list(" \t\n\r\x0b\x0c  　 '’\"`.,;:!?-_=+*/\\|(){}[]<>@#$%^&~0123456789aAbBcdeEfgstTlLvVrRmMdD")
ALPH += list("éÉßñüÜøŒǅʰΩωжЖאبहि中文字あアー가각😀👍🏽́‍­½Ⅷ٣३")
WORDS = ["the", " The", "'s", "'T", "'ll", "'LL", "n't", "'Re", "don't", "I'M", "  ", "\n\n", " \n", "\r\n", "123", "1234567", "3.14", "x=1", "__init__", "camelCase", "snake_case", "http://a.b/c?d=e", "中文字符", "русский", "ελληνικά", "日本語のテキスト", "한국어", "😀😀", "a"*70, " "*40, "\n"*9, "\t\t\tcode();"]
t0 = time.time(); nd = 0
for it in range(iters):
    docs = []
    for _ in range(rng.randint(1, 60)):
        r = rng.random()
        if r < 0.03: docs.append(b"")
        elif r < 0.5:
            docs.append("".join(rng.choice(ALPH) for _ in range(rng.randint(1, 300))).encode("utf-8"))
        elif r < 0.9:
            docs.append("".join(rng.choice(WORDS) if rng.random() < 0.6 else rng.choice(ALPH) for _ in range(rng.randint(1, 400))).encode("utf-8"))
        else:
            docs.append((rng.choice(WORDS) * rng.randint(1, 300)).encode("utf-8")[:9000])
    text, offs = H.pack_docs(docs)
    toks, toffs = tw.encode_batch(text, offs)
    _, etoks, eoffs = R.encode_batch(np.frombuffer(text, dtype=np.uint8), offs, n_threads=4, want_tokens=True)
    nd += len(docs)
    if not (np.array_equal(toffs, eoffs) and np.array_equal(toks, etoks)):
        # find the first differing document
        for d in range(len(docs)):
            a = toks[toffs[d]:toffs[d+1]]; b = etoks[eoffs[d]:eoffs[d+1]]
            if not np.array_equal(a, b):
                print("MISMATCH", which, "iter", it, "doc", d, repr(docs[d][:200])); break
        break
print(which, "seed", seed, ":", nd, "documents,", round(time.time()-t0,1), "s, ok" )
