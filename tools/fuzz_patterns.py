"""Desk fuzz (CPU): random patterns of the generic compiler's grammar (td_regex.cpp) against PCRE2's INTERPRETER behind the
reference's split loop (oracle/pcre2_interp.c) — not the compiled reference, whose PCRE2 10.39 JIT has bugs of its own on such
patterns (oracle/pcre2_probe.c).  tests/test_generic_pattern.py runs the same generator with fixed seeds.
    python tools/fuzz_patterns.py <seed> <patterns>"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from oracle import ref


def interp(pat, b):
    return ref.interp_split(pat, b)


rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ATOMS = [r"\s", r"\S", r"\d", r"\w", r"\W", r"\p{L}", r"\p{N}", r"\p{Lu}", r"\p{Ll}", r"[a-z]", r"[A-Z0-9_]", r"[^\s\p{L}\p{N}]", r"[^a-c\n]", r".", r"\h", r"[[:alpha:]]",
         "a", "b", " ", r"\n", "x", "é", "中", r"\.", "-", r"\v", r"\V", r"\H", r"\N", r"\p{Han}", r"[\p{Latin}0-9]", r"[[:upper:][:digit:]]", r"[^[:space:]]", r"\P{L}", r"\x41", r"\x{e9}", r"[\x{4e00}-\x{9fff}]", r"\D", r"\p{P}", r"\p{Zs}"]
QUANT = ["", "", "", "?", "*", "+", "{1,3}", "{2}", "{0,2}", "?+", "*+", "++", "??", "*?", "+?", "{1,3}?", "{2,}+"]
GROUPS = [r"(?:ab|a)", r"(?i:the|an|a)", r"(?:x|y|[01])", r"(?:'s|'t)", r"(?:a|b|[xy])", r"(?i:k|s)", r"(?:é|中)"]
GQ = ["", "?", "?+", "??", "+", "*", "{1,2}", "++", "*?"]
ZW = [r"\A", r"\Z", r"(?<![0-9])", r"(?=\p{L})", r"(?=\s)", r"(?!\S)", r"(?=[0-9])", r"(?!a)", r"\b", r"\B", "^", "$", r"\z", r"(?<=a)", r"(?<!\s)", r"(?<=\p{L})"]
def alt():
    parts = []
    for _ in range(rng.randrange(1, 4)):
        r = rng.random()
        if r < 0.70: parts.append(rng.choice(ATOMS) + rng.choice(QUANT))
        elif r < 0.85: parts.append(rng.choice(GROUPS) + rng.choice(GQ))
        else: parts.append(rng.choice(ZW))
    return "".join(parts)
def strings(n):
    al = " \t\n\r_aAbBxXyY019.,'-éÉ中ſK  "
    out = []
    for i in range(n):
        out.append("".join(rng.choice(al) for _ in range(rng.randrange(0, 40))))
    out += ["", "a", "ab", "the an a", "x01y", "a  b", "aaa", "'s't", "ABC abc 123", "\n\n", " \t "]
    return out
bad = 0; tried = 0; accepted = 0
S = strings(120)
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
    pat = "|".join(alt() for _ in range(rng.randrange(1, 5)))
    tried += 1
    try:
        H.rx_split(pat, b"abc")
    except ValueError:
        continue
    try:
        interp(pat, b'a')
    except Exception as e:
        print("PCRE2 rejects what we accept:", repr(pat), repr(e)[:80]); bad += 1; continue
    accepted += 1
    for s in S:
        b = s.encode("utf-8")
        got = [b[a:e] for a, e in H.rx_split(pat, b)]
        want = interp(pat, b)
        if got != want:
            print("MISMATCH", repr(pat), repr(s), got[:6], want[:6]); bad += 1; break
print("patterns tried", tried, "accepted", accepted, "bad", bad)
