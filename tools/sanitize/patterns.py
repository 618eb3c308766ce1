"""Mutated split patterns through the ASan/UBSan build of rx_compile + the matcher.  usage: python tools/sanitize/patterns.py <seed> <n> <harness>"""
import sys, random, subprocess
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tokendagger_amd import vocab_io
import test_generic_pattern as T
seed = int(sys.argv[1]); n = int(sys.argv[2])
rng = random.Random(seed)
seeds = list(T.PATTERNS.values()) + [vocab_io.LLAMA4_PAT_STR, vocab_io.TEKKEN_PAT_STR] + T.REJECTED
META = list("()[]{}|?*+\\^$.-:,!=<>") + ["\\p{", "\\P{", "(?:", "(?i:", "(?=", "(?!", "(?<=", "(?<!", "[[:", ":]]", "\\x{", "{1,", "?+", "*+", "++", "??", "\\h", "\\N", "\\v", "é", "中", "\x00", "\xff"]
def mutate(p):
    b = p
    for _ in range(rng.randint(1, 5)):
        r = rng.random()
        pos = rng.randrange(len(b) + 1)
        if r < 0.4: b = b[:pos] + rng.choice(META) + b[pos:]
        elif r < 0.6 and b: q = rng.randrange(len(b)); b = b[:q] + b[q + rng.randint(1, 4):]
        elif r < 0.8: b = b[:pos] + rng.choice(seeds)[:rng.randint(1, 12)] + b[pos:]
        else: b = b[:pos]
    return b.replace("\n", "\\n")
lines = []
for i in range(n):
    lines.append(mutate(rng.choice(seeds)) if rng.random() < 0.9 else T._random_pattern(rng))
open('/tmp/rx_asan_in.txt', 'w', encoding='utf-8', errors='surrogateescape').write("\n".join(lines) + "\n")
p = subprocess.run([sys.argv[3], "/tmp/rx_asan_in.txt"], capture_output=True, text=True, timeout=600)
print("rc", p.returncode, p.stdout.strip()); 
if p.returncode != 0: print(p.stderr[-2500:])
