"""The reference's code benchmark, its own way: ONE encode call per file of its file set, through both modules
(/root/reference/tests/code_performance_benchmark.py:338-396 times Tokenizer.encode(content) per file and reports tokens/s per file; the file
set it selects here — 21 files, 2 146 667 bytes — is tests/golden/code_corpus.npz, made by tools/make_code_corpus.py).  This package:
enc.encode(text) (MI355X; host str in, Python list out); the reference: CoreBPE.encode(text, set()) of its unmodified module on one CPU
core (oracle/ref_latency.py, a process of its own).  -> a table for profiles/ (VERDICT r5 item 5c)."""
import json, os, subprocess, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
import tokendagger as tiktoken

z = np.load(os.path.join(ROOT, "tests", "golden", "code_corpus.npz"), allow_pickle=True)
text, offs, names = z["text"].tobytes(), z["offsets"], [str(x) for x in z["names"]]
T = [(names[i], text[int(offs[i]):int(offs[i + 1])].decode("utf-8")) for i in range(len(names))]
pat, mr, sp = H.llama4()
enc = tiktoken.Encoding("llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=sp)
mine, mine_np = {}, {}
for name, t in T:
    n = 200 if len(t) < 2000 else 30 if len(t) < 100000 else 8
    for _ in range(3):
        enc.encode(t)
    t0 = time.perf_counter()
    for _ in range(n):
        ids = enc.encode(t)
    mine[name] = ((time.perf_counter() - t0) / n * 1e6, len(ids))
    if hasattr(enc, "encode_to_numpy"):
        for _ in range(2):
            enc.encode_to_numpy(t)
        t0 = time.perf_counter()
        for _ in range(n):
            a = enc.encode_to_numpy(t)
        mine_np[name] = (time.perf_counter() - t0) / n * 1e6
ref = {}
if os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "refmod")):
    with tempfile.NamedTemporaryFile("w", suffix=".json", dir="/tmp", delete=False) as f:
        json.dump(T, f)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_latency.py"), f.name], capture_output=True, text=True, timeout=1800)
    if r.returncode == 0:
        ref = json.loads(r.stdout.strip().splitlines()[-1])
    else:
        print("reference module failed:", r.stderr[-300:])
print("# one encode call per file of the reference's code_performance_benchmark file set (21 files): enc.encode(text) -> list[int], microseconds per call;")
print(f"# this package (MI355X) | its encode_to_numpy | the reference's own module (CPU, one of {os.cpu_count()} hardware threads) | ids (must agree)")
print(f"{'file':<40} {'bytes':>8} {'this, us':>10} {'numpy, us':>10} {'reference, us':>14} {'ratio':>7} {'ids':>8}")
tot_m = tot_r = 0.0
for name, t in T:
    us, k = mine[name]
    rv = ref.get(name)
    same = "" if not rv or rv[1] == k else f"  IDS DIFFER ({rv[1]})"
    tot_m += us; tot_r += rv[0] if rv else 0.0
    print(f"{name:<40} {len(t.encode()):>8} {us:>10.1f} {(f'{mine_np[name]:.1f}' if name in mine_np else '-'):>10} {(f'{rv[0]:.1f}' if rv else '-'):>14} {(f'{rv[0] / us:.2f}x' if rv else '-'):>7} {k:>8}{same}")
print(f"{'all 21 files, one after the other':<40} {len(text):>8} {tot_m:>10.1f} {(f'{sum(mine_np.values()):.1f}' if mine_np else '-'):>10} {tot_r:>14.1f} {(f'{tot_r / tot_m:.2f}x' if tot_r else '-'):>7}")
