"""Planning aid (CPU): how many wavefront-rounds td_merge_pieces needs for the missed pieces of a corpus under different batching
schemes — the current five length classes with batches that run as long as their longest chain, sixteen classes, and lanes that
take the next piece as soon as enough of them are free (rolling batches).  Pieces and their merge counts come from the compiled
reference's split and the restatement's encode.  usage: python tools/sim_merge_batches.py"""
import sys, random
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import helpers as H, td_corpus
R = H.ref_tokenizer(); O = H.port_tokenizer()
def pieces_of(kind, mb):
    x, offs = getattr(td_corpus, kind)(mb << 20, seed=7)
    out = []
    data = x.tobytes()
    for d in range(len(offs) - 1):
        doc = data[offs[d]:offs[d+1]]
        for p in R.split_pieces(doc):
            n = len(p)
            if n < 2 or n > 64: continue
            ids = O.encode(p)
            if len(ids) == 1: continue          # whole-piece hit
            out.append((n, n - len(ids)))        # (bytes, merges = rounds)
    return out
def cls5(n): return 0 if n <= 8 else 1 if n <= 16 else 2 if n <= 32 else 3 if n <= 48 else 4
units = [1, 1, 2, 3, 4]
def sim_batches(P, classify, units_of):
    # arrival order, per-class FIFO batches of 64/u: wave-rounds = sum of max chain per batch
    q = {}
    tot = 0; lane_rounds = 0; useful = 0
    for n, r in P:
        c = classify(n); q.setdefault(c, []).append(r)
    for c, L in q.items():
        per = 64 // units_of(c)
        for i in range(0, len(L), per):
            b = L[i:i+per]; tot += max(b); useful += sum(b)
    return tot, useful
def sim_rolling(P, classify, units_of, refill_min):
    # per class: lanes take the next piece when at least refill_min lanes are free (or nothing else runs)
    q = {}
    for n, r in P: q.setdefault(classify(n), []).append(r)
    tot = 0
    for c, L in q.items():
        per = 64 // units_of(c)
        lanes = [0] * per; i = 0
        while True:
            free = [k for k in range(per) if lanes[k] == 0]
            if i < len(L) and (len(free) >= refill_min or len(free) == per):
                for k in free:
                    if i < len(L): lanes[k] = L[i]; i += 1
                tot += 0.5   # a refill step costs about half a round (set-up of the refilled lanes, the others wait)
            if all(v == 0 for v in lanes):
                if i >= len(L): break
                continue
            lanes = [v - 1 if v > 0 else 0 for v in lanes]; tot += 1
    return tot
for kind, mb in (("mixed", 4), ("code", 4)):
    P = pieces_of(kind, mb)
    print(kind, len(P), "missed pieces, avg rounds", round(sum(r for _, r in P) / len(P), 2))
    t5, useful = sim_batches(P, cls5, lambda c: units[c])
    print("  5 classes, batches     : wave-rounds", t5, " lane utilisation", round(useful / (t5 * 64), 3))
    fine = lambda n: min((n - 1) // 4, 15)
    funits = lambda c: 1 if c < 4 else 2 if c < 8 else 3 if c < 12 else 4
    t16, _ = sim_batches(P, fine, funits)
    print("  16 classes (4-byte steps): wave-rounds", t16)
    for rm in (8, 16, 32):
        print("  rolling, refill at", rm, "free lanes:", round(sim_rolling(P, cls5, lambda c: units[c], rm)))
