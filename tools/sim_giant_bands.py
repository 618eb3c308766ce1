"""CPU model of td_giant_pieces' rounds (csrc/td_kernels.hip): bands of ranks merged per round instead of one rank.

A round: candidates = pairs ranking strictly below their left neighbour pair and not above their right one (runs of equal
ranks that start so: every second pair); the candidates below a bound B merge together; B is lowered until every pair below
B that is left over is overlapped by a merging pair AND every pair the merges create (and every transient pair between two
merges one part apart) ranks at or above B; a bound at the lowest rank present = the plain sequential step.  Used to check
the rule against the heap form of the reference's merge loop (oracle/td_oracle.c; /root/reference/src/tiktoken/tiktoken.cpp:298-368)
before it went into the kernel, and by tests/test_giant_bands_model.py.

    python tools/sim_giant_bands.py            # a few large pieces: rounds, lookups, exactness
"""
from __future__ import annotations

import random
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

INF = 1 << 40


class BandMerger:
    def __init__(self, mergeable_ranks: dict):
        self.mr = mergeable_ranks
        self.tok = {r: b for b, r in mergeable_ranks.items()}

    def rank_pair(self, a: int, b: int) -> int:
        r = self.mr.get(self.tok[a] + self.tok[b])
        return INF if r is None else r

    def merge(self, piece: bytes):
        """-> (ids, rounds, bound iterations, sequential steps, pair lookups)"""
        rp = self.rank_pair
        ids = [self.mr[bytes([c])] for c in piece]
        rk = [rp(ids[i], ids[i + 1]) for i in range(len(ids) - 1)] + [INF]
        rounds = inner = singles = lookups = 0
        while True:
            m = len(ids)
            g = min(rk) if m else INF
            if g >= INF:
                break
            rounds += 1
            rka = np.asarray(rk, dtype=np.int64)
            left = np.concatenate([[INF], rka[:-1]])
            right = np.concatenate([rka[1:], [INF]])
            idx = np.arange(m)
            run_start = np.maximum.accumulate(np.where(rka != left, idx, 0))
            base = (rka < INF) & (rka <= right) & (rka < left)[run_start] & (((idx - run_start) & 1) == 0)
            B = INF
            while True:
                inner += 1
                sel = base & (rka < B)
                sel_l = np.concatenate([[False], sel[:-1]])
                sel_r = np.concatenate([sel[1:], [False]])
                surv = ~sel & ~sel_l & ~sel_r
                vmin = int(rka[surv].min()) if surv.any() else INF
                for i in map(int, np.flatnonzero(sel)):
                    a_id = int(rka[i])
                    if i + 2 < m:
                        nxt = int(rka[i + 2]) if sel[i + 2] else ids[i + 2]
                        vmin = min(vmin, rp(a_id, nxt)); lookups += 1
                        if sel[i + 2]:
                            t = rp(a_id, ids[i + 2]) if rka[i] <= rka[i + 2] else rp(ids[i + 1], int(rka[i + 2]))
                            vmin = min(vmin, t); lookups += 1
                    if i >= 1 and not (i >= 2 and sel[i - 2]):
                        vmin = min(vmin, rp(ids[i - 1], a_id)); lookups += 1
                smax = int(rka[sel].max()) if sel.any() else 0
                if smax < vmin:
                    break
                B = vmin
                if B <= g:
                    break
            if B <= g:  # the sequential step
                singles += 1
                sel = np.zeros(m, bool)
                sel[int(np.flatnonzero(rka == g)[0])] = True
            new_ids, changed, old_pos = [], [], []
            i = 0
            while i < m:
                old_pos.append(i)
                if sel[i]:
                    new_ids.append(int(rka[i])); changed.append(True); i += 2
                else:
                    new_ids.append(ids[i]); changed.append(False); i += 1
            new_rk = []
            for k in range(len(new_ids)):
                if k + 1 >= len(new_ids):
                    new_rk.append(INF)
                elif not changed[k] and not changed[k + 1]:
                    new_rk.append(rk[old_pos[k]])
                else:
                    new_rk.append(rp(new_ids[k], new_ids[k + 1])); lookups += 1
            ids, rk = new_ids, new_rk
        return ids, rounds, inner, singles, lookups


def main():
    import helpers as H
    from oracle import port
    _, mr, _ = H.llama4()
    bm = BandMerger(mr)
    O = port.OracleTokenizer(mr)
    port.set_heap_threshold(0)
    rng = random.Random(5)
    cases = {
        "random letters 20K": bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(20000)),
        "random letters 200K": bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(200000)),
        "a * 50K": b"a" * 50000,
        "ab * 20K": b"ab" * 20000,
        "DNA 100K": bytes(rng.choice(b"ACGT") for _ in range(100000)),
        "skewed letters 100K": bytes(rng.choice(b"eeeeeeetttttaaaaooooiiinnnssshhrrdlcumwfgypbvkjxqz") for _ in range(100000)),
        "runs of a, b, c 60K": b"".join(bytes([rng.choice(b"abc")]) * rng.randrange(1, 9) for _ in range(13000)),
    }
    for name, piece in cases.items():
        t0 = time.time()
        ids, rounds, inner, singles, lookups = bm.merge(piece)
        want = O.encode_ordinary(piece)  # (a run of lower-case or of upper-case letters is ONE piece of the Llama-4 pattern)
        ok = len(want) == len(ids) and bool((np.asarray(ids) == want).all())
        print(f"{name}: {len(piece)} bytes -> {len(ids)} ids; {rounds} rounds, {inner} bound iterations, {singles} sequential steps, {lookups} lookups, "
              f"{time.time() - t0:.1f} s, {'EXACT' if ok else 'MISMATCH'}", flush=True)


if __name__ == "__main__":
    main()
