"""CPU model of the merge rounds of lp_merge_lds<64> (csrc/td_kernels.hip; round 6): EVERY pair of the lowest rank at once, as far as that is
what the reference's sequential loop does (/root/reference/src/tiktoken/tiktoken.cpp:322-343: lowest rank first, leftmost on ties).

A round: r = the lowest rank present.  The pairs of rank r are taken greedily from the left (in a run of consecutive ones every second one:
the others lose a part to their left neighbour's merge).  The sequential loop merges exactly these, in this order, AS LONG AS no pair that
the merges create — the pair (part in front, merged part), where the part in front is itself a merged part when the pair before was taken
two positions to the left, and the pair (merged part, the still unmerged part behind it) — ranks at or below r: such a pair lies to the left
of every rank-r pair that is still to come, so the loop would take it first (strictly lower rank, or the leftmost of equal ranks).  The
round therefore applies the selected merges up to AND INCLUDING the first one that creates such a pair, and stops there; the next round
starts from the lowest rank again.  At least one merge per round (the plain sequential step), hundreds on repetitive pieces
('a' * 1000: 10 rounds instead of 999).  Checked here against the heap form of the reference's loop (oracle/td_oracle.c) before it went
into the kernel; tests/test_rank_batches_model.py.

    python tools/sim_rank_batches.py
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


class RankBatchMerger:
    def __init__(self, mergeable_ranks: dict):
        self.mr = mergeable_ranks
        self.tok = {r: b for b, r in mergeable_ranks.items()}

    def rank_pair(self, a: int, b: int) -> int:
        r = self.mr.get(self.tok[a] + self.tok[b])
        return INF if r is None else r

    def merge(self, piece: bytes):
        """-> (ids, rounds, merges)"""
        rp = self.rank_pair
        ids = [self.mr[bytes([c])] for c in piece]
        rk = [rp(ids[i], ids[i + 1]) for i in range(len(ids) - 1)]
        rounds = merges = 0
        while rk:
            r = min(rk)
            if r >= INF:
                break
            rounds += 1
            m = len(ids)
            sel, q = [], 0
            while q < m - 1:
                if rk[q] == r:
                    sel.append(q)
                    q += 2
                else:
                    q += 1
            # how many of them the sequential loop merges before something else comes first
            napply = len(sel)
            newL, newR = {}, {}
            for k, w in enumerate(sel):
                prev = r if (k > 0 and sel[k - 1] == w - 2) else (ids[w - 1] if w > 0 else None)
                L = rp(prev, r) if prev is not None else INF
                R = rp(r, ids[w + 2]) if w + 2 < m else INF
                newL[w], newR[w] = L, R
                if k + 1 < len(sel) and (L <= r or R <= r):
                    napply = k + 1
                    break
            app = set(sel[:napply])
            new_ids, new_rk = [], []
            q = 0
            while q < m:
                if q in app:
                    new_ids.append(r)
                    # the pair that starts at the merged part: with the next merged part (that one's L) or with the unmerged part behind it
                    nxt = q + 2
                    if nxt < m:
                        new_rk.append(newL[nxt] if nxt in app else newR[q])
                    q += 2
                else:
                    new_ids.append(ids[q])
                    if q + 1 < m:
                        new_rk.append(newL[q + 1] if (q + 1) in app else rk[q])
                    q += 1
            merges += napply
            ids, rk = new_ids, new_rk[:len(new_ids) - 1]
        return ids, rounds, merges


def main():
    import helpers as H
    from oracle import port
    _, mr, _ = H.llama4()
    bm = RankBatchMerger(mr)
    O = port.OracleTokenizer(mr)
    port.set_heap_threshold(0)
    rng = random.Random(5)
    cases = {"a * 1000": b"a" * 1000, "abc * 300": b"abc" * 300, "dashes * 500": b"-" * 500, "blanks * 100": b" " * 100,
             "random letters 1000": bytes(rng.choice(b"abcdefghijklmnopqrstuvwxyz") for _ in range(1000)),
             "DNA 1000": bytes(rng.choice(b"ACGT") for _ in range(1000)),
             "runs of a, b, c": b"".join(bytes([rng.choice(b"abc")]) * rng.randrange(1, 9) for _ in range(200))[:1000]}
    for name, piece in cases.items():
        t0 = time.time()
        ids, rounds, merges = bm.merge(piece)
        want = O.encode_ordinary(piece) if piece.strip() else None
        ok = want is None or (len(want) == len(ids) and bool((np.asarray(ids) == want).all()))
        print(f"{name}: {len(piece)} bytes -> {len(ids)} ids; {rounds} rounds for {merges} merges, {time.time() - t0:.2f} s, {'EXACT' if ok else 'MISMATCH'}")


if __name__ == "__main__":
    main()
