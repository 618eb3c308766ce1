"""Multi-GPU plumbing for the tokenizer path: documents shard trivially, so the only exchange is the
epilogue gather of per-rank {documents, tokens} -> global offsets (RCCL over xGMI on MI355X; the same
code runs on gloo for CPU tests).  One process per GPU, launched by torch.distributed.run."""
from __future__ import annotations

import numpy as np


def shard_documents(doc_offsets: np.ndarray, world: int, rank: int) -> tuple[int, int]:
    """Contiguous document range [d0, d1) of `rank`, balanced by BYTES (not by document count).
    Every document lands on exactly one rank; ranks may be empty when there are fewer documents than ranks."""
    offs = np.asarray(doc_offsets, dtype=np.int64)
    n_docs = len(offs) - 1
    total = int(offs[-1] - offs[0])
    bounds = [0]
    for r in range(1, world):
        target = offs[0] + (total * r) // world
        d = int(np.searchsorted(offs, target, side="left"))
        bounds.append(min(max(d, bounds[-1]), n_docs))
    bounds.append(n_docs)
    return bounds[rank], bounds[rank + 1]


def gather_counts(n_docs_local: int, n_tokens_local, device=None, group=None):
    """all-gather of {tokens, documents} per rank (the layout of the C ABI's td_comm_gather_counts).
    -> (doc_base, token_base, total_docs, total_tokens, table) where *_base are this rank's exclusive prefix sums, computed
    by the C entry td_comm_bases.  `n_tokens_local` may be an int or a 0-d/1-element tensor that already lives on `device`
    (no host sync needed before the collective).  The collective itself is torch.distributed's here (RCCL with the "nccl"
    backend, gloo on CPU); tokendagger_amd.capi.RcclComm is the same exchange without torch."""
    import torch
    import torch.distributed as dist
    from . import capi
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    mine = torch.zeros(2, dtype=torch.int64, device=device)
    if torch.is_tensor(n_tokens_local):
        mine[0:1] = n_tokens_local.reshape(-1)[:1].to(torch.int64)
    else:
        mine[0] = int(n_tokens_local)
    mine[1] = n_docs_local
    table = torch.zeros(2 * world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(table, mine, group=group)
    t = table.view(world, 2).cpu().numpy()
    tok_base, doc_base, tok_total, doc_total = capi.comm_bases(t, rank)
    return doc_base, tok_base, doc_total, tok_total, t
