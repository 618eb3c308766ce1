"""N>1 path on CPU: world_size-2 gloo processes exercise document sharding and the count/offset gather
(the only collective of the tokenizer path; RCCL on the GPU box, gloo here)."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    from tokendagger_amd import dist as tdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(0)  # same corpus description on every rank
    lens = rng.integers(0, 5000, size=1000)
    offs = np.zeros(len(lens) + 1, dtype=np.int64)
    np.cumsum(lens, out=offs[1:])
    d0, d1 = tdist.shard_documents(offs, world, rank)
    # stand-in for the local GPU encode: tokens per document = ceil(bytes / 4)
    tok_per_doc = (lens + 3) // 4
    n_tok_local = int(tok_per_doc[d0:d1].sum())
    doc_base, tok_base, tot_docs, tot_tok, table = tdist.gather_counts(d1 - d0, torch.tensor([n_tok_local]))
    q.put((rank, d0, d1, doc_base, tok_base, tot_docs, tot_tok, int(offs[d1] - offs[d0]), int(tok_per_doc[:d0].sum())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_shard_and_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # contiguous cover of all documents, byte-balanced, consistent bases
    assert res[0][1] == 0 and res[-1][2] == 1000
    for a, b in zip(res, res[1:]):
        assert a[2] == b[1]
    total_bytes = sum(r[7] for r in res)
    for r in res:
        assert abs(r[7] - total_bytes / world) < 6000, "shards are balanced by bytes to within one document"
        assert r[3] == r[1], "document base == first document index"
        assert r[4] == r[8], "token base == tokens of all earlier documents"
        assert r[5] == 1000


def test_shard_documents_edge_cases():
    from tokendagger_amd.dist import shard_documents
    offs = np.asarray([0, 10, 10, 10, 50], dtype=np.int64)
    parts = [shard_documents(offs, 8, r) for r in range(8)]
    assert parts[0][0] == 0 and parts[-1][1] == 4
    assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
    assert shard_documents(np.asarray([0], dtype=np.int64), 2, 1) == (0, 0)


def test_bases_from_a_gathered_table_c_entry():
    """td_comm_bases (the host half of the C ABI's multi-GPU epilogue): exclusive prefix sums over {tokens, documents}."""
    from tokendagger_amd import capi
    table = np.asarray([[10, 2], [0, 0], [7, 5], [3, 1]], dtype=np.int64)
    assert [capi.comm_bases(table, r) for r in range(4)] == [(0, 0, 20, 8), (10, 2, 20, 8), (10, 2, 20, 8), (17, 7, 20, 8)]
    assert capi.comm_bases(np.asarray([5, 1], dtype=np.int64), 0) == (0, 0, 5, 1)
    with pytest.raises(capi.TokenDaggerHipError):
        capi.comm_bases(np.asarray([5, -1], dtype=np.int64), 0)
    with pytest.raises(capi.TokenDaggerHipError):
        capi.comm_bases(table, 4)
