"""td_clone (include/tokendagger_hip.h): a second handle on the SAME device tables, with a workspace of its own — what a host
thread per HIP stream needs ("one lock + one workspace per handle; concurrency costs a 30 MB table copy per
handle").  The reference shares one CoreBPE between the threads of its pool (tokendagger/wrapper.py:212-235)."""
import threading

import numpy as np
import pytest

import helpers as H
from oracle import ref

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not ref.available(), reason="compiled reference (oracle/_ref) not built")
@pytest.mark.parametrize("graphs", [0, 1], ids=["plain-launches", "hipgraph-replay"])
def test_clones_encode_concurrently_and_outlive_the_original(graphs):
    """Four host threads, a clone and a stream each, 50 identical calls in a row (with TD_OPT_GRAPH on: one plain step, one
    capture, 48 replays per thread) while a fifth thread keeps torch's default (legacy) stream busy with copies — the library
    must neither use the legacy stream itself nor put an application stream into capture (the capture on the
    caller's stream + hipMemcpy on the null stream broke exactly this)."""
    import torch
    import td_corpus
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    R = ref.RefTokenizer(pat, mr, special)
    first = capi.HipTokenizer(pat, mr, special, device=0)
    first.set_option(capi.TD_OPT_GRAPH, graphs)  # (clones inherit the options set so far)
    mem0 = torch.cuda.mem_get_info(0)[0]
    toks = [first] + [first.clone() for _ in range(3)]
    assert mem0 - torch.cuda.mem_get_info(0)[0] < (8 << 20), "a clone has no copy of the ~30 MB of tables"
    corpora = [td_corpus.english(6 << 20, seed=31), td_corpus.mixed(4 << 20, seed=32), td_corpus.code(5 << 20, seed=33), td_corpus.english(3 << 20, seed=34)]
    want = []
    for x, offs in corpora:
        _, et, eo = R.encode_batch(x, offs, n_threads=8, want_tokens=True)
        want.append((et, eo))
    got = [None] * 4
    errors = []
    stop = threading.Event()
    legacy_rounds = [0]

    def legacy_stream_traffic():  # what any other part of an application does meanwhile: torch's default stream is the null stream
        try:
            torch.cuda.set_device(0)
            while not stop.is_set():
                v = torch.zeros(1).cuda().cpu()
                assert float(v[0]) == 0.0
                legacy_rounds[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(("legacy", repr(e)))

    def work(k):
        try:
            torch.cuda.set_device(0)
            s = torch.cuda.Stream()
            x, offs = corpora[k]
            dt = torch.from_numpy(x).cuda()
            do = torch.from_numpy(offs).cuda()
            dk = torch.empty(len(x) + 1024, dtype=torch.int32, device="cuda")
            dto = torch.empty(len(offs), dtype=torch.int64, device="cuda")
            torch.cuda.synchronize()
            for it in range(50):  # (repeated: the calls of the four handles overlap on the device)
                toks[k].encode_device(dt.data_ptr(), len(x), do.data_ptr(), len(offs) - 1, dk.data_ptr(), dk.numel(), dto.data_ptr(), s.cuda_stream)
                if it % 10 == 9:
                    toks[k].device_status(s.cuda_stream)
            s.synchronize()
            toks[k].device_status(s.cuda_stream)
            eo = dto.cpu().numpy()
            got[k] = (dk[:eo[-1]].cpu().numpy(), eo)
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((k, repr(e)))

    fifth = threading.Thread(target=legacy_stream_traffic)
    fifth.start()
    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    stop.set()
    fifth.join()
    assert not errors, errors
    assert legacy_rounds[0] > 0
    for k in range(4):
        assert np.array_equal(got[k][1], want[k][1]), k
        assert np.array_equal(got[k][0], want[k][0]), k
    # the tables live as long as one handle does: close the original first, then use a clone
    first.close()
    x, offs = corpora[1]
    t2, o2 = toks[2].encode_batch(x.tobytes(), offs)
    assert np.array_equal(o2, want[1][1]) and np.array_equal(t2, want[1][0])
    for t in toks[1:]:
        t.close()
