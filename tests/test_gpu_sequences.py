"""The two launch sequences of a step (round 6, TD_OPT_SPARSE): SPARSE — td_prepare_mark, td_split_tiles, td_tail, td_giant_scan,
td_pack_plain, td_pack_rest — and DENSE — the phases of td_tail / td_giant_scan as kernels of their own (td_far_probe, td_collect_misses,
td_merge_pieces, td_copy_dups, td_long_pieces, td_giant_pieces, td_scan_tiles).  Either must give the reference's ids on ANY text
(CoreBPE::encode, /root/reference/src/tiktoken/tiktoken.cpp:169-234): the library picks one from the last counters it has seen, and a
wrong guess may only cost time.  Every text below goes through both, forced, with plain launches and with graph replay, and through the
library's own choice, against the compiled reference; td_tail's fallback for a launch whose workgroups are not resident together
(workgroup 0 walks the phases alone) is forced with a grid far larger than the device holds."""
from __future__ import annotations

import os

import numpy as np
import pytest

import helpers as H
import td_corpus
from oracle import ref
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


def _texts():
    """name -> (bytes, offsets): plain prose; prose with every kind of rare work in it; dense text."""
    out = {}
    x, o = td_corpus.english(3 << 20, seed=11)
    out["english"] = (x, o)
    # prose + far pieces (longer than the tile loop's window), long pieces (65..1024 bytes), giant pieces (> 1 KiB, > 16 KiB: all workgroups),
    # a tile with many missed pieces, a run of digits across tiles, a tile of more pieces than the loop's list holds
    extra = [
        ("x" * 300 + " ").encode(), ("ab" * 2000 + "\n").encode(), ("q" * 40000 + " tail\n").encode(), ("中文" * 700 + "\n").encode(),
        (" ".join("zq%dxj" % i for i in range(3000)) + "\n").encode(), ("7" * 20000 + "\n").encode(), ("a b " * 6000 + "\n").encode(),
        ("\n".join("https://example.org/" + "p%d/" % i * 12 + "index.html?q=%d" % i for i in range(400)) + "\n").encode(),
        ("é" * 5000 + " " + "ß" * 90 + "\n").encode(),
    ]
    xs = [x[: 1 << 20]]
    offs = list(o[o <= (1 << 20)])
    if offs[-1] != 1 << 20:
        offs.append(1 << 20)
    pos = 1 << 20
    for e in extra:
        xs.append(np.frombuffer(e, dtype=np.uint8))
        pos += len(e)
        offs.append(pos)
    tailx, tailo = td_corpus.english(1 << 20, seed=12)
    xs.append(tailx)
    offs.extend(list(pos + tailo[1:]))
    out["english_with_rare_work"] = (np.concatenate(xs), np.asarray(offs, dtype=np.int64))
    out["mixed"] = td_corpus.mixed(3 << 20, seed=13)
    out["code"] = td_corpus.code(2 << 20, seed=14) if hasattr(td_corpus, "code") else td_corpus.mixed(1 << 20, seed=15)
    return out


def _run(tok, torch, x, o, label, et, eo):
    n, nd = len(x), len(o) - 1
    s = torch.cuda.current_stream().cuda_stream
    dt, do = torch.from_numpy(np.ascontiguousarray(x)).cuda(), torch.from_numpy(o).cuda()
    dk = torch.zeros(n + 1024, dtype=torch.int32, device="cuda")
    dto = torch.zeros(nd + 1, dtype=torch.int64, device="cuda")
    tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), n + 1024, dto.data_ptr(), s)
    tok.device_status(s)
    toff = dto.cpu().numpy()
    assert np.array_equal(toff, eo), f"{label}: document offsets differ from the reference"
    got = dk[: int(toff[-1])].cpu().numpy()
    bad = np.flatnonzero(got != et)
    assert bad.size == 0, f"{label}: ids differ from the reference, first at token {int(bad[0])}"


def test_both_sequences_give_the_reference_ids_on_every_kind_of_text():
    import torch
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        for name, (x, o) in _texts().items():
            _, et, eo = H.ref_tokenizer().encode_batch(x, o, n_threads=os.cpu_count() or 1, want_tokens=True)
            for sparse in (1, 0, -1):
                tok.set_option(capi.TD_OPT_SPARSE, sparse)
                for graph in (0, 1):
                    tok.set_option(capi.TD_OPT_GRAPH, graph)
                    for rep in range(3 if graph else 2):
                        _run(tok, torch, x, o, f"{name}: sparse={sparse} graph={graph} call {rep}", et, eo)
                        if sparse >= 0:
                            assert tok.info(capi.TD_INFO_SPARSE) == sparse, "the forced sequence is the one that ran"
            # the library's own choice after it has seen this text's counters
            chosen = tok.info(capi.TD_INFO_SPARSE)
            if name == "english":
                assert chosen == 1, "plain prose takes the sparse sequence once its counters have been read"
            if name == "mixed":
                assert chosen == 0, "text with a flagged tile in every tile takes the dense sequence"
    finally:
        tok.set_option(capi.TD_OPT_GRAPH, 0)
        tok.close()


def test_a_fresh_handle_starts_dense_and_a_wrong_guess_only_costs_time():
    """The first call of a handle (no counters seen) is dense; after plain prose the handle is sparse, and the NEXT call — dense text, far
    pieces, giant pieces: everything td_tail has phases for — still runs sparse and must be right."""
    import torch
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        texts = _texts()
        refs = {k: H.ref_tokenizer().encode_batch(v[0], v[1], n_threads=os.cpu_count() or 1, want_tokens=True) for k, v in texts.items()}
        x, o = texts["english"]
        _run(tok, torch, x, o, "first call", refs["english"][1], refs["english"][2])
        assert tok.info(capi.TD_INFO_SPARSE) == 0
        for name in ("english_with_rare_work", "mixed", "code"):
            x, o = texts["english"]
            _run(tok, torch, x, o, "prose", refs["english"][1], refs["english"][2])
            xx, oo = texts[name]
            _run(tok, torch, xx, oo, f"{name} right behind prose", refs[name][1], refs[name][2])
            assert tok.info(capi.TD_INFO_SPARSE) == 1, "the guess was 'sparse' (made from the prose call's counters)"
    finally:
        tok.close()


def test_td_tail_alone_when_its_workgroups_cannot_meet():
    """A grid far larger than what is resident: the launch's first barrier gives up (50 ms) and workgroup 0 walks the phases alone."""
    import torch
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    old = os.environ.get("TD_TAIL_BLOCKS_PER_CU")
    try:
        x, o = _texts()["english_with_rare_work"]
        k = int(np.searchsorted(o, 1 << 20))  # (the first document of the rare work; one workgroup alone is slow: 150 documents of prose, the rare work, 100 more)
        d0, d1 = k - 150, k + 9 + 100
        x, o = x[int(o[d0]): int(o[d1])], (o[d0: d1 + 1] - o[d0]).astype(np.int64)
        _, et, eo = H.ref_tokenizer().encode_batch(x, o, n_threads=os.cpu_count() or 1, want_tokens=True)
        tok.set_option(capi.TD_OPT_SPARSE, 1)
        os.environ["TD_TAIL_BLOCKS_PER_CU"] = "256"
        _run(tok, torch, x, o, "td_tail with 65 536 workgroups", et, eo)
    finally:
        if old is None:
            os.environ.pop("TD_TAIL_BLOCKS_PER_CU", None)
        else:
            os.environ["TD_TAIL_BLOCKS_PER_CU"] = old
        tok.close()


def test_td_far_probe_alone_when_its_workgroups_cannot_meet():
    """The dense sequence's td_far_probe with a grid far larger than what is resident: same fallback, same ids."""
    import torch
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    old = os.environ.get("TD_FAR_PROBE_BLOCKS_PER_CU")
    try:
        x, o = _texts()["english_with_rare_work"]
        k = int(np.searchsorted(o, 1 << 20))
        d0, d1 = k - 150, k + 9 + 100
        x, o = x[int(o[d0]): int(o[d1])], (o[d0: d1 + 1] - o[d0]).astype(np.int64)
        _, et, eo = H.ref_tokenizer().encode_batch(x, o, n_threads=os.cpu_count() or 1, want_tokens=True)
        tok.set_option(capi.TD_OPT_SPARSE, 0)
        os.environ["TD_FAR_PROBE_BLOCKS_PER_CU"] = "256"
        _run(tok, torch, x, o, "td_far_probe with 65 536 workgroups", et, eo)
        assert tok.info(capi.TD_INFO_FAR_PIECES) > 0, "the text has pieces the tile loop's window cannot see through"
    finally:
        if old is None:
            os.environ.pop("TD_FAR_PROBE_BLOCKS_PER_CU", None)
        else:
            os.environ["TD_FAR_PROBE_BLOCKS_PER_CU"] = old
        tok.close()
