"""TD_OPT_OVERLAP (round 5): once a handle has seen long pieces, td_long_pieces / td_giant_pieces run beside td_collect_misses ->
td_merge_pieces -> td_copy_dups on two streams (fork behind the lookups, join in front of the scan; parallel branches when the step is
captured into a hipGraph; since round 6 the long pieces keep the caller's stream and the short pieces' chain takes the handle's second one).  Same ids either way, with plain launches and with graph replay, against the compiled reference
(CoreBPE::encode, /root/reference/src/tiktoken/tiktoken.cpp:169-234); the switch must actually engage (TD_INFO_LONG_PIECES >= 2048)."""
from __future__ import annotations

import os

import numpy as np
import pytest

import helpers as H
import td_corpus
from oracle import ref
from tokendagger_amd import capi

pytestmark = pytest.mark.gpu


def test_long_pieces_beside_the_short_ones_change_no_id():
    import torch
    assert ref.available(), "oracle/_ref/libtdref.so is missing: build it where /root/reference exists (oracle/build_ref.sh)"
    pat, mr, special = H.llama4()
    tok = capi.HipTokenizer(pat, mr, special, device=0)
    try:
        x, o = td_corpus.mixed(12 << 20, seed=23)
        # a few pieces above 1 KiB too (td_giant_pieces is on the second stream as well)
        giant = ("\n" + "ab" * 3000 + "\n" + "中" * 900 + "\n").encode("utf-8")
        x = np.concatenate([x, np.frombuffer(giant, dtype=np.uint8)])
        o = np.concatenate([o, [len(x)]]).astype(np.int64)
        _, et, eo = H.ref_tokenizer().encode_batch(x, o, n_threads=os.cpu_count() or 1, want_tokens=True)
        n, nd = len(x), len(o) - 1
        s = torch.cuda.current_stream().cuda_stream
        dt, do = torch.from_numpy(x).cuda(), torch.from_numpy(o).cuda()
        dk = torch.empty(n + 1024, dtype=torch.int32, device="cuda")
        dto = torch.empty(nd + 1, dtype=torch.int64, device="cuda")
        for overlap in (1, 0, 1):
            tok.set_option(capi.TD_OPT_OVERLAP, overlap)
            for graph in (0, 1):
                tok.set_option(capi.TD_OPT_GRAPH, graph)
                for rep in range(4):  # (the first call of a handle runs in line: it has not seen long pieces yet; graph: capture on the 2nd)
                    dk.zero_()
                    dto.zero_()
                    tok.encode_device(dt.data_ptr(), n, do.data_ptr(), nd, dk.data_ptr(), n + 1024, dto.data_ptr(), s)
                    tok.device_status(s)
                    assert tok.info(capi.TD_INFO_LONG_PIECES) >= 2048, "the corpus has enough long pieces for the second stream to be taken"
                    toff = dto.cpu().numpy()
                    assert np.array_equal(toff, eo), f"overlap={overlap} graph={graph} call {rep}: document offsets differ from the reference"
                    got = dk[:int(toff[-1])].cpu().numpy()
                    bad = np.flatnonzero(got != et)
                    assert bad.size == 0, f"overlap={overlap} graph={graph} call {rep}: ids differ from the reference, first at token {int(bad[0])}"
    finally:
        tok.set_option(capi.TD_OPT_GRAPH, 0)
        tok.close()
