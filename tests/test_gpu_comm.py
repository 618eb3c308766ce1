"""The C ABI's multi-GPU epilogue (td_comm_*: RCCL opened by the tokenizer library itself) on the one GPU a test box has:
a communicator of world size 1 runs the real ncclCommInitRank / ncclAllGather / grouped send-recv code path.  World sizes
2 and 3 are covered on CPU (tests/test_dist.py: gloo carries the table, td_comm_bases turns it into bases)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world_of_one_rccl_through_the_c_abi():
    import torch
    from tokendagger_amd import capi
    uid = capi.comm_unique_id()
    assert len(uid) == capi.TD_COMM_ID_BYTES and any(uid)
    comm = capi.RcclComm(uid, 1, 0, 0)
    s = torch.cuda.current_stream().cuda_stream
    counts = torch.tensor([12345, 67], dtype=torch.int64, device="cuda")
    table = torch.zeros(2, dtype=torch.int64, device="cuda")
    comm.gather_counts(counts.data_ptr(), table.data_ptr(), s)
    torch.cuda.synchronize()
    t = table.cpu().numpy()
    assert t.tolist() == [12345, 67]
    assert capi.comm_bases(t, 0) == (0, 0, 12345, 67)
    toks = torch.arange(12345, dtype=torch.int32, device="cuda")
    root = torch.zeros(12345 + 8, dtype=torch.int32, device="cuda")
    comm.gather_tokens(toks.data_ptr(), t, 0, root.data_ptr(), root.numel(), s)
    torch.cuda.synchronize()
    assert torch.equal(root[:12345], toks)
    with pytest.raises(capi.TokenDaggerHipError):
        comm.gather_tokens(toks.data_ptr(), t, 0, root.data_ptr(), 100, s)  # root buffer too small
    comm.close()
    assert torch.cuda.current_device() == 0
