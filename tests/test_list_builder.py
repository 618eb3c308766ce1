"""The result lists of `encode_batch(list[str]) -> list[list[int]]` (csrc/py_binding.cpp: IntCache): one shared int object
per token id, reference counts added per distinct id, the lists' item arrays filled by threads with the GIL released.
The reference returns the same type through pybind11's STL caster (/root/reference/src/py_binding.cpp:25-39 + wrapper.py:212-235);
what a caller can observe — list of lists of int, equal to the ids — must be the same, and the reference counts must come
back to where they were when the lists die.  No device needed: the module-level hook `_ids_to_lists` runs the builder alone."""
import gc
import sys

import numpy as np
import pytest

# (the extension module is linked against libtokendagger_hip.so, whose HIP runtime has to be the one torch bundles when torch is
# in the process: capi.load_library() sees to that BEFORE the module is loaded — pytest imports this file while collecting the
# GPU suite too, and a system HIP runtime bound here left torch without a GPU in round 4's first closing run)
capi = pytest.importorskip("tokendagger_amd.capi")
try:
    capi.load_library()
except ImportError as e:  # not built
    pytest.skip(str(e), allow_module_level=True)
core = pytest.importorskip("tokendagger_amd._tokendagger_core")


def _case(n, n_docs, seed, max_id=200_000):
    rng = np.random.default_rng(seed)
    ids = (rng.zipf(1.3, size=n) % (max_id + 1)).astype(np.int32) if n else np.zeros(0, np.int32)
    cuts = np.sort(rng.integers(0, n + 1, size=max(n_docs - 1, 0))) if n_docs else np.zeros(0, np.int64)
    offs = np.concatenate([[0], cuts, [n]]).astype(np.int64) if n_docs else np.asarray([0], np.int64)
    return ids, offs


@pytest.mark.parametrize("n,n_docs", [(0, 0), (0, 3), (1, 1), (10, 3), (1000, 50), (65535, 7), (65536, 7), (300_000, 1000), (3_000_000, 40),
                                      (2_000_000, 150_000)])
def test_lists_equal_the_ids(n, n_docs):
    ids, offs = _case(n, n_docs, seed=n + n_docs)
    got = core._ids_to_lists(ids, offs)
    assert type(got) is list and len(got) == len(offs) - 1
    for d in (0, len(got) // 2, len(got) - 1):
        if 0 <= d < len(got):
            assert type(got[d]) is list and all(type(v) is int for v in got[d][:50])
    assert got == [ids[offs[d]:offs[d + 1]].tolist() for d in range(len(offs) - 1)]


def test_reference_counts_are_balanced_and_lists_are_ordinary_lists():
    ids = np.full(200_000, 77_777, np.int32)   # (above the small-call path: counted per id, not per element)
    ids[::1000] = 5
    offs = np.asarray([0, 150_000, 150_000, 200_000], np.int64)
    lists = core._ids_to_lists(ids, offs)
    x = lists[0][1]
    assert x == 77_777 and sys.getrefcount(x) >= 200_000 - 200  # every slot holds a reference to the one shared object
    lists[0].append(1)          # ordinary, mutable lists
    lists[0][0] = -3
    lists[2].clear()
    del lists
    gc.collect()
    assert sys.getrefcount(x) == 2   # ours + the argument's: the slots' references are gone with the lists


def test_bad_arguments():
    with pytest.raises(Exception):
        core._ids_to_lists(np.asarray([1, 2, 3], np.int32), np.asarray([0, 5], np.int64))
    with pytest.raises(Exception):
        core._ids_to_lists(np.asarray([1, -2, 3], np.int32), np.asarray([0, 3], np.int64))
    big = np.full(70_000, -1, np.int32)
    with pytest.raises(Exception):
        core._ids_to_lists(big, np.asarray([0, 70_000], np.int64))
