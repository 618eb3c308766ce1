import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Collection order of the GPU suite (the driver runs `pytest -x -m gpu`): the core parity evidence first — Llama-4 ids against
# the oracle and the compiled reference, full-size, fused loop — then the pattern families and the rows next to the hot path,
# and what exercises options, threads, streams and communicators LAST, so that a peripheral failure can never hide the
# headline evidence again.  Files not listed keep their alphabetical place in the middle.
_ORDER_FIRST = ["test_gpu_parity.py", "test_gpu_fullsize.py", "test_gpu_fused.py", "test_python_api.py", "test_special_reference.py",
                "test_gpu_special_device.py", "test_gpu_small_decode.py"]
_ORDER_LAST = ["test_gpu_comm.py", "test_gpu_clone.py"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        name = Path(str(item.fspath)).name
        if name in _ORDER_FIRST:
            return (0, _ORDER_FIRST.index(name))
        if name in _ORDER_LAST:
            return (2, _ORDER_LAST.index(name))
        return (1, 0)
    items.sort(key=key)  # (stable: the order inside a file and among unlisted files stays)


# VERDICT r4 (parity evidence must not be skippable): the reference-parity tests are `skipif(not ref.available())` because the compiled
# reference (oracle/_ref/libtdref.so, git-ignored) can only be built where /root/reference exists.  On a CPU-only checkout that is a
# skip; in a GPU run (`-m gpu`) it is a FAILURE — a green GPU suite must mean "compared with the reference", never "skipped".
@pytest.hookimpl(tryfirst=True)
def pytest_runtest_setup(item):
    if item.get_closest_marker("gpu") is None:
        return
    for m in item.iter_markers(name="skipif"):
        if m.args and m.args[0] and "oracle/_ref" in str(m.kwargs.get("reason", "")):
            pytest.fail("GPU parity test without its checker: oracle/_ref is missing (build it where /root/reference exists: "
                        "oracle/build_ref.sh; the prebuilt .so travels with the snapshot).  " + str(m.kwargs.get("reason")), pytrace=False)


@pytest.fixture(scope="session")
def fullsize_hashes():
    """Hashes of the compiled reference's ids and offsets on the full-size corpora (tools/make_fullsize_golden.py): one per 2^18 ids."""
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "fullsize_hashes.npz")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "llama4_golden.npz", allow_pickle=True)


@pytest.fixture(scope="session")
def tekken_golden():
    """Compiled-reference outputs for the Mistral tekken split pattern over the Llama-4 vocabulary, on the documents
    of llama4_golden.npz (tools/make_golden.py tekken)."""
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "tekken_style_golden.npz", allow_pickle=True)


@pytest.fixture(scope="session")
def cl100k_golden():
    """Compiled-reference outputs for the cl100k_base / Llama-3 split pattern over the Llama-4 vocabulary, on the
    documents of llama4_golden.npz (tools/make_golden.py cl100k)."""
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "cl100k_style_golden.npz", allow_pickle=True)


@pytest.fixture(scope="session")
def gpt2_golden():
    """Compiled-reference outputs for the GPT-2 (r50k_base / p50k_base) split pattern over the Llama-4 vocabulary, on
    the documents of llama4_golden.npz (tools/make_golden.py gpt2)."""
    import numpy as np
    return np.load(ROOT / "tests" / "golden" / "gpt2_style_golden.npz", allow_pickle=True)
