import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
