"""The C-ABI shared library loads here (no GPU) and exports every symbol include/tokendagger_hip.h declares;
constructing a tokenizer without a HIP device fails loudly instead of falling back to a CPU path."""
import ctypes
import re
from pathlib import Path

import pytest

import cases
import helpers as H

ROOT = Path(__file__).resolve().parents[1]


def _declared():
    hdr = (ROOT / "include" / "tokendagger_hip.h").read_text()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(td_[a-z_]+)\s*\(", hdr)))


def test_header_symbols_are_exported():
    import __graft_entry__ as g
    g.build_hip()
    from tokendagger_amd import capi
    lib = capi.load_library()
    names = _declared()
    assert len(names) >= 13
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/tokendagger_hip.h but not exported"
    assert set(capi.EXPORTS) == set(names)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tokendagger_amd import capi
    pat, mr, special = H.llama4()
    small = dict(list(mr.items())[:300])
    with pytest.raises(capi.TokenDaggerHipError) as e:
        capi.HipTokenizer(pat, small, {}, device=0)
    assert e.value.code == 7 and "no CPU path" in str(e.value)


def test_unsupported_pattern_rejected_before_touching_the_gpu():
    from tokendagger_amd import capi
    with pytest.raises(capi.TokenDaggerHipError) as e:
        capi.HipTokenizer(cases.UNSUPPORTED_PATTERN, {b"a": 0}, {}, device=0)  # (capturing group + back-reference: outside the generic subset)
    assert e.value.code == 2 and "not supported" in str(e.value)


def test_encoding_constructor_rejects_unsupported_pattern_on_cpu():
    """The construction-time contract the GPU test test_python_api.py::test_attributes_and_errors relies on, checked without
    a GPU: the pattern is compiled before any device is touched, so the failure is TD_E_PATTERN (2) here as well, through the
    Python surface (TokenDaggerError) and through the C ABI; and the patterns the generic compiler accepts do NOT fail with
    TD_E_PATTERN (on a CPU box they get as far as the missing device)."""
    import cases
    import tokendagger
    from tokendagger_amd import capi
    with pytest.raises(tokendagger.TokenDaggerError):
        tokendagger.Encoding(name="bad", pat_str=cases.UNSUPPORTED_PATTERN, mergeable_ranks={b"a": 0})
    with pytest.raises(capi.TokenDaggerHipError) as e:
        capi.HipTokenizer(cases.UNSUPPORTED_PATTERN, {b"a": 0}, {}, device=0)
    assert e.value.code == 2
    import torch
    if torch.cuda.is_available():
        return
    for pat in cases.SUPPORTED_GENERIC_PATTERNS:
        with pytest.raises(capi.TokenDaggerHipError) as e:
            capi.HipTokenizer(pat, {bytes([b]): b for b in range(256)}, {}, device=0)
        assert e.value.code != 2, (pat, str(e.value))


def test_product_does_not_reference_the_oracle():
    # nothing under tokendagger_amd/ or include/ may import, link or open oracle/ or the CPU twin
    bad = []
    for p in list((ROOT / "tokendagger_amd").rglob("*")) + list((ROOT / "include").rglob("*")) + [ROOT / "tokendagger" / "__init__.py"]:
        if p.is_file() and p.suffix in {".py", ".cpp", ".h", ".hip", ".inc"}:
            t = p.read_text(errors="ignore")
            for needle in ("oracle/", "oracle.", "libtdref", "libtdoracle", "libtdtwin", "td_twin"):
                for line in t.splitlines():
                    if needle in line and not line.lstrip().startswith(("//", "#", "*", "\"\"\"")) and "tests/twin" not in line:
                        bad.append((str(p.relative_to(ROOT)), line.strip()[:100]))
    assert not bad, bad
