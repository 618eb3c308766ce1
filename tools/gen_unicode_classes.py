#!/usr/bin/env python
"""Generate the per-code-point pre-tokenizer class table by PROBING the PCRE2 runtime in this image.

The reference compiles its split pattern with PCRE2_UTF|PCRE2_UCP (tiktoken.cpp:51-58) against the
system PCRE2 (10.39 -> Unicode 14.0.0 here).  Python's `unicodedata` is a different Unicode version,
so the classes are obtained from PCRE2 itself: for every property used by the supported split
patterns we run `[prop]+` over one subject that contains every scalar value in order and read the
match ranges.  Output: a two-stage table (256-code-point blocks) as a C include, written to BOTH
  tokendagger_amd/csrc/generated/unicode_classes.inc   (product: host tables -> HBM)
  oracle/generated/unicode_classes.inc                 (oracle restatement)

Class ids (4 bits) — see tokendagger_amd/csrc/td_classes.h:
  0 OTHER  1 APOS(')  2 SLASH(/)  3 SP(U+0020)  4 WS(other \\s, not CR/LF)  5 CRLF
  6 UP(Lu|Lt)  7 LW(Ll)  8 LB(Lm|Lo)  9 MK(M)  10 NUM(N)
"""
import ctypes
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
pcre = ctypes.CDLL("libpcre2-8.so.0")
pcre.pcre2_compile_8.restype = ctypes.c_void_p
pcre.pcre2_compile_8.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32,
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t), ctypes.c_void_p]
pcre.pcre2_match_data_create_from_pattern_8.restype = ctypes.c_void_p
pcre.pcre2_match_data_create_from_pattern_8.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
pcre.pcre2_match_8.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t,
                               ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
pcre.pcre2_get_ovector_pointer_8.restype = ctypes.POINTER(ctypes.c_size_t)
pcre.pcre2_get_ovector_pointer_8.argtypes = [ctypes.c_void_p]
pcre.pcre2_jit_compile_8.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
PCRE2_UTF, PCRE2_UCP, NO_UTF_CHECK = 0x00080000, 0x00020000, 0x40000000


def versions():
    a = ctypes.create_string_buffer(64); b = ctypes.create_string_buffer(64)
    pcre.pcre2_config_8(11, a); pcre.pcre2_config_8(10, b)
    return a.value.decode(), b.value.decode()


# subject = every Unicode scalar value in order
cps = np.array([c for c in range(0x110000) if not (0xD800 <= c <= 0xDFFF)], dtype=np.int64)
subject = "".join(map(chr, cps)).encode("utf-8")
lens = np.where(cps < 0x80, 1, np.where(cps < 0x800, 2, np.where(cps < 0x10000, 3, 4)))
starts = np.zeros(len(cps) + 1, dtype=np.int64)
np.cumsum(lens, out=starts[1:])
assert starts[-1] == len(subject)


def members(prop: str) -> np.ndarray:
    err = ctypes.c_int(0); eo = ctypes.c_size_t(0)
    pat = ("(?:" + prop + ")++").encode()  # possessive: no backtracking stack on 40k-long runs
    code = pcre.pcre2_compile_8(pat, len(pat), PCRE2_UTF | PCRE2_UCP, ctypes.byref(err), ctypes.byref(eo), None)
    assert code, (prop, err.value)
    pcre.pcre2_jit_compile_8(code, 1)
    md = pcre.pcre2_match_data_create_from_pattern_8(code, None)
    ov = pcre.pcre2_get_ovector_pointer_8(md)
    mask = np.zeros(len(cps), dtype=bool)
    pos = 0
    while pos < len(subject):
        rc = pcre.pcre2_match_8(code, subject, len(subject), pos, NO_UTF_CHECK, md, None)
        if rc < 0:
            assert rc == -1, (prop, rc)  # only NOMATCH may end the walk
            break
        a, b = ov[0], ov[1]
        i0 = int(np.searchsorted(starts, a)); i1 = int(np.searchsorted(starts, b))
        assert starts[i0] == a and starts[i1] == b
        mask[i0:i1] = True
        pos = b
    return mask


def main():
    ver, uver = versions()
    print(f"PCRE2 {ver}, Unicode {uver}", file=sys.stderr)
    m = {p: members(p) for p in [r"\s", r"\p{L}", r"\p{N}", r"\p{M}", r"\p{Lu}", r"\p{Ll}", r"\p{Lt}",
                                 r"\p{Lm}", r"\p{Lo}", r"\S"]}
    S, L, N, M = m[r"\s"], m[r"\p{L}"], m[r"\p{N}"], m[r"\p{M}"]
    Lu, Ll, Lt, Lm, Lo = m[r"\p{Lu}"], m[r"\p{Ll}"], m[r"\p{Lt}"], m[r"\p{Lm}"], m[r"\p{Lo}"]
    assert (L == (Lu | Ll | Lt | Lm | Lo)).all()
    assert (Lu.astype(int) + Ll + Lt + Lm + Lo).max() == 1
    assert not (S & (L | N | M)).any() and not (L & N).any() and not (L & M).any() and not (N & M).any()
    assert (m[r"\S"] == ~S).all()
    print("sizes:", {k: int(v.sum()) for k, v in m.items()}, file=sys.stderr)
    cls = np.zeros(0x110000, dtype=np.uint8)
    c = np.zeros(len(cps), dtype=np.uint8)
    c[S] = 4
    c[Lu | Lt] = 6
    c[Ll] = 7
    c[Lm | Lo] = 8
    c[M] = 9
    c[N] = 10
    cls[cps] = c
    assert cls[0x20] == 4 and cls[0x0A] == 4 and cls[0x0D] == 4
    cls[0x20] = 3
    cls[0x0A] = 5
    cls[0x0D] = 5
    assert cls[0x27] == 0 and cls[0x2F] == 0
    cls[0x27] = 1
    cls[0x2F] = 2
    ws = [hex(x) for x in np.nonzero((cls == 3) | (cls == 4) | (cls == 5))[0]]
    print("\\s members:", ws, file=sys.stderr)
    # surrogates (unreachable from valid UTF-8) stay OTHER
    blocks = cls.reshape(-1, 256)
    uniq, inv = np.unique(blocks, axis=0, return_inverse=True)
    print(f"{len(uniq)} unique 256-blocks -> stage2 {uniq.size} B, stage1 {len(inv) * 2} B", file=sys.stderr)
    lines = [
        "// GENERATED by tools/gen_unicode_classes.py — do not edit.",
        f"// Probed from PCRE2 {ver} (Unicode {uver}) with PCRE2_UTF|PCRE2_UCP semantics.",
        "// class ids: 0 OTHER 1 APOS 2 SLASH 3 SP 4 WS 5 CRLF 6 UP(Lu|Lt) 7 LW(Ll) 8 LB(Lm|Lo) 9 MK(M) 10 NUM(N)",
        f"#define TD_UCLS_NBLOCKS {len(uniq)}",
        "static const unsigned short td_ucls_stage1[4352] = {",
    ]
    inv = inv.reshape(-1)
    for i in range(0, len(inv), 32):
        lines.append(" " + ",".join(str(int(x)) for x in inv[i:i + 32]) + ",")
    lines.append("};")
    lines.append(f"static const unsigned char td_ucls_stage2[{uniq.size}] = {{")
    flat = uniq.reshape(-1)
    for i in range(0, len(flat), 64):
        lines.append(" " + ",".join(str(int(x)) for x in flat[i:i + 64]) + ",")
    lines.append("};")
    text = "\n".join(lines) + "\n"
    for out in [ROOT / "tokendagger_amd/csrc/generated/unicode_classes.inc",
                ROOT / "oracle/generated/unicode_classes.inc"]:
        out.parent.mkdir(parents=True, exist_ok=True)
        out.write_text(text)
        print("wrote", out, len(text), file=sys.stderr)


if __name__ == "__main__":
    main()
