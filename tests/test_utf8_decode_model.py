"""CPU model of the class-mask phase's UTF-8 decode (tokendagger_amd/csrc/td_kernels.hip, td_split_tiles, "non-ASCII" branch of phase 1):
the branch-free form of round 6 — one dword of four bytes, a mask compare for the continuation bytes, selects for the length — against the
form it replaced (per-byte tests with && and a three-way ?: for the code point) and against Python's own decoder, for every lead byte with
every combination of representative following bytes.  (What the reference does with such text: PCRE2 under PCRE2_NO_UTF_CHECK on valid
UTF-8, /root/reference/src/tiktoken/tiktoken.cpp:91; the classification of malformed sequences is this package's, td_common.h classify_at.)"""
import itertools


def declared_len(b: int) -> int:  # td_common.h utf8_declared_len
    if 0xC2 <= b <= 0xDF:
        return 2
    if 0xE0 <= b <= 0xEF:
        return 3
    if 0xF0 <= b <= 0xF4:
        return 4
    return 1


def old_form(b, c1, c2, c3):
    need = declared_len(b) - 1
    ok = need > 0 and (c1 & 0xC0) == 0x80 and (need < 2 or (c2 & 0xC0) == 0x80) and (need < 3 or (c3 & 0xC0) == 0x80)
    if need == 1:
        c = ((b & 0x1F) << 6) | (c1 & 0x3F)
    elif need == 2:
        c = ((b & 0x0F) << 12) | ((c1 & 0x3F) << 6) | (c2 & 0x3F)
    else:
        c = ((b & 0x07) << 18) | ((c1 & 0x3F) << 12) | ((c2 & 0x3F) << 6) | (c3 & 0x3F)
    return ok, need, c


def new_form(b, c1, c2, c3):
    x = b | (c1 << 8) | (c2 << 16) | (c3 << 24)
    need = declared_len(x & 0xFF) - 1
    cm = (0xC0C0C000 >> (8 * (3 - need))) & 0xC0C0C000
    ok = (need > 0) & ((x & cm) == (cm & 0x80808000))
    d1, d2, d3 = (x >> 8) & 0x3F, (x >> 16) & 0x3F, (x >> 24) & 0x3F
    c = (((x & 0xFF) & (0x3F >> need)) << 6) | d1
    c = ((c << 6) | d2) if need >= 2 else c
    c = ((c << 6) | d3) if need >= 3 else c
    return bool(ok), need, c


FOLLOW = [0x00, 0x20, 0x7F, 0x80, 0x8F, 0x90, 0x9F, 0xA0, 0xBF, 0xC0, 0xC2, 0xE0, 0xF0, 0xFF]


def test_branch_free_decode_equals_the_form_it_replaced():
    for b in range(256):
        for c1, c2, c3 in itertools.product(FOLLOW, repeat=3):
            o, n = old_form(b, c1, c2, c3), new_form(b, c1, c2, c3)
            assert o[0] == n[0] and o[1] == n[1], (hex(b), hex(c1), hex(c2), hex(c3), o, n)
            if o[0]:
                assert o[2] == n[2], (hex(b), hex(c1), hex(c2), hex(c3), o, n)


def test_branch_free_decode_equals_python_on_every_scalar_value_it_accepts():
    # the kernel accepts a character when `ok` and the value is at most U+10FFFF and no surrogate (tbj); overlong forms of three and four
    # bytes decode to a small value and are classified by that value, as classify_at (td_common.h) does — Python rejects those, so only
    # well-formed sequences are compared here, all of them
    for cp in itertools.chain(range(0x80, 0xD800), range(0xE000, 0x110000, 7), [0x10FFFF]):
        enc = chr(cp).encode("utf-8")
        bs = list(enc) + [0x41] * (4 - len(enc))
        ok, need, c = new_form(*bs[:4])
        assert ok and need == len(enc) - 1 and c == cp, (hex(cp), ok, need, hex(c))
