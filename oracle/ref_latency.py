"""TEST INFRASTRUCTURE ONLY (tools/gpu_latency.py): single-call latency of the REFERENCE's own Python module on a list of texts.

    python oracle/ref_latency.py <texts.json>      (a JSON list of [name, text]) -> one JSON line {name: microseconds per encode call}

Loads oracle/_ref/refmod/_tokendagger_core*.so (= /root/reference/src/py_binding.cpp + tiktoken.cpp, unmodified, built by
oracle/build_ref.sh) in a process of its own, builds CoreBPE as the reference's wrapper does and times CoreBPE.encode(text, set())
— the call the reference's latency benchmark makes (/root/reference/tests/performance_benchmark.py:413-457) through wrapper.encode."""
from __future__ import annotations

import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import ref_pybench  # noqa: E402  (its loader)

HERE = Path(__file__).resolve().parent


def main():
    texts = json.loads(Path(sys.argv[1]).read_text())
    so = next((HERE / "_ref" / "refmod").glob("_tokendagger_core*.so"))
    core = ref_pybench._load("_tokendagger_core", so)
    vocab_io = ref_pybench._load("td_vocab_io_standalone", HERE.parent / "tokendagger_amd" / "vocab_io.py")
    _, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
    merged = dict(ranks)
    for k, v in special.items():
        merged[k.encode("utf-8")] = v
    items, sitems = [], []
    for tb, r in merged.items():
        it = core.VocabItem(); it.rank = r; it.token_bytes = list(tb); it.token_string = ""
        items.append(it)
    for s, r in special.items():
        it = core.VocabItem(); it.rank = r; it.token_bytes = list(s.encode("utf-8")); it.token_string = s
        sitems.append(it)
    bpe = core.CoreBPE(pat, items, sitems)
    out = {}
    empty = set()
    for name, text in texts:
        n = 200 if len(text) < 2000 else 30 if len(text) < 100000 else 5
        for _ in range(3):
            bpe.encode(text, empty)
        t0 = time.perf_counter()
        for _ in range(n):
            ids, _last = bpe.encode(text, empty)
        out[name] = [round((time.perf_counter() - t0) / n * 1e6, 2), len(ids)]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
