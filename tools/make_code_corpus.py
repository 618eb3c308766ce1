#!/usr/bin/env python
"""Generate tests/golden/code_corpus.npz: BASELINE config 5's input, i.e. the file set the reference's
tests/code_performance_benchmark.py selects from its own repository (find_code_files, :268-319: the listed
extensions under src/, tokendagger/, tests/ and the top level, minus extern/build/...; files above 1 MiB characters
and empty files are skipped, :338-345), each file one document, plus the COMPILED REFERENCE's ids for every file
(CoreBPE::encode(content, {}), what benchmark_single_file times, :360-383).  Run in the build container only
(/root/reference is not on the GPU box); the output is committed test DATA, no reference code is executed from it.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import helpers as H  # noqa: E402

REPO = Path("/root/reference")
EXTS = ['.py', '.cpp', '.c', '.h', '.hpp', '.js', '.ts', '.java', '.rs', '.go', '.rb', '.php', '.cs', '.swift', '.kt', '.scala',
        '.sh', '.bat', '.ps1', '.sql', '.json', '.xml', '.yaml', '.yml', '.md', '.txt', '.makefile', '.cmake']
EXCLUDE = {'extern', 'build', '__pycache__', '.git', '.vscode', 'node_modules', 'target', 'dist', 'out', '.pytest_cache'}
INCLUDE = {'src', 'tokendagger', 'tests'}


def should_include(path: Path) -> bool:  # code_performance_benchmark.py:284-301 (paths are absolute there)
    if any(p in EXCLUDE for p in path.parts):
        return False
    if any(p in INCLUDE for p in path.parts):
        return True
    return len(path.parts) <= 2


def read_safely(p: Path):
    for enc in ('utf-8', 'latin-1', 'cp1252'):
        try:
            return p.read_text(encoding=enc)
        except (UnicodeDecodeError, UnicodeError):
            continue
    return None


def main():
    files = []
    for ext in EXTS:
        for f in REPO.glob(f"**/*{ext}"):
            if f.is_file() and should_include(f):
                files.append(f.relative_to(REPO))
    for special in ('Makefile', 'CMakeLists.txt'):
        if (REPO / special).exists():
            files.append(Path(special))
    files = sorted(set(files), key=lambda p: ((REPO / p).stat().st_size, str(p)))
    R = H.ref_tokenizer()
    docs, names, enc, enc_offs = [], [], [], [0]
    for rel in files:
        content = read_safely(REPO / rel)
        if content is None or len(content) > 1024 * 1024 or not content.strip():
            print("skip", rel)
            continue
        b = content.encode("utf-8")
        docs.append(b); names.append(str(rel))
        e = R.encode(b)
        enc.append(e); enc_offs.append(enc_offs[-1] + len(e))
    text, offs = H.pack_docs(docs)
    out = ROOT / "tests" / "golden" / "code_corpus.npz"
    np.savez_compressed(out, text=np.frombuffer(text, dtype=np.uint8), offsets=offs, names=np.asarray(names),
                        enc=np.concatenate(enc).astype(np.int32), enc_offsets=np.asarray(enc_offs, dtype=np.int64))
    print(f"{len(docs)} files, {len(text)} bytes, {enc_offs[-1]} ids -> {out} ({out.stat().st_size} bytes)")
    for nm, d in zip(names, docs):
        print(f"  {len(d):8d}  {nm}")


if __name__ == "__main__":
    main()
