#!/usr/bin/env python
"""Headline benchmark: GB/s of raw UTF-8 text tokenized (Llama-4-Scout vocab) on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N=1 default; N>1 self-spawns N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Workload = the configuration BASELINE.json's metric is quoted on: ONE 1024 MiB synthetic English corpus (the
reference generator's shape, td_corpus.english; reference: tests/throughput_test.py:246-333,399-416), Llama-4-Scout
vocabulary, CoreBPE::encode semantics.  A "step" is one pass of the hot path (td_encode_device: regex
pre-tokenization + whole-piece lookup + byte-pair merge + packing, all kernels) over this rank's share of the corpus,
which is RESIDENT IN HBM when the timed region starts.

N GPUs (BASELINE config 3, strong scaling): every rank builds the same seeded corpus, takes its contiguous,
byte-balanced document range (tokendagger_amd.dist.shard_documents), and every step ends with the path's only
collective: an RCCL all-gather of per-rank {tokens, documents} -> global token / document bases.  value = the whole
corpus's bytes / the slowest rank's time.  `--scaling weak` keeps --size-mb per GPU instead.

After the timed region EVERY document's ids and offsets are compared with the compiled reference (oracle/_ref, the
unmodified tiktoken.cpp) on the host cores — the oracle is the checker only, never the thing measured.
Rank 0 prints ONE JSON line (fields: README / DESIGN.md section 5).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


# ---------------------------------------------------------------------------------------------- corpora
_UNIT_CACHE: dict = {}


def build_corpus(kind: str, n_bytes: int, seed: int):
    """Seeded synthetic corpus: a 32 MiB generator block tiled to n_bytes (documents stay independent)."""
    import td_corpus
    if kind == "code_files":
        return td_corpus.code_files(n_bytes)
    unit_bytes = min(n_bytes, 32 << 20)
    key = (kind, unit_bytes, seed)
    if key not in _UNIT_CACHE:
        _UNIT_CACHE[key] = getattr(td_corpus, kind)(unit_bytes, seed=seed)
    unit, uo = _UNIT_CACHE[key]
    reps = (n_bytes + unit_bytes - 1) // unit_bytes
    if reps == 1:
        return unit, uo
    x = np.tile(unit, reps)[:n_bytes]
    offs = np.concatenate([uo[:-1] + r * unit_bytes for r in range(reps)] + [[n_bytes]])
    offs = np.unique(offs[offs <= n_bytes]).astype(np.int64)
    if kind != "english":  # non-ASCII corpora: never cut a document inside a character
        while (x[-1] & 0xC0) == 0x80 or x[-1] >= 0xC0:
            x[-1] = 0x20
            if len(x) >= 2 and x[-2] < 0x80:
                break
            x[-2] = 0x20
    return np.ascontiguousarray(x), offs


def kernel_source_sha() -> str:
    """Identity of the device code the traffic numbers in profiles/hbm_traffic.json belong to."""
    h = hashlib.sha256()
    for f in ("td_kernels.hip", "td_common.h", "td_kernels.h"):
        h.update((ROOT / "tokendagger_amd" / "csrc" / f).read_bytes())
    return h.hexdigest()[:16]


def _cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _merged_ranks(ranks: dict, special: dict) -> dict:
    mr = dict(ranks)  # the reference's tests enter the specials as regular tokens too (throughput_test.py:211-213)
    for k, v in special.items():
        mr[k.encode("utf-8")] = v
    return mr


def reference_tokenizer(pat: str, ranks: dict, special: dict):
    from oracle import ref
    if not ref.available():
        raise RuntimeError("oracle/_ref/libtdref.so not built")
    return ref.RefTokenizer(pat, _merged_ranks(ranks, special), special)


def chunk_bounds(offs: np.ndarray, n_bytes: int, n_chunks: int, ascii_only: bool) -> np.ndarray:
    """The reference benchmark's chunking: T x 10 equal slices (tests/throughput_test.py:399-410).  Slices are by
    character there, so only an ASCII corpus can be cut anywhere; otherwise the cuts snap to document starts."""
    import td_corpus
    if ascii_only:
        return td_corpus.chunk_offsets(n_bytes, n_chunks)
    targets = (np.arange(1, n_chunks, dtype=np.int64) * n_bytes) // n_chunks
    cuts = offs[np.searchsorted(offs, targets, side="left").clip(0, len(offs) - 1)]
    return np.unique(np.concatenate([[0], cuts, [n_bytes]])).astype(np.int64)


def cpu_baseline(x: np.ndarray, offs: np.ndarray, ranks: dict, special: dict, pat: str, kind: str):
    """BASELINE.md section 3, "pure C++": T std::threads each calling the reference's CoreBPE::encode(chunk, {}) on
    T x 10 equal slices of the same corpus; T = 1 and T = all hardware threads; 1 warm-up + 3 timed runs, median."""
    cores = os.cpu_count() or 1
    n = int(offs[-1])
    ascii_only = kind == "english"

    def timed(R, sample_bytes: int, threads: int):
        sample_bytes = min(sample_bytes, n)
        if not ascii_only:  # end the sample at a document boundary
            sample_bytes = int(offs[max(1, int(np.searchsorted(offs, sample_bytes, side="right")) - 1)])
        so = offs[:int(np.searchsorted(offs, sample_bytes, side="right"))]
        ch = chunk_bounds(so, sample_bytes, threads * 10, ascii_only)
        # warm-up = one FULL untimed run: the first pass of T fresh threads over chunks of this size pays the page faults
        # of their malloc arenas (measured: 1.02 s, then 0.20 / 0.18 s for the same work) and must not be a sample
        R.encode_batch(x, ch, n_threads=threads, want_tokens=False)
        runs = [R.encode_batch(x, ch, n_threads=threads, want_tokens=False)[0] for _ in range(3)]
        return sample_bytes, len(ch) - 1, statistics.median(runs), runs

    try:
        R = reference_tokenizer(pat, ranks, special)
        # T = all: the whole corpus up to 1 GiB, so the timed region stays above a second on a 256-thread host
        b_all, c_all, s_all, runs_all = timed(R, 1 << 30, cores)
        # T = 1: 48 MiB (about a second per run at the reference's single-thread rate)
        b_1, c_1, s_1, runs_1 = timed(R, 48 << 20, 1)
        # the same work list the GPU is timed on (identical document boundaries): every document of
        # the corpus its own CoreBPE::encode call, dealt to T std::threads; 1 warm-up + 3 runs, median
        R.encode_batch(x, offs, n_threads=cores, want_tokens=False)
        runs_docs = [R.encode_batch(x, offs, n_threads=cores, want_tokens=False)[0] for _ in range(3)]
        s_docs = statistics.median(runs_docs)
        return {
            "value": round(b_all / s_all / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "reference",
            "mib_per_s": round(b_all / s_all / 2**20, 1),
            "sample": f"first {b_all / 2**20:.0f} MiB of the same corpus as {c_all} equal slices (T x 10, the reference "
                      f"benchmark's chunking), CoreBPE::encode per slice on {cores} std::threads, 1 warm-up + 3 runs, median",
            "runs_s": [round(v, 4) for v in runs_all],
            "single_thread": {"value": round(b_1 / s_1 / 1e9, 4), "unit": "GB/s", "mib_per_s": round(b_1 / s_1 / 2**20, 1),
                              "sample": f"first {b_1 / 2**20:.0f} MiB as {c_1} slices, 1 thread, median of 3",
                              "runs_s": [round(v, 4) for v in runs_1]},
            "same_documents": {"value": round(n / s_docs / 1e9, 4), "unit": "GB/s", "mib_per_s": round(n / s_docs / 2**20, 1),
                               "sample": f"the whole corpus on the GPU's own document boundaries ({len(offs) - 1} documents, one CoreBPE::encode "
                                         f"call each), {cores} std::threads, 1 warm-up + 3 runs, median",
                               "runs_s": [round(v, 4) for v in runs_docs]},
            "nproc": cores, "cpu_model": _cpu_model(),
            "python_encode_batch": python_encode_batch_baseline(x, n, cores) if ascii_only else None,
        }
    except Exception as e:  # compiled reference unusable here: time the single-threaded C restatement instead
        from oracle import port
        if not port.available():
            subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")], stdout=subprocess.DEVNULL)
        O = port.OracleTokenizer(ranks, _port_variant(pat))
        sample_docs = max(1, int(np.searchsorted(offs, 8 << 20)))
        s_offs = offs[:sample_docs + 1]
        s_bytes = int(s_offs[-1])
        t0 = time.perf_counter()
        O.encode_batch(x[:s_bytes].tobytes(), s_offs)
        sec = time.perf_counter() - t0
        sys.stderr.write(f"[bench] compiled reference unavailable ({e}); cpu_baseline uses the C restatement\n")
        return {"value": round(s_bytes / sec / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
                "sample": f"first {s_bytes / 2**20:.0f} MiB ({sample_docs} documents), oracle/td_oracle.c, one thread, one run",
                "nproc": cores, "cpu_model": _cpu_model()}


def python_encode_batch_baseline(x: np.ndarray, n: int, cores: int):
    """The reference's benchmark METHOD (tests/throughput_test.py:399-422): Tokenizer.encode_batch(T x 10 slices, num_threads=T)
    through the reference's own pybind module, in a process of its own (oracle/ref_pybench.py).  None when that module was not
    prebuilt (oracle/build_ref.sh needs /root/reference)."""
    import tempfile
    mod = list((ROOT / "oracle" / "_ref" / "refmod").glob("_tokendagger_core*.so")) if (ROOT / "oracle" / "_ref" / "refmod").exists() else []
    if not mod:
        return None
    sample = min(n, 256 << 20)
    out = {}
    try:
        with tempfile.NamedTemporaryFile(suffix=".bin", dir="/tmp") as f:
            x[:sample].tofile(f)
            f.flush()
            for threads in sorted({cores, 8}):
                r = subprocess.run([sys.executable, str(ROOT / "oracle" / "ref_pybench.py"), f.name, str(sample), str(threads), "3"],
                                   capture_output=True, text=True, timeout=600)
                if r.returncode != 0:
                    sys.stderr.write(f"[bench] ref_pybench failed: {r.stderr[-300:]}\n")
                    return None
                j = json.loads(r.stdout.strip().splitlines()[-1])
                sec = statistics.median(j["seconds"])
                out[f"threads_{threads}"] = {"value": round(j["bytes"] / sec / 1e9, 4), "unit": "GB/s", "mib_per_s": round(j["bytes"] / sec / 2**20, 1),
                                             "runs_s": [round(v, 4) for v in j["seconds"]], "chunks": j["chunks"]}
        out["sample"] = (f"first {sample >> 20} MiB, T x 10 equal slices, ONE encode_batch(chunks, num_threads=T) call through the reference's own "
                         f"_tokendagger_core (py_binding.cpp + tiktoken.cpp unmodified): Python threads, list[list[int]] out; 1 warm-up + 3 runs, median")
        return out
    except Exception as e:  # noqa: BLE001
        sys.stderr.write(f"[bench] python_encode_batch baseline unavailable: {e}\n")
        return None


def _port_variant(pat: str):
    from oracle import port
    from tokendagger_amd import vocab_io
    return port.VARIANT_TEKKEN if pat == vocab_io.TEKKEN_PAT_STR else port.VARIANT_LLAMA4


def verify_against_oracle(x, offs, got_tok, got_off, pat, ranks, special, threads: int):
    """Every document's ids and offsets vs the compiled reference; falls back to a sample against the restatement.
    -> (label, ok, detail)"""
    n_docs = len(offs) - 1
    try:
        R = reference_tokenizer(pat, ranks, special)
        # whole documents in big groups keep the reference's work list short; offsets are compared per document
        _, et, eo = R.encode_batch(x, offs, n_threads=max(1, threads), want_tokens=True)
        ok = bool(np.array_equal(eo, got_off) and np.array_equal(et, got_tok))
        detail = "" if ok else _first_mismatch(offs, eo, got_off, et, got_tok)
        return "reference-full", ok, detail
    except Exception as e:
        sys.stderr.write(f"[bench] compiled reference unavailable for verification ({e}); sampling against the restatement\n")
        from oracle import port
        if not port.available():
            subprocess.check_call([str(ROOT / "oracle" / "build_oracle.sh")], stdout=subprocess.DEVNULL)
        O = port.OracleTokenizer(ranks, _port_variant(pat))
        k = max(1, min(n_docs, int(np.searchsorted(offs, offs[0] + (4 << 20)))))
        base = int(offs[0])
        et, eo = O.encode_batch(x[base:int(offs[k])].tobytes(), offs[:k + 1] - base)
        ok = bool(np.array_equal(eo, got_off[:k + 1]) and np.array_equal(et, got_tok[:int(got_off[k])]))
        return "port-sample", ok, "" if ok else "mismatch inside the first 4 MiB"


def verify_special_against_oracle(x, offs, got_tok, got_off, pat, ranks, special, threads: int):
    """allowed_special = all: tiktoken's segmentation restated here (scanning forward, the longest special literal at a
    position is cut out; the reference's own loop, tiktoken.cpp:130-154, is undefined behaviour), every segment encoded by the
    compiled reference as a subject of its own, the special ids put in between.  -> (label, ok, detail)"""
    import re
    R = reference_tokenizer(pat, ranks, special)
    lits = sorted((k.encode("utf-8") for k in special), key=lambda b: -len(b))
    rx = re.compile(b"|".join(re.escape(b) for b in lits))
    ids = {k.encode("utf-8"): v for k, v in special.items()}
    buf = x.tobytes()
    seg_offs, after, doc_first = [0], [], [0]  # segments as documents of the reference; the special id behind each (-1: none)
    seg_src = []
    for d in range(len(offs) - 1):
        lo, hi = int(offs[d]), int(offs[d + 1])
        p = lo
        for m in rx.finditer(buf, lo, hi):
            seg_src.append((p, m.start())); after.append(ids[m.group()]); p = m.end()
        seg_src.append((p, hi)); after.append(-1)
        doc_first.append(len(seg_src))
    seg_text = np.concatenate([x[a:b] for a, b in seg_src]) if seg_src else np.zeros(0, np.uint8)
    lens = np.asarray([b - a for a, b in seg_src], dtype=np.int64)
    so = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    if len(seg_text) == 0:
        seg_text = np.zeros(1, np.uint8)
    _, et, eo = R.encode_batch(seg_text, so, n_threads=max(1, threads), want_tokens=True)
    # stitch
    aft = np.asarray(after, dtype=np.int64)
    extra = (aft >= 0).astype(np.int64)
    seg_out = np.diff(eo) + extra                      # ids each segment contributes, its special included
    out_off = np.concatenate([[0], np.cumsum(seg_out)])
    exp = np.empty(int(out_off[-1]), dtype=np.int32)
    # the segments' own ids
    starts = out_off[:-1]
    idx = np.repeat(starts - eo[:-1], np.diff(eo)) + np.arange(len(et))
    exp[idx] = et
    sp_pos = (out_off[1:] - 1)[aft >= 0]
    exp[sp_pos] = aft[aft >= 0].astype(np.int32)
    exp_doc = out_off[np.asarray(doc_first, dtype=np.int64)]
    ok = bool(np.array_equal(exp_doc, got_off) and np.array_equal(exp, got_tok))
    return "reference-full on every segment between special tokens (tiktoken segmentation restated in bench.py)", ok, "" if ok else _first_mismatch(offs, exp_doc, got_off, exp, got_tok)


def _first_mismatch(offs, eo, go, et, gt) -> str:
    bad = np.nonzero(eo != go)[0]
    d = int(bad[0]) - 1 if len(bad) else -1
    if d < 0:
        m = min(len(et), len(gt))
        i = int(np.nonzero(et[:m] != gt[:m])[0][0]) if m and np.any(et[:m] != gt[:m]) else m
        d = int(np.searchsorted(eo, i, side="right")) - 1
    d = max(d, 0)
    return f"first differing document {d} (bytes {int(offs[d])}..{int(offs[d + 1])})"


# ---------------------------------------------------------------------------------------------- launch
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--corpus", default="english", choices=["english", "mixed", "code", "code_files", "chat"])
    ap.add_argument("--allowed-special", default="none", choices=["none", "all"],
                    help="all: every special token is searched for and cut out on the device (td_encode_device_with_special); "
                         "meant for --corpus chat")
    ap.add_argument("--size-mb", type=int, default=1024, help="MiB of text (whole job for strong scaling, per GPU for weak)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    ap.add_argument("--pattern", default="llama4", choices=["llama4", "tekken", "generic:autogen", "generic:words"],
                    help="split pattern; 'tekken' = the Mistral tekken pattern over the Llama-4 vocabulary, the labelled "
                         "surrogate for BASELINE config 4 (tekken.json is absent from the reference checkout)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only for exercising the multi-rank path with several ranks on ONE GPU (RCCL refuses that)")
    ap.add_argument("--single-document", action="store_true", help="the whole corpus as ONE document (parallelism inside a document)")
    ap.add_argument("--collective", default="torch", choices=["torch", "capi"],
                    help="who issues the step's all-gather of {tokens, documents}: torch.distributed (default) or the C ABI's "
                         "td_comm_gather_counts (RCCL opened by the tokenizer library itself; needs --dist-backend nccl)")
    ap.add_argument("--same-gpu", action="store_true", help="all ranks use cuda:0 (multi-rank logic test on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    return ap.parse_args(argv)


def main():
    a = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if a.gpus > 1 and env_world is None:
        # `python bench.py --gpus N`: become the launcher of N ranks, one per GPU (what the driver does itself
        # with torch.distributed.run); the JSON line of rank 0 passes through on stdout
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(Path(__file__).resolve())] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.call(cmd, env=env))
    world = int(env_world or "1")
    if world != a.gpus and not (a.gpus == 1 and os.environ.get("TD_BENCH_FORCE_DIST") == "1"):
        raise SystemExit(f"bench: --gpus {a.gpus} but WORLD_SIZE={world}; launch N ranks for --gpus N")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if a.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))

    # stdout carries exactly ONE JSON line.  Native libraries write there too (RCCL prints its version banner and its
    # warnings to stdout), so file descriptor 1 is pointed at stderr for the whole run and the JSON line goes out
    # through a private duplicate of the original stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    from tokendagger_amd import capi, vocab_io
    from tokendagger_amd import dist as tdist

    use_dist = world > 1 or os.environ.get("TD_BENCH_FORCE_DIST") == "1"  # the latter: the collective at world size 1
    dist = None
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    gdev = dev if a.dist_backend == "nccl" else torch.device("cpu")  # where the gathered counts live

    name, pat, ranks, special = vocab_io.load_tdv(vocab_io.default_vocab_path())
    if a.pattern == "tekken":
        pat = vocab_io.TEKKEN_PAT_STR
    elif a.pattern == "generic:autogen":  # the reference's tests/autogenned_test.py:66 (skips '_' and non-ASCII letters): the generic engine
        pat = r"[a-zA-Z]+|\s+|[0-9]+|[^\w\s]"
    elif a.pattern == "generic:words":
        pat = r"\w+|[^\w\s]+|\s+"
    tok = capi.HipTokenizer(pat, ranks, special, device=dev.index)
    # a repeated step as ONE hipGraph launch (opt-in since round 4: captured on a private stream of the handle during the
    # warm-up); TD_BENCH_GRAPH=0 times plain launches
    tok.set_option(capi.TD_OPT_GRAPH, 0 if os.environ.get("TD_BENCH_GRAPH") == "0" else 1)

    # ---- the corpus and this rank's share of it -------------------------------------------------------------
    weak = a.scaling == "weak" and world > 1
    total_bytes = (a.size_mb << 20) * (world if weak else 1)
    gx, goffs = build_corpus(a.corpus, total_bytes, seed=1000)  # identical on every rank
    total_bytes = len(gx)  # (the file-set corpus is a whole number of sets, not exactly --size-mb)
    if a.single_document:
        goffs = np.asarray([0, total_bytes], dtype=np.int64)
    g_docs = len(goffs) - 1
    d0, d1 = tdist.shard_documents(goffs, world, rank)
    b0, b1 = int(goffs[d0]), int(goffs[d1])
    x = gx[b0:b1]
    offs = (goffs[d0:d1 + 1] - b0).astype(np.int64)
    n, n_docs = b1 - b0, d1 - d0
    d_text = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_offs = torch.from_numpy(offs).to(dev)
    cap = (n // 2 if a.corpus == "english" else n) + 1024
    d_tok = torch.empty(cap, dtype=torch.int32, device=dev)
    # [0..n_docs] = token offsets (element n_docs = this rank's token total, written by every step),
    # [n_docs+1] = this rank's document count: elements n_docs, n_docs+1 are what the all-gather sends
    d_toff = torch.zeros(n_docs + 2, dtype=torch.int64, device=dev)
    d_toff[n_docs + 1] = n_docs
    tok.reserve(max(n, 1), n_docs + 1)
    stream = torch.cuda.current_stream(dev)
    mine = d_toff[n_docs:n_docs + 2]
    gathered = torch.zeros(2 * world, dtype=torch.int64, device=gdev) if use_dist else None

    comm = None
    if use_dist and a.collective == "capi":
        if a.dist_backend != "nccl":
            raise SystemExit("bench: --collective capi is the RCCL entry of the C ABI: use --dist-backend nccl")
        idt = torch.zeros(capi.TD_COMM_ID_BYTES, dtype=torch.uint8, device=dev)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(capi.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        comm = capi.RcclComm(bytes(idt.cpu().numpy().tobytes()), world, rank, dev.index)

    allowed_ids = sorted(special.values()) if a.allowed_special == "all" else []

    def step():
        if allowed_ids:
            tok.encode_device_with_special(d_text.data_ptr(), n, d_offs.data_ptr(), n_docs, allowed_ids, d_tok.data_ptr(), cap, d_toff.data_ptr(),
                                           stream.cuda_stream)
        else:
            tok.encode_device(d_text.data_ptr(), n, d_offs.data_ptr(), n_docs, d_tok.data_ptr(), cap, d_toff.data_ptr(),
                              stream.cuda_stream)
        if comm is not None:  # the same exchange through the C ABI (td_comm_gather_counts), on the step's own stream
            comm.gather_counts(mine.data_ptr(), gathered.data_ptr(), stream.cuda_stream)
        elif use_dist:  # the path's only exchange: per-rank {tokens, documents} -> global bases
            dist.all_gather_into_tensor(gathered, mine if gdev is dev else mine.cpu())

    def fence():
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # the timed region runs WITHOUT the per-kernel events (a repeated step is then one hipGraph launch, captured during the
    # warm-up); the kernel segments are timed afterwards on a few more steps of the same work with the events on
    tok.set_option(capi.TD_OPT_PROFILE, 0)
    for _ in range(max(a.warmup, 0)):
        step()
    torch.cuda.synchronize(dev)
    tok.device_status(stream.cuda_stream)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    tok.device_status(stream.cuda_stream)
    tok.set_option(capi.TD_OPT_PROFILE, 1)
    tok.profile_read()
    for _ in range(min(max(a.steps, 1), 20)):
        step()
    torch.cuda.synchronize(dev)
    tok.device_status(stream.cuda_stream)
    prof = tok.profile_read_all()
    tok.set_option(capi.TD_OPT_PROFILE, 0)
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=gdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    n_tok = int(d_toff[n_docs].item())
    # ---- what a step costs before it has touched any text: the same launches on ONE tile (8 KiB, one document) -------
    fixed_us = None
    if rank == 0 and n > 8192 and not a.no_cpu_baseline:  # (profiling runs pass --no-cpu-baseline: only the timed steps' launches then)
        tok.set_option(capi.TD_OPT_PROFILE, 0)
        one = torch.tensor([0, 8192], dtype=torch.int64, device=dev)
        t_one = torch.empty(8192 + 1024, dtype=torch.int32, device=dev)
        o_one = torch.empty(2, dtype=torch.int64, device=dev)
        for _ in range(20):
            tok.encode_device(d_text.data_ptr(), 8192, one.data_ptr(), 1, t_one.data_ptr(), 8192 + 1024, o_one.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(200):
            tok.encode_device(d_text.data_ptr(), 8192, one.data_ptr(), 1, t_one.data_ptr(), 8192 + 1024, o_one.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(dev)
        fixed_us = round((time.perf_counter() - t1) / 200 * 1e6, 1)
        tok.device_status(stream.cuda_stream)
        # (the last full step's outputs are still in d_tok / d_toff: the tile went to buffers of its own)
    # ---- parity of what was just timed: EVERY document of this rank's shard vs the compiled reference -----------
    verified, vdetail = None, ""
    if not a.no_verify:
        got_off = d_toff[:n_docs + 1].cpu().numpy()
        got_tok = d_tok[:n_tok].cpu().numpy()
        threads = max(1, (os.cpu_count() or 1) // world)
        if allowed_ids:
            label, ok, vdetail = verify_special_against_oracle(x, offs, got_tok, got_off, pat, ranks, special, threads)
        else:
            label, ok, vdetail = verify_against_oracle(x, offs, got_tok, got_off, pat, ranks, special, threads)
        if use_dist:
            flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=gdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        if not ok:
            raise SystemExit(f"bench: GPU token ids differ from the oracle ({label}) on rank {rank}: {vdetail}")
        verified = label
    # ---- the gathered table: every rank's {tokens, documents}; bases must be the prefix sums --------------------
    tab = None
    if use_dist:
        tab = gathered.view(world, 2).cpu().numpy()
        assert int(tab[rank, 0]) == n_tok and int(tab[rank, 1]) == n_docs, "all-gather returned foreign counts"
        assert int(tab[:, 1].sum()) == g_docs, "documents lost or duplicated by the sharding"
        doc_base, tok_base = int(tab[:rank, 1].sum()), int(tab[:rank, 0].sum())
        assert doc_base == d0, "document base differs from the shard's first document"
        # token base of this rank == number of ids the ranks before it produced (checked against their own totals
        # above through the MIN-reduced verification: every shard's ids equal the reference's)
        assert tok_base >= 0
    if rank == 0:
        job_bytes = total_bytes
        ms_step = elapsed / a.steps * 1e3
        value = job_bytes / (elapsed / a.steps) / 1e9
        job_tok = int(tab[:, 0].sum()) if tab is not None else n_tok
        b_alg = n + 4 * n_tok + 8 * (n_docs + 1)  # SURVEY 8(d), this rank's launch: read text once, write ids + offsets once
        sums, k_n = prof
        avg = {k: v / max(k_n, 1) for k, v in sums.items()}
        k_name = max(avg, key=avg.get) if avg else None
        k_avg_ms = avg[k_name] if k_name else None
        achieved = b_alg / (k_avg_ms * 1e-3) / 1e9 if k_avg_ms else None
        step_achieved = (job_bytes + 4 * job_tok + 8 * (g_docs + 1)) / (ms_step * 1e-3) / 1e9 / world
        # every kernel segment against the bytes IT has to move (its own inputs read once, its own outputs written once;
        # DESIGN.md section 5 lists them), from the HIP events around the segments of the timed steps
        own = {
            "td_prepare+td_mark_docs": n // 8 + 12 * n_docs,                     # zero the document bitmap; offsets in, bits + first documents out
            "td_split_tiles": n + n // 8 + n // 8 + 4 * n_tok + 12 * n_docs,      # text + document bits in; START bits, one slot per piece, document slots out
            "td_pack_tokens": 8 * n_tok + 20 * n_docs,                            # slots in, ids out; per document: offset + slot in, token offset out
        }
        per_kernel = {k: {"ms": round(v, 4), "necessary_bytes": own.get(k), "gbs": round(own[k] / (v * 1e-3) / 1e9, 1) if own.get(k) and v > 0 else None,
                          "frac_of_hbm_peak": round(own[k] / (v * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if own.get(k) and v > 0 else None}
                      for k, v in ({k: s_ / max(k_n, 1) for k, s_ in sums.items()}).items()}
        traffic, traffic_note = None, "no PMC pass recorded for this workload"
        tfile = ROOT / "profiles" / "hbm_traffic.json"
        if tfile.exists():
            try:
                tj = json.loads(tfile.read_text())
                ent = tj.get(f"{a.corpus}_{a.pattern}_{n >> 20}")
                if ent and tj.get("kernel_source_sha") == kernel_source_sha():
                    traffic = ent.get(k_name)
                    traffic_note = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this exact kernel source ({tj.get('source', '')}); "
                                    f"2 x FETCH + WRITE per the gfx950 correction; all kernels of a step: {ent.get('_all')} B "
                                    f"= {ent.get('_all', 0) / max(b_alg, 1):.2f} x algorithmic")
                elif ent:
                    traffic_note = "profiles/hbm_traffic.json was measured on a different kernel source: not quoted"
            except Exception:
                pass
        out = {
            "metric": "GB/s raw text tokenized (Llama-4 vocab, 1024 MB corpus)",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "u8",
            "data": f"synthetic (seeded td_corpus.{a.corpus}, 32 MiB generator block tiled)" if a.corpus != "code_files" else
                    "the reference's code_performance_benchmark file set (tests/golden/code_corpus.npz), tiled",
            "config": {"workload": f"Llama-4-Scout vocab{' + tekken split pattern (config 4 surrogate: tekken.json missing)' if a.pattern == 'tekken' else (' + generic split pattern ' + pat if a.pattern.startswith('generic') else '')}, "
                                   f"{job_bytes >> 20} MiB {'of the code_performance_benchmark file set' if a.corpus == 'code_files' else 'synthetic ' + a.corpus + ' text'}, {g_docs} documents, "
                                   f"CoreBPE::encode semantics{' with allowed_special = all (searched for and cut out on the device)' if a.allowed_special == 'all' else ''}, input resident in HBM",
                       "bytes": job_bytes, "tokens": job_tok, "docs": g_docs,
                       "bytes_rank0": n, "tokens_rank0": n_tok, "docs_rank0": n_docs,
                       "parallelism": (f"dp{world}: contiguous byte-balanced document shards, "
                                       f"{('RCCL (C ABI td_comm_gather_counts)' if a.collective == 'capi' else 'RCCL (torch.distributed)') if a.dist_backend == 'nccl' else 'gloo (host)'} all-gather of "
                                       f"{{tokens, documents}} every step" if world > 1 else
                                       (f"single GPU + the RCCL all-gather at world size 1 ({'C ABI td_comm_gather_counts' if a.collective == 'capi' else 'torch.distributed'})" if use_dist and a.dist_backend == "nccl" else "single GPU")),
                       "verified_vs_oracle": verified},
            "roofline": {"bound": "hbm", "kernel": k_name, "achieved": round(achieved, 2) if achieved else None,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if achieved else None,
                         "traffic": traffic, "traffic_note": traffic_note, "algorithmic_bytes_per_launch": b_alg,
                         "kernel_ms_avg": round(k_avg_ms, 4) if k_avg_ms else None, "launches_timed": k_n,
                         "kernel_timing_note": "HIP events on the launch stream around every kernel segment, on steps of the same work run right behind the timed region (the timed region itself runs without them)",
                         "all_kernels_ms_avg": {k: round(v, 4) for k, v in avg.items()},
                         "per_kernel": per_kernel, "fixed_overhead_us": fixed_us,
                         "whole_step": {"achieved": round(step_achieved, 2), "frac": round(step_achieved / HBM_PEAK_GBS, 5),
                                        "note": "algorithmic bytes of the job / ms_per_step / GPUs: every kernel and gap of a step"}},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(gx, goffs, ranks, special, pat, a.corpus)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
