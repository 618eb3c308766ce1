"""GPU fuzz: td_encode_device (the fused tile loop and everything behind it) against the compiled reference on batches of nasty
documents — mixed scripts, white-space runs of every kind, contractions, digits, emoji sequences, long runs — for the four pattern
families, through both launch sequences (forced and chosen), batches of a few KB to a few MB.
    python tools/gpu_fuzz_vs_reference.py <seconds per family> [seed]
(the generator is the desk fuzz's, tools/fuzz_twin_vs_reference.py; the reference: CoreBPE::encode, /root/reference/src/tiktoken/tiktoken.cpp:169-234,
compiled into oracle/_ref)"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import helpers as H
from tokendagger_amd import capi

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ALPH = list(" \t\n\r\x0b\x0c  　 '’\"`.,;:!?-_=+*/\\|(){}[]<>@#$%^&~0123456789aAbBcdeEfgstTlLvVrRmMdD")
ALPH += list("éÉßñüÜøŒǅʰΩωжЖאبहि中文字あアー가각😀👍🏽́‍­½Ⅷ٣३")
WORDS = ["the", " The", "'s", "'T", "'ll", "'LL", "n't", "'Re", "don't", "I'M", "  ", "\n\n", " \n", "\r\n", "123", "1234567", "3.14", "x=1", "__init__", "camelCaseWord",
         "http://a.b/c?d=e", "привет", "мир", "你好世界", "こんにちは", "안녕하세요", "😀😀", "👍🏽", "naïve", "Straße", "\t\t", " \t ", "....", "----", "====", "a" * 70, "ab" * 40, "中" * 30]
pat, mr, special = H.llama4()
FAM = {"llama4": (pat, H.ref_tokenizer), "tekken": (H.TEKKEN_PAT, H.ref_tokenizer_tekken), "cl100k": (H.CL100K_PAT, H.ref_tokenizer_cl100k), "gpt2": (H.GPT2_PAT, H.ref_tokenizer_gpt2)}
s = torch.cuda.current_stream().cuda_stream
total_docs = total_bytes = 0
ok = True
for which, (p, Rf) in FAM.items():
    R = Rf()
    tok = capi.HipTokenizer(p, mr, special, device=0)
    rng = random.Random(seed0 * 1000003 + hash(which) % 1000)
    t0 = time.time(); it = 0; nd = 0; nb = 0
    while time.time() - t0 < budget and ok:
        docs = []
        scale = rng.choice([1, 1, 4, 20, 100])
        for _ in range(rng.randint(1, 60) * scale):
            r = rng.random()
            if r < 0.03: docs.append(b"")
            elif r < 0.5: docs.append("".join(rng.choice(ALPH) for _ in range(rng.randint(1, 300))).encode("utf-8"))
            elif r < 0.9: docs.append("".join(rng.choice(WORDS) if rng.random() < 0.6 else rng.choice(ALPH) for _ in range(rng.randint(1, 400))).encode("utf-8"))
            elif r < 0.99: docs.append((rng.choice(WORDS) * rng.randint(1, 300)).encode("utf-8")[:9000].decode("utf-8", "ignore").encode("utf-8"))
            else: docs.append((rng.choice(WORDS) * rng.randint(300, 4000)).encode("utf-8")[:60000].decode("utf-8", "ignore").encode("utf-8"))
        text, offs = H.pack_docs(docs)
        x = np.frombuffer(text, dtype=np.uint8)
        if len(x) == 0:
            continue
        _, et, eo = R.encode_batch(x, offs, n_threads=os.cpu_count() or 1, want_tokens=True)
        n, ndoc = len(x), len(offs) - 1
        dt, do = torch.from_numpy(x.copy()).cuda(), torch.from_numpy(offs).cuda()
        dk = torch.empty(n + 1024, dtype=torch.int32, device="cuda")
        dto = torch.empty(ndoc + 1, dtype=torch.int64, device="cuda")
        for sp in rng.sample([-1, 0, 1], 2):
            tok.set_option(capi.TD_OPT_SPARSE, sp)
            dk.zero_(); dto.zero_()
            tok.encode_device(dt.data_ptr(), n, do.data_ptr(), ndoc, dk.data_ptr(), n + 1024, dto.data_ptr(), s)
            tok.device_status(s)
            toff = dto.cpu().numpy()
            got = dk[:int(toff[-1])].cpu().numpy() if np.array_equal(toff, eo) else None
            if got is None or not np.array_equal(got, et):
                ok = False
                for d in range(ndoc):
                    a = dk[int(toff[d]):int(toff[d + 1])].cpu().numpy(); b = et[eo[d]:eo[d + 1]]
                    if not np.array_equal(a, b):
                        print("MISMATCH", which, "iteration", it, "sparse option", sp, "document", d, "of", ndoc, repr(docs[d][:300])); break
                break
        it += 1; nd += ndoc; nb += n
    tok.close()
    total_docs += nd; total_bytes += nb
    print(f"{which}: {it} batches, {nd} documents, {nb / 1e6:.1f} MB in {time.time() - t0:.0f} s: {'equal to the compiled reference' if ok else 'MISMATCH'}", flush=True)
    if not ok:
        break
print(f"total {total_docs} documents, {total_bytes / 1e6:.1f} MB: {'ok' if ok else 'FAILED'}")
sys.exit(0 if ok else 1)
