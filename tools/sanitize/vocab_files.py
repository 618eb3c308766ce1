"""Mutated vocabulary files (tiktoken .model, HF config, tekken.json, the wrapper's two JSON files) through the ASan/UBSan build of
td_vocab.cpp's loaders: loaded or rejected, never a report.  usage: python tools/sanitize/vocab_files.py <seed> <files> <harness>"""
import sys, os, random, json, base64, tempfile, subprocess
import sys, os, random, json, base64, tempfile
seed = int(sys.argv[1]); iters = int(sys.argv[2])
rng = random.Random(seed)
d = tempfile.mkdtemp()
toks = [bytes([i]) for i in range(256)] + [b"ab", b"the", " world".encode(), "é".encode(), b"\n\n"]
model = b"".join(base64.b64encode(t) + b" " + str(i).encode() + b"\n" for i, t in enumerate(toks))
hf = json.dumps({"added_tokens_decoder": {"300": {"content": "<|a|>", "special": True}, "301": {"content": "<|b|>", "special": True}}, "x": [1, 2.5e3, None, True, {"y": "é😀"}]}).encode()
tek = json.dumps({"config": {"pattern": "\\w+|\\s+", "default_vocab_size": 270, "default_num_special_tokens": 3, "version": "v3"},
                  "vocab": [{"rank": i, "token_bytes": base64.b64encode(t).decode(), "token_str": None} for i, t in enumerate(toks)],
                  "special_tokens": [{"rank": 0, "token_str": "<unk>"}]}).encode()
vj = json.dumps({"vocab": [{"rank": i, "token_bytes": list(t), "token_string": ""} for i, t in enumerate(toks)]}).encode()
def mutate(b):
    b = bytearray(b)
    for _ in range(rng.randint(1, 6)):
        r = rng.random()
        if not b: break
        if r < 0.3: b[rng.randrange(len(b))] = rng.randrange(256)
        elif r < 0.5: del b[rng.randrange(len(b)):rng.randrange(len(b)) + rng.randint(1, 40)]
        elif r < 0.7: p = rng.randrange(len(b)); b[p:p] = bytes(rng.choice(b'{}[]",:\\ \n0123456789-+eE.tfn\x00\xff') for _ in range(rng.randint(1, 8)))
        elif r < 0.85: b = b[:rng.randrange(len(b) + 1)]
        else: p = rng.randrange(len(b)); b[p:p] = b[p:p + rng.randint(1, 200)] * rng.randint(1, 3)
    return bytes(b)
sp = json.dumps({"<|a|>": 300, "<|b|>": 301, "é": 302}).encode()
d2 = tempfile.mkdtemp(); args = []
for it in range(iters):
    kind = it % 5
    src = [model, hf, tek, vj, sp][kind]
    path = os.path.join(d2, "f%d_%d" % (kind, it))
    open(path, "wb").write(mutate(src))
    args += ["thkjs"[kind], path]
    if len(args) >= 400 or it == iters - 1:
        p = subprocess.run([sys.argv[3]] + args, capture_output=True, text=True)
        if p.returncode != 0:
            print("SANITIZER REPORT / crash, rc", p.returncode); print(p.stderr[-3000:]); sys.exit(1)
        for a in args[1::2]: os.remove(a)
        args = []
print("asan fuzz seed", seed, "clean,", iters, "files")
