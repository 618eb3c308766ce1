"""Single-call latency of the drop-in surface next to the reference's own module, on text shapes of the categories the reference's
latency benchmark uses (/root/reference/tests/performance_benchmark.py:239-387: minimal, short, sentences, paragraphs, code, unicode,
numbers and punctuation, repetitive, whitespace, long documents — the strings here are this repository's own): enc.encode(text) of this
package (GPU: one launch per call for inputs of at most 4 KiB, the general path above) and CoreBPE.encode(text, set()) of the reference
(CPU; oracle/ref_latency.py in a process of its own).  -> a table for profiles/, one line per text."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import helpers as H
import td_corpus
import tokendagger as tiktoken

eng = td_corpus.english(1 << 20, seed=5)[0].tobytes().decode()
mix = td_corpus.mixed(1 << 18, seed=5)[0].tobytes().decode("utf-8", "ignore")
code = td_corpus.code(1 << 18, seed=5)[0].tobytes().decode()
T = [("minimal: empty", ""), ("minimal: one blank", " "), ("minimal: newline", "\n"), ("minimal: one letter", "a"), ("minimal: one CJK character", "中"),
     ("short: hello", "Hello, world!"), ("short: two words", "good morning"), ("short: number", "3.14159"), ("short: emoji", "😀🚀"),
     ("sentence: 45 B", "The quick brown fox jumps over the lazy dog. "), ("sentence: 90 B", "The quick brown fox jumps over the lazy dog. " * 2),
     ("sentence: question", "What time does the next train to Berlin leave, and from which platform?"),
     ("sentence: quoted", 'She said, "It\'s not what you think — it\'s worse," and left.'),
     ("paragraph: 300 B", eng[:300]), ("paragraph: 600 B", eng[:600]), ("paragraph: 900 B", eng[:900]), ("paragraph: 1500 B", eng[:1500]),
     ("paragraph: 2500 B", eng[:2500]), ("paragraph: 4000 B", eng[:4000]),
     ("code: 100 B", code[:100]), ("code: 400 B", code[:400]), ("code: 1000 B", code[:1000]), ("code: 4000 B", code[:4000]),
     ("code: json", json.dumps({"name": "tokenizer", "ids": list(range(40)), "nested": {"a": [1.5, 2.25, None], "ok": True}})),
     ("code: html", "<div class=\"row\"><span id='x1'>value &amp; more</span><a href=\"https://example.org/a?b=c\">link</a></div>" * 3),
     ("code: sql", "SELECT u.id, COUNT(*) AS n FROM users u JOIN orders o ON o.user_id = u.id WHERE o.total > 100.0 GROUP BY u.id ORDER BY n DESC LIMIT 10;"),
     ("unicode: mixed scripts 140 B", "Hello 世界! Привет мир! مرحبا بالعالم! 🌍🚀✨ "), ("unicode: mixed scripts 420 B", "Hello 世界! Привет мир! مرحبا بالعالم! 🌍🚀✨ " * 3),
     ("unicode: mixed corpus 1000 B", mix[:400]), ("unicode: mixed corpus 4000 B", mix[:1600]), ("unicode: CJK sentence", "今天天气很好,我们一起去公园散步吧。" * 3),
     ("unicode: accents", "naïve café résumé jalapeño Ångström straße über coöperate " * 3), ("unicode: emoji sequences", "👨‍💻👩‍🔬🏳️‍🌈🇺🇸👍🏽❤️🔥🎉 " * 4),
     ("numbers: digits", "1234567890 " * 20), ("numbers: decimals", "3.14159 2.71828 1.41421 0.57721 6.02e23 -273.15 " * 4),
     ("numbers: dates and times", "2024-01-15T10:30:00Z 1999/12/31 23:59:59 +0100 " * 4), ("punctuation: runs", "!!! ??? ... --- *** /// ((( ))) [[[ ]]] {{{ }}} <<< >>> " * 3),
     ("punctuation: operators", "a+=b; c<<=2; d->e; f::g; h&&i||!j; k==l!=m<=n>=o; " * 3),
     ("repetitive: one letter x 100", "a" * 100), ("repetitive: one letter x 1000", "a" * 1000), ("repetitive: word x 200", "test " * 200),
     ("repetitive: abc x 300", "abc" * 300), ("repetitive: dashes x 500", "-" * 500),
     ("whitespace: blanks x 100", " " * 100), ("whitespace: newlines x 100", "\n" * 100), ("whitespace: tabs and blanks", "\t \t  \n" * 40),
     ("whitespace: indented code", ("        if x:\n            return y\n" * 20)),
     ("long: 16 KB English", eng[:16000]), ("long: 64 KB English", eng[:64000]), ("long: 256 KB English", eng[:256000]), ("long: 1 MB English", eng[:1000000]),
     ("long: 64 KB code", code[:64000]), ("long: 64 KB mixed scripts", mix[:26000]),
     ("long: 100 KB of one line", ("lorem ipsum dolor sit amet consectetur " * 2600)[:100000]),
     ("edge: long word 60 B", "x" * 20 + "y" * 20 + "z" * 20), ("edge: long identifier", "thisIsAVeryLongCamelCaseIdentifierNameThatGoesOnAndOn_with_snake_case_too_1234"),
     ("edge: url", "https://subdomain.example.com:8080/path/to/resource.html?query=value&other=123#fragment"),
     ("edge: base64", "QWxhZGRpbjpvcGVuIHNlc2FtZQ==" * 8), ("edge: hex", "0xDEADBEEFCAFEBABE0123456789ABCDEF" * 6)]
pat, mr, sp = H.llama4()
enc = tiktoken.Encoding("llama4", pat_str=pat, mergeable_ranks=mr, special_tokens=sp)
mine = {}
for name, text in T:
    n = 200 if len(text) < 2000 else 30 if len(text) < 100000 else 5
    for _ in range(5):
        enc.encode(text)
    t0 = time.perf_counter()
    for _ in range(n):
        ids = enc.encode(text)
    mine[name] = ((time.perf_counter() - t0) / n * 1e6, len(ids))
ref = {}
if any(os.scandir(os.path.join(ROOT, "oracle", "_ref", "refmod"))) if os.path.isdir(os.path.join(ROOT, "oracle", "_ref", "refmod")) else False:
    with tempfile.NamedTemporaryFile("w", suffix=".json", dir="/tmp", delete=False) as f:
        json.dump(T, f)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "ref_latency.py"), f.name], capture_output=True, text=True, timeout=900)
    if r.returncode == 0:
        ref = json.loads(r.stdout.strip().splitlines()[-1])
    else:
        print("reference module failed:", r.stderr[-300:])
print(f"# enc.encode(text), microseconds per call: this package (MI355X) | the reference's own module (CPU, {os.cpu_count()} hardware threads present, one used) | ids (must agree)")
print(f"{'text':<36} {'bytes':>8} {'this, us':>10} {'reference, us':>14} {'ratio':>7} {'ids':>8}")
for name, text in T:
    us, k = mine[name]
    rv = ref.get(name)
    same = "" if not rv or rv[1] == k else f"  IDS DIFFER ({rv[1]})"
    print(f"{name:<36} {len(text.encode()):>8} {us:>10.1f} {(f'{rv[0]:.1f}' if rv else '-'):>14} {(f'{rv[0] / us:.2f}x' if rv else '-'):>7} {k:>8}{same}")
