"""Named edge-case inputs for parity tests (UTF-8 source).

Categories mirror what the reference's own scripts exercise:
  - tests/test_tokendagger_vs_tiktoken.py:227-237   basic strings incl. empty / " " / "\\n\\t" / CJK+emoji /
                                                    a string containing special-token text
  - tests/performance_benchmark.py:256-315          minimal, special-token text, unicode, punctuation,
                                                    numbers, repetitive patterns, code, JSON
plus the pre-tokenizer corner cases listed in SURVEY.md section 7 (contractions incl. U+017F, U+180E as
whitespace, overlapping letter classes, look-ahead whitespace rule, long single-class runs).
"""
import json

BASIC = [
    "Hello, world!",
    "The quick brown fox jumps over the lazy dog.",
    "This is a test of the tokenization system.",
    "Special tokens: <|begin_of_text|>Hello<|end_of_text|>",
    "Unicode test: 你好世界 \U0001F30D",
    "Numbers and symbols: 123 456.789 @#$%",
    "",
    " ",
    "\n\t",
]

MINIMAL = ["\n", "\t", "a", "hi", "the", "A", "AB", "Ab", "aB", "1", "12", "123", "1234", "!", "!!", " !", "  !", " a", "  a", "a ",
           "a  ", "a\n", "a \n", "a\n ", "\n\n", " \n ", "\r\n", "\r", "\n\r\n\r", "\t\t\t", " \t \t"]

SPECIAL_TEXT = [
    "<|begin_of_text|>", "<|end_of_text|>", "<|begin_of_text|>Hello<|end_of_text|>", "<|fim_prefix|>code<|fim_suffix|>",
    "Multiple <|begin_of_text|> special <|end_of_text|> tokens", "<|", "|>", "<|not_a_special|>",
]

UNICODE = [
    "Hello 世界", "Café résumé naïve", "こんにちは世界", "\U0001F680\U0001F31F✨\U0001F4AB⭐",
    "\U0001F468‍\U0001F4BB\U0001F469‍\U0001F52C\U0001F9D1‍\U0001F3A8",
    "Ĥëłłø Wörłð", "αβγδεζηθικλμνξοπρστυφχψω",
    "\U0001F1FA\U0001F1F8\U0001F1EC\U0001F1E7\U0001F1EB\U0001F1F7\U0001F1E9\U0001F1EA\U0001F1EF\U0001F1F5",
    "中A b", "中ABCDEFGHIJk", "中ABCDEFGHIJ ", "ABC中DEF", "ǅemal ǈubav", "ʰello", "áb̀c",
    "́a", ".́a", "́ a", "नमस्ते दुनिया", "สวัสดีชาวโลก", "مُحَمَّد",
    "a\u00a0b", "a\u180eb", "a\u3000\u3000b", "a\u2003b", "a\u2028\u2029b", "a\u0085b", "a\u202f\u205fb", "a\x0b\x0cb",
    "\x00\x01\x02", "a\x00b", "\x7f", "\u200b\u200d\ufeff", "\U0010FFFF", "\U00020000\U0002A6D6", "\ufffd",
]

CONTRACTIONS = [
    "it's", "IT'S", "it'S", "don't", "we're", "WE'RE", "I've", "I'm", "they'll", "THEY'LL", "he'd", "x's's", "a'ſ", "a'ſt",
    "a'", "a''", "a'x", "a'r", "a'rx", "a'l", "a'll", "a'lll", "'s", " 's", "中's", "1's", "a 's", "a's's't're've'm'll'd",
    "O'Reilly's", "rock'n'roll", "'tis", "y'all've", "a’s",
]

PUNCT = [
    "!@#$%^&*()_+-={}[]|\\:;\"'<>?,./ ", "Hello, world! How are you? I'm fine.", "Testing... ellipsis... and --- dashes.",
    "(Parentheses) [brackets] {braces} <angles>", "Quote: \"Hello,\" she said. 'Indeed,' he replied.",
    "Code: x = y + z; if (x > 0) { return true; }", "a/b", "a//b", "!/", "!/\n/", "!\n", "!\r\n\r\n", "!\n/!", " /", "//", "/ /",
    "path/to/file.txt", "http://example.com/a?b=c&d=e#f", "...\n\n...", "-\n-\n-", "\n/", "\n!", "\n'", "a\n/b",
]

NUMBERS = ["123456789", "1.234567890", "2024-01-15T14:30:00Z", "Price: $123.45 (was $150.00)", "Version 2.1.3-beta.4",
           "Phone: +1-555-123-4567", "IP: 192.168.1.1:8080", "1234567", "12 345 6789", "٣٤٥٦٧٨٩", "१२३४", "Ⅳ½²³",
           "a1b22c333d4444", "1a", "a1", "1 2  3   4", "0x1F", "1e10", "3.14159265358979"]

WHITESPACE = [" " * 2, " " * 3, " " * 64, " " * 65, " " * 200, "\n" * 3, "\n" * 70, " \n" * 40, "  a", "   a", "a   ", "a  \n  b",
              "\t\ta", " \t a", "a \t", "if x:\n    return y\n\n\n    z = 1\n", "\n \n \n", "  \n", "\n  ", "\r\n\r\n  x", " \r", "\r "]

REPETITIVE = ["a" * 100, "the " * 50, "hello world " * 25, "Lorem ipsum dolor sit amet " * 10, "abcdefghijklmnopqrstuvwxyz" * 4,
              "a" * 64, "a" * 65, "a" * 63, "A" * 100, "aA" * 60, "ab" * 200, "=" * 80, "=" * 64 + "\n", "-" * 65, "#" * 300,
              "中" * 30, "中" * 100, "1" * 100, "é" * 40, "\U0001F600" * 20, "\U0001F600\n" * 40, "x" * 1000, "Z" * 4200,
              " " * 5000, "ab1" * 300, ".," * 100, "a " * 100, "a\n" * 100]

CODE = [
    "\ndef factorial(n):\n    if n <= 1:\n        return 1\n    return n * factorial(n - 1)\n\nprint(factorial(5))\n            ",
    "\nfunction quickSort(arr) {\n    if (arr.length <= 1) return arr;\n    const pivot = arr[Math.floor(arr.length / 2)];\n"
    "    return [...quickSort(arr.filter(x => x < pivot)), pivot];\n}\n",
    "\n#include <iostream>\n#include <vector>\n\nint main() {\n    std::vector<int> v = {3, 1, 4, 1, 5, 9, 2, 6};\n"
    "    for (int i : v) {\n        std::cout << i << \" \";\n    }\n    return 0;\n}\n",
    "std::vector<std::pair<size_t, int>> parts; // get_thread_local_match_data()",
    "x" * 70 + " = " + "y" * 90 + ";",
    "this_is_a_very_long_identifier_name_that_goes_on_and_on_and_on_for_more_than_sixty_four_bytes_total = 1",
    "AAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAAA",
    "aGVsbG8gd29ybGQgdGhpcyBpcyBhIGJhc2U2NCBibG9iIHdpdGhvdXQgYW55IHNwYWNlcyBpbiBpdCBhdCBhbGw=",
]

STRUCTURED = [
    json.dumps({"name": "John Doe", "age": 30, "city": "New York", "hobbies": ["reading", "swimming", "coding"],
                "address": {"street": "123 Main St", "zipcode": "10001"}}, indent=2),
    json.dumps([{"id": i, "value": f"item_{i}"} for i in range(100)]),
]

ALL = {
    "basic": BASIC, "minimal": MINIMAL, "special_text": SPECIAL_TEXT, "unicode": UNICODE, "contractions": CONTRACTIONS,
    "punct": PUNCT, "numbers": NUMBERS, "whitespace": WHITESPACE, "repetitive": REPETITIVE, "code": CODE,
    "structured": STRUCTURED,
}

# decode id lists of the reference's decode test (test_tokendagger_vs_tiktoken.py:356-362, Llama branch)
DECODE_IDS = [[1, 2, 3], [100, 200, 300], [1000, 2000, 3000], list(range(10)), list(range(100, 110))]


def all_strings() -> list[tuple[str, str]]:
    return [(f"{k}[{i}]", s) for k, v in ALL.items() for i, s in enumerate(v)]


# outside the generic split-pattern subset (a capturing group and a back-reference; the lazy quantifier behind it is
# inside the subset since round 3): construction must fail with TD_E_PATTERN
UNSUPPORTED_PATTERN = r"(\w+)\s+\1|\S+?"
# patterns the generic compiler accepts: construction-time negative tests must not use these
SUPPORTED_GENERIC_PATTERNS = [r"\w+|\s+", r"[a-z]+|\s+", r"\p{L}+|\p{N}{1,3}| ?[^\s\p{L}\p{N}]+|\s+"]
