"""tokendagger_amd — MI355X-native, tiktoken-compatible tokenizer (drop-in for the TokenDagger hot path).

    import tokendagger_amd as tiktoken            # or: import tokendagger as tiktoken
    enc = tiktoken.Encoding(name="llama4", pat_str=..., mergeable_ranks=..., special_tokens=...)
    enc.encode("Hello, world!")

Public names mirror the reference package (/root/reference/tokendagger/__init__.py:5-25).  The native
pieces are `libtokendagger_hip.so` (HIP kernels + C ABI, include/tokendagger_hip.h) and the pybind11
module `_tokendagger_core`; both must be built (python -c "import __graft_entry__ as g; g.build()").
Importing the tokenizer classes without them raises ImportError — there is no CPU fallback.
`tokendagger_amd.vocab_io` and `tokendagger_amd.capi` import without the native module.
"""
__version__ = "0.1.0"

__all__ = ["Tokenizer", "TokenDaggerError", "load_tokenizer", "create_tokenizer", "Encoding", "llama4_scout", "load_tiktoken_bpe", "core"]


def __getattr__(name):  # lazy: keep `import tokendagger_amd.vocab_io` usable before the extension is built
    if name in __all__:
        from . import capi
        capi.load_library()  # shares the HIP runtime with torch, then loads libtokendagger_hip.so
        try:
            from . import _tokendagger_core as core
        except ImportError as e:
            raise ImportError("tokendagger_amd native extension not found; build it with "
                              "`python -c \"import __graft_entry__ as g; g.build()\"` (no CPU fallback exists)") from e
        from . import wrapper
        g = globals()
        g.update(Tokenizer=wrapper.Tokenizer, TokenDaggerError=wrapper.TokenDaggerError,
                 load_tokenizer=wrapper.load_tokenizer, create_tokenizer=wrapper.create_tokenizer,
                 Encoding=wrapper.Encoding, llama4_scout=wrapper.llama4_scout,
                 load_tiktoken_bpe=wrapper.load_tiktoken_bpe, core=core)
        return g[name]
    raise AttributeError(f"module 'tokendagger_amd' has no attribute {name!r}")
