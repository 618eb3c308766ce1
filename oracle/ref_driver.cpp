// TEST INFRASTRUCTURE ONLY (oracle/): C-ABI driver around the UNMODIFIED reference tokenizer core.
//
// This translation unit is compiled together with /root/reference/src/tiktoken/tiktoken.cpp (where
// it lies; no reference source is copied into this repo) into oracle/_ref/libtdref.so by
// oracle/build_ref.sh.  It gives tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() a
// way to run the real reference (`tiktoken::CoreBPE`, tiktoken.hpp:38-88) on raw byte buffers:
//   - tdref_encode            -> CoreBPE::encode(text, {})            tiktoken.cpp:169-234
//   - tdref_encode_ordinary   -> CoreBPE::encode_ordinary(text)       tiktoken.cpp:156-167
//   - tdref_split             -> CoreBPE::split_text(text, 0, len)    tiktoken.cpp:70-128 (private)
//   - tdref_decode            -> CoreBPE::decode_bytes(tokens)        tiktoken.cpp:236-255
//   - tdref_time_encode_batch -> N std::threads each calling CoreBPE::encode on whole documents
//                                (the "pure C++" CPU baseline of BASELINE.md section 3)
// Nothing under tokendagger_amd/ may link or load this library.

#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

// split_text is private in the reference class; the oracle needs piece boundaries to pin the
// pre-tokenizer separately from the merge loop.
#define private public
#include "tiktoken.hpp"
#undef private

namespace {
thread_local std::string g_err;

struct Ref {
    tiktoken::CoreBPE* bpe;
};

std::vector<VocabItem> make_items(int64_t n, const uint8_t* bytes, const int64_t* offs,
                                  const int32_t* ranks, bool with_string) {
    std::vector<VocabItem> v;
    v.reserve(n);
    for (int64_t i = 0; i < n; ++i) {
        VocabItem it;
        it.rank = ranks[i];
        it.token_bytes.assign(bytes + offs[i], bytes + offs[i + 1]);
        if (with_string) it.token_string.assign((const char*)bytes + offs[i], offs[i + 1] - offs[i]);
        v.push_back(std::move(it));
    }
    return v;
}
}  // namespace

extern "C" {

const char* tdref_last_error() { return g_err.c_str(); }

void* tdref_create(const char* pattern, int64_t n_vocab, const uint8_t* bytes, const int64_t* offs,
                   const int32_t* ranks, int64_t n_special, const uint8_t* sbytes,
                   const int64_t* soffs, const int32_t* sranks) {
    try {
        auto vocab = make_items(n_vocab, bytes, offs, ranks, false);
        auto special = make_items(n_special, sbytes, soffs, sranks, true);
        Ref* r = new Ref;
        r->bpe = new tiktoken::CoreBPE(std::string(pattern), vocab, special);
        return r;
    } catch (const std::exception& e) {
        g_err = e.what();
        return nullptr;
    }
}

void tdref_destroy(void* h) {
    Ref* r = (Ref*)h;
    if (!r) return;
    delete r->bpe;
    delete r;
}

static int64_t copy_out(const std::vector<int>& v, int32_t* out, int64_t cap) {
    if ((int64_t)v.size() > cap) {
        g_err = "output capacity too small";
        return -2;
    }
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];
    return (int64_t)v.size();
}

// CoreBPE::encode with an empty allowed-special set (the path every BASELINE config times).
int64_t tdref_encode(void* h, const uint8_t* text, int64_t len, int32_t* out, int64_t cap,
                     int32_t* last_piece_token_len) {
    try {
        emhash8::HashSet<std::string> none;
        auto res = ((Ref*)h)->bpe->encode(std::string((const char*)text, len), none);
        if (last_piece_token_len) *last_piece_token_len = res.second;
        return copy_out(res.first, out, cap);
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// CoreBPE::encode with a caller-supplied allowed-special set (reference path has UB, SURVEY 8a/A6;
// exposed only so tests can document what the reference does).
int64_t tdref_encode_special(void* h, const uint8_t* text, int64_t len, const char* const* allowed,
                             int64_t n_allowed, int32_t* out, int64_t cap) {
    try {
        emhash8::HashSet<std::string> set;
        for (int64_t i = 0; i < n_allowed; ++i) set.insert(std::string(allowed[i]));
        auto res = ((Ref*)h)->bpe->encode(std::string((const char*)text, len), set);
        return copy_out(res.first, out, cap);
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int64_t tdref_encode_ordinary(void* h, const uint8_t* text, int64_t len, int32_t* out, int64_t cap) {
    try {
        auto res = ((Ref*)h)->bpe->encode_ordinary(std::string((const char*)text, len));
        return copy_out(res, out, cap);
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Piece END offsets of split_text(text, 0, len); returns the number of pieces.
int64_t tdref_split(void* h, const uint8_t* text, int64_t len, int64_t* ends, int64_t cap) {
    try {
        std::string s((const char*)text, len);
        auto pieces = ((Ref*)h)->bpe->split_text(s, 0, s.size());
        if ((int64_t)pieces.size() > cap) {
            g_err = "output capacity too small";
            return -2;
        }
        int64_t pos = 0;
        for (size_t i = 0; i < pieces.size(); ++i) {
            pos += (int64_t)pieces[i].size();
            ends[i] = pos;
        }
        return (int64_t)pieces.size();
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// the pieces themselves (split_text returns strings: with a pattern that skips text, their lengths do not add up to offsets)
int64_t tdref_split_bytes(void* h, const uint8_t* text, int64_t len, uint8_t* out, int64_t cap_bytes, int64_t* lens, int64_t cap) {
    try {
        std::string s((const char*)text, len);
        auto pieces = ((Ref*)h)->bpe->split_text(s, 0, s.size());
        if ((int64_t)pieces.size() > cap) { g_err = "output capacity too small"; return -2; }
        int64_t pos = 0;
        for (size_t i = 0; i < pieces.size(); ++i) {
            if (pos + (int64_t)pieces[i].size() > cap_bytes) { g_err = "output capacity too small"; return -2; }
            memcpy(out + pos, pieces[i].data(), pieces[i].size());
            pos += (int64_t)pieces[i].size();
            lens[i] = (int64_t)pieces[i].size();
        }
        return (int64_t)pieces.size();
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

int64_t tdref_decode(void* h, const int32_t* toks, int64_t n, uint8_t* out, int64_t cap) {
    try {
        std::vector<int> v(toks, toks + n);
        auto res = ((Ref*)h)->bpe->decode_bytes(v);
        if ((int64_t)res.size() > cap) {
            g_err = "output capacity too small";
            return -2;
        }
        if (!res.empty()) memcpy(out, res.data(), res.size());
        return (int64_t)res.size();
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Batch encode on n_threads std::threads (work-stealing over documents), CoreBPE::encode(doc, {}).
// out_offsets[n_docs+1] receives token offsets, out_tokens (capacity cap) the ids (may be NULL to
// only time).  Returns seconds spent inside the threaded encode region, or <0 on error.
double tdref_time_encode_batch(void* h, const uint8_t* text, const int64_t* doc_offsets,
                               int64_t n_docs, int n_threads, int32_t* out_tokens, int64_t cap,
                               int64_t* out_offsets, int64_t* n_tokens) {
    Ref* r = (Ref*)h;
    std::vector<std::vector<int>> per_doc((size_t)n_docs);
    std::atomic<int64_t> next{0};
    std::atomic<int> failed{0};
    std::string err;
    auto work = [&]() {
        emhash8::HashSet<std::string> none;
        for (;;) {
            int64_t d = next.fetch_add(1);
            if (d >= n_docs) break;
            try {
                std::string s((const char*)text + doc_offsets[d], doc_offsets[d + 1] - doc_offsets[d]);
                per_doc[(size_t)d] = r->bpe->encode(s, none).first;
            } catch (const std::exception& e) {
                if (!failed.exchange(1)) err = e.what();
            }
        }
    };
    auto t0 = std::chrono::steady_clock::now();
    if (n_threads <= 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int i = 0; i < n_threads; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    auto t1 = std::chrono::steady_clock::now();
    if (failed.load()) {
        g_err = err;
        return -1.0;
    }
    int64_t total = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        if (out_offsets) out_offsets[d] = total;
        if (out_tokens) {
            if (total + (int64_t)per_doc[(size_t)d].size() > cap) {
                g_err = "output capacity too small";
                return -2.0;
            }
            for (size_t i = 0; i < per_doc[(size_t)d].size(); ++i) out_tokens[total + i] = per_doc[(size_t)d][i];
        }
        total += (int64_t)per_doc[(size_t)d].size();
    }
    if (out_offsets) out_offsets[n_docs] = total;
    if (n_tokens) *n_tokens = total;
    return std::chrono::duration<double>(t1 - t0).count();
}

// Which PCRE2 / Unicode tables the oracle is actually running on.
int tdref_pcre2_version(char* buf, int cap, char* ubuf, int ucap) {
    char tmp[64] = {0}, utmp[64] = {0};
    pcre2_config_8(PCRE2_CONFIG_VERSION, tmp);
    pcre2_config_8(PCRE2_CONFIG_UNICODE_VERSION, utmp);
    snprintf(buf, cap, "%s", tmp);
    snprintf(ubuf, ucap, "%s", utmp);
    return 0;
}

}  // extern "C"
