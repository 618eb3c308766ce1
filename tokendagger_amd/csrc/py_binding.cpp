// Python extension `_tokendagger_core`: the reference's FFI surface (class names, method names,
// argument meaning, exception type: /root/reference/src/py_binding.cpp:7-49) re-implemented over the
// C ABI of libtokendagger_hip.so (include/tokendagger_hip.h).  No tokenization happens in this file;
// it marshals Python objects to flat buffers, releases the GIL and calls td_*.
//
//   VocabItem            rank:int, token_bytes:list[int], token_string:str         (py_binding.cpp:11-15)
//   TiktokenError        raised for every TD_E_* status                             (py_binding.cpp:18)
//   CoreBPE(pattern, vocab, special_vocab)                                          (py_binding.cpp:22-24)
//     .encode_ordinary(text) -> list[int]                                           (:25-29)
//     .encode(text, allowed_special:set[str]) -> (list[int], last_piece_token_len)  (:30-39)
//     .decode_bytes(tokens) -> list[int]                                            (:40-44)
//     .special_tokens() -> list[str]                                                (:45-46)
//     .encode_with_special_tokens(text) -> list[int]                                (:47-49)
//   plus array-in/array-out methods the reference does not have (encode_batch_*, *_numpy), because a
//   Python list of ints per document caps throughput far below what the GPU path delivers.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <algorithm>
#include <cstring>
#include <map>
#include <chrono>
#include <thread>
#include <sys/mman.h>
#include <memory>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "tokendagger_hip.h"

namespace py = pybind11;

struct VocabItem {
    int rank = 0;
    std::vector<unsigned char> token_bytes;
    std::string token_string;
};

class TiktokenError : public std::runtime_error {
public:
    explicit TiktokenError(const std::string& m) : std::runtime_error(m) {}
};

namespace {

void pack(const std::vector<VocabItem>& v, bool use_string, std::vector<uint8_t>& bytes, std::vector<int64_t>& offs,
          std::vector<int32_t>& ranks) {
    offs.assign(1, 0);
    for (const auto& it : v) {
        if (use_string) bytes.insert(bytes.end(), it.token_string.begin(), it.token_string.end());
        else bytes.insert(bytes.end(), it.token_bytes.begin(), it.token_bytes.end());
        offs.push_back((int64_t)bytes.size());
        ranks.push_back(it.rank);
    }
    if (bytes.empty()) bytes.push_back(0);
}

// allowed special strings -> concatenated bytes + offsets (td_encode_*_with_special_strs)
void pack_allowed(const std::set<std::string>& allowed, std::vector<uint8_t>& bytes, std::vector<int64_t>& offs) {
    offs.assign(1, 0);
    for (const auto& s : allowed) {
        bytes.insert(bytes.end(), s.begin(), s.end());
        offs.push_back((int64_t)bytes.size());
    }
    if (bytes.empty()) bytes.push_back(0);
}

// Fresh memory of megabytes that is about to be written once from end to end (the packed texts, the ids, the item arrays of long
// lists): ask for huge pages — 4 KiB at a time it is a page fault per 512 ids, 107 000 of them for the lists of 256 MiB of English,
// and the faults of a process's threads do not scale with the threads.  Only the 2 MiB-aligned interior; a hint, never an error
// (transparent huge pages set to "never": nothing happens).
inline void hint_huge(void* p, size_t bytes) {
#ifdef MADV_HUGEPAGE
    constexpr uintptr_t H = (uintptr_t)2 << 20;
    static const bool on = !(getenv("TD_HUGE_PAGES") && atoi(getenv("TD_HUGE_PAGES")) == 0);
    if (!on || bytes < 2 * H) return;
    const uintptr_t lo = ((uintptr_t)p + H - 1) & ~(H - 1), hi = ((uintptr_t)p + bytes) & ~(H - 1);
    if (hi > lo) (void)madvise((void*)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}

// uninitialised id buffer sized for the worst case (one id per input byte): no capacity miss, no second encode, and
// only the pages that receive ids are ever touched
struct IdBuf {
    std::unique_ptr<int32_t[]> p;
    int64_t cap;
    explicit IdBuf(size_t n_bytes) : p(new int32_t[n_bytes + 16]), cap((int64_t)n_bytes + 16) { hint_huge(p.get(), (n_bytes + 16) * 4); }
    int32_t* data() { return p.get(); }
};

// ---- list[str] -> one buffer, ids -> list[list[int]]: the two ends of encode_batch -------------------------------------
// The reference builds its result through pybind11's STL caster: one PyLong per id under the GIL (~30 ns each), the same
// wall this binding hit in round 3 (0.29 GB/s with the GPU path at 27 GB/s behind it: 1.0 x the reference).  Here
//  * the texts are not copied into std::strings: their UTF-8 is taken where CPython keeps it (PyUnicode_AsUTF8AndSize) and
//    copied into ONE buffer by a few threads with the GIL released;
//  * the int objects are SHARED: one PyLong per token id, made once per tokenizer (ints are immutable; a list of ids holds
//    references, and nothing in Python's semantics promises fresh objects).  A call counts how often each id occurs (threads,
//    no GIL), adds that count to the id's reference count once (under the GIL: 200 000 additions instead of 55 M
//    Py_INCREFs), allocates the lists, and then threads store the pointers into the lists' item arrays with the GIL
//    released — the lists are not reachable from anywhere yet, so these are plain memory writes.
inline int pool_threads(size_t work_items) {
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t want = work_items / (1u << 18) + 1;  // a thread per 256 Ki items
    return (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(hw ? hw : 4, 32), want));
}
template <class F>
void parallel_ranges(size_t n, int threads, F&& f) {  // f(lo, hi, thread index)
    if (threads <= 1 || n == 0) { f((size_t)0, n, 0); return; }
    std::vector<std::thread> th;
    for (int k = 1; k < threads; ++k) th.emplace_back([&, k] { f(n * k / threads, n * (k + 1) / threads, k); });
    f((size_t)0, n / threads, 0);
    for (auto& x : th) x.join();
}

class IntCache {  // one shared PyLong per token id
public:
    ~IntCache() { clear(); }
    void clear() {
        for (PyObject* o : objs_) Py_XDECREF(o);
        objs_.clear();
    }
    // Made ONCE (GIL held), for every id the tokenizer can produce, and never resized: threads of other encode_batch calls on
    // the same tokenizer read the table with the GIL released.
    void ensure(int64_t max_id) {
        if (!objs_.empty()) return;
        std::vector<PyObject*> v((size_t)std::max<int64_t>(max_id, 0) + 1, nullptr);
        for (size_t i = 0; i < v.size(); ++i) {
            v[i] = PyLong_FromLong((long)i);
            if (!v[i]) {
                for (size_t k = 0; k < i; ++k) Py_DECREF(v[k]);
                throw py::error_already_set();
            }
        }
        objs_.swap(v);
    }
    bool covers(int64_t id) const { return id < (int64_t)objs_.size(); }
    // ids the table does not hold (never from a tokenizer's own encode): plain lists, one fresh int per id
    static py::list plain_lists(const int32_t* ids, const int64_t* offs, int64_t n_docs) {
        py::list outer((size_t)n_docs);
        for (int64_t d = 0; d < n_docs; ++d) {
            const Py_ssize_t len = (Py_ssize_t)(offs[d + 1] - offs[d]);
            PyObject* l = PyList_New(len);
            if (!l) throw py::error_already_set();
            for (Py_ssize_t i = 0; i < len; ++i) {
                PyObject* o = PyLong_FromLong((long)ids[offs[d] + i]);
                if (!o) { Py_DECREF(l); throw py::error_already_set(); }
                PyList_SET_ITEM(l, i, o);
            }
            PyList_SET_ITEM(outer.ptr(), (Py_ssize_t)d, l);
        }
        return outer;
    }
    // ids[0..n) cut at offs[0..n_docs] -> list[list[int]] (GIL held on entry and on return)
    py::list lists(const int32_t* ids, const int64_t* offs, int64_t n_docs) {
        const int64_t n = offs[n_docs];
        // The bulk path below writes PyListObject's fields itself (ob_item from PyMem_Malloc, allocated, size set last) and adds reference
        // counts in bulk: that is the layout and the allocator pairing of the default CPython builds 3.8 .. 3.13 (list_dealloc frees ob_item
        // with PyMem_Free).  Free-threaded builds (Py_GIL_DISABLED: a _PyListArray header in front of the items, per-thread reference
        // counts), PyPy and versions nobody has looked at take the portable path — one PyList_New(len) + one reference at a time.
#if defined(Py_GIL_DISABLED) || defined(PYPY_VERSION) || PY_VERSION_HEX < 0x03080000 || PY_VERSION_HEX >= 0x030E0000
        constexpr bool kBulkLists = false;
#else
        constexpr bool kBulkLists = true;
#endif
        if (!kBulkLists || n < (1 << 16)) {  // small calls: one reference at a time (no histogram of the id range to clear)
            int32_t mx = -1;
            for (int64_t i = 0; i < n; ++i) {
                if (ids[i] < 0) throw std::runtime_error("negative token id");
                mx = std::max(mx, ids[i]);
            }
            ensure(std::max<int64_t>(mx, max_id_hint));
            if (!covers(mx)) return plain_lists(ids, offs, n_docs);
            py::list outer((size_t)n_docs);
            for (int64_t d = 0; d < n_docs; ++d) {
                const Py_ssize_t len = (Py_ssize_t)(offs[d + 1] - offs[d]);
                PyObject* l = PyList_New(len);
                if (!l) throw py::error_already_set();
                for (Py_ssize_t i = 0; i < len; ++i) {
                    PyObject* o = objs_[(size_t)ids[offs[d] + i]];
                    Py_INCREF(o);
                    PyList_SET_ITEM(l, i, o);
                }
                PyList_SET_ITEM(outer.ptr(), (Py_ssize_t)d, l);
            }
            return outer;
        }
        const int threads = pool_threads((size_t)n);
        const bool timing = getenv("TD_LIST_TIMING") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto t_0 = now();
        auto lap = [&](const char* what) { if (timing) { auto t = now(); fprintf(stderr, "[lists] %s %.2f ms\n", what, std::chrono::duration<double, std::milli>(t - t_0).count()); t_0 = t; } };
        std::vector<std::vector<uint32_t>> hist((size_t)threads);
        // ONE pass over the ids: the histogram, sized for every id the tokenizer can produce (max_id_hint; the table of ints is made for
        // exactly those), with the range check in it — an id outside takes the plain path below.  (Round 4 found the bounds in a pass of
        // its own: 16 of 70 ms per 13 M ids.)
        ensure(max_id_hint);
        lap("ensure");
        const size_t hsize = objs_.size();
        std::vector<uint8_t> bad((size_t)threads, 0);
        {
            py::gil_scoped_release rel;
            parallel_ranges((size_t)n, threads, [&](size_t lo, size_t hi, int k) {
                auto& h = hist[(size_t)k];
                h.assign(hsize, 0u);
                uint32_t* hp = h.data();
                uint8_t b = 0;
                for (size_t i = lo; i < hi; ++i) {
                    const uint32_t v = (uint32_t)ids[i];
                    if (v < hsize) ++hp[v]; else b = 1;
                }
                bad[(size_t)k] = b;
            });
        }
        for (int k = 0; k < threads; ++k)
            if (bad[(size_t)k]) {
                for (int64_t i = 0; i < n; ++i) if (ids[i] < 0) throw std::runtime_error("negative token id");
                return plain_lists(ids, offs, n_docs);
            }
        const int64_t max_seen = (int64_t)hsize - 1;
        lap("histogram");
        // The lists first: a failed allocation then leaves nothing to roll back.  They are taken out of the cycle collector's
        // sight until they are filled — their items are NULL while the GIL is released below, and gc.get_objects() / get_referrers()
        // of another thread must not be handed a half-made list.
        // Round 5: their item arrays are NOT zeroed first.  PyList_New(len) callocs; for lists of a few hundred KB (2560 slices of
        // 256 MiB) glibc serves that from recycled heap memory and calloc clears it — 440 MB of memset under the GIL for arrays that
        // are overwritten from end to end a moment later.  An empty list gets an uninitialised array from the same allocator list_dealloc
        // frees it with; until it is filled its size stays 0, so a list that is dropped half-way (an allocation fails) frees its
        // array and touches no item.
        py::list outer((size_t)n_docs);
        PyObject_GC_UnTrack(outer.ptr());  // (the outer list too: it is the way to the half-made inner ones — ADVICE r5; list_dealloc untracks again, which is allowed)
        std::vector<PyObject**> items((size_t)n_docs, nullptr);
        for (int64_t d = 0; d < n_docs; ++d) {
            const Py_ssize_t len = (Py_ssize_t)(offs[d + 1] - offs[d]);
            PyObject* l = PyList_New(0);
            if (!l) throw py::error_already_set();  // (outer owns the ones made so far: empty lists with an array of their own)
            PyObject_GC_UnTrack(l);
            PyList_SET_ITEM(outer.ptr(), (Py_ssize_t)d, l);
            if (len > 0) {
                PyObject** arr = (PyObject**)PyMem_Malloc((size_t)len * sizeof(PyObject*));
                if (!arr) { PyErr_NoMemory(); throw py::error_already_set(); }
                hint_huge(arr, (size_t)len * sizeof(PyObject*));
                ((PyListObject*)l)->ob_item = arr;
                ((PyListObject*)l)->allocated = len;
                items[(size_t)d] = arr;
            }
        }
        lap("PyList_New");
        // the references the lists are about to hold, added per distinct id
        for (int64_t id = 0; id <= max_seen; ++id) {
            uint64_t c = 0;
            for (int k = 0; k < threads; ++k)
                if (!hist[(size_t)k].empty()) c += hist[(size_t)k][(size_t)id];
            if (c) Py_SET_REFCNT(objs_[(size_t)id], Py_REFCNT(objs_[(size_t)id]) + (Py_ssize_t)c);
        }
        lap("refcounts");
        {
            py::gil_scoped_release rel;
            PyObject* const* objs = objs_.data();
            parallel_ranges((size_t)n_docs, pool_threads((size_t)n), [&](size_t lo, size_t hi, int) {
                for (size_t d = lo; d < hi; ++d) {
                    PyObject** it = items[d];
                    const int32_t* p = ids + offs[d];
                    const int64_t len = offs[d + 1] - offs[d];
                    for (int64_t i = 0; i < len; ++i) it[i] = objs[(size_t)p[i]];
                }
            });
        }
        lap("fill");
        for (int64_t d = 0; d < n_docs; ++d) {  // (complete now: the size, then back into the collector's sight)
            PyObject* l = PyList_GET_ITEM(outer.ptr(), (Py_ssize_t)d);
            Py_SET_SIZE(l, (Py_ssize_t)(offs[d + 1] - offs[d]));
            PyObject_GC_Track(l);
        }
        PyObject_GC_Track(outer.ptr());
        lap("track");
        return outer;
    }
    int64_t max_id_hint = 0;  // the highest id the tokenizer can produce (regular and special tokens)
private:
    std::vector<PyObject*> objs_;
};

// list[str] (any sequence of str) -> concatenated UTF-8 + offsets; the bytes are copied with the GIL released
struct PackedTexts {
    std::unique_ptr<uint8_t[]> buf;
    std::vector<int64_t> offs;
    size_t total = 0;
    explicit PackedTexts(const py::sequence& texts) {
        // A PRIVATE tuple of the items: the UTF-8 pointers taken below live in the str objects, and the copy runs with the GIL
        // released — the caller's own list may be changed by another thread meanwhile, this tuple keeps every str alive until the copy is done.
        py::object fast = py::reinterpret_steal<py::object>(PySequence_Tuple(texts.ptr()));
        if (!fast) throw py::error_already_set();
        const size_t n = (size_t)PyTuple_GET_SIZE(fast.ptr());
        PyObject** it = ((PyTupleObject*)fast.ptr())->ob_item;
        std::vector<const char*> ptr(n);
        offs.assign(n + 1, 0);
        for (size_t i = 0; i < n; ++i) {
            Py_ssize_t len = 0;
            if (!PyUnicode_Check(it[i])) throw py::type_error("encode_batch expects a sequence of str");
            const char* p = PyUnicode_AsUTF8AndSize(it[i], &len);  // (cached on the str object, which the tuple keeps alive)
            if (!p) throw py::error_already_set();
            ptr[i] = p;
            offs[i + 1] = offs[i] + (int64_t)len;
        }
        total = (size_t)offs[n];
        buf.reset(new uint8_t[total + 64]);
        hint_huge(buf.get(), total + 64);
        {
            py::gil_scoped_release rel;
            parallel_ranges(n, pool_threads(total), [&](size_t lo, size_t hi, int) {
                for (size_t i = lo; i < hi; ++i) memcpy(buf.get() + offs[i], ptr[i], (size_t)(offs[i + 1] - offs[i]));
            });
        }
    }
};

class CoreBPE {
public:
    CoreBPE(const std::string& pattern, const std::vector<VocabItem>& vocab, const std::vector<VocabItem>& special, int device) {
        std::vector<uint8_t> b, sb;
        std::vector<int64_t> o, so;
        std::vector<int32_t> r, sr;
        pack(vocab, false, b, o, r);
        pack(special, true, sb, so, sr);
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_create(pattern.c_str(), (int64_t)r.size(), b.data(), o.data(), r.data(), (int64_t)sr.size(), sb.data(),
                           so.data(), sr.data(), device, &h_);
        }
        if (rc != TD_OK) throw TiktokenError(td_last_error(nullptr));
        for (const auto& it : special) special_ids_[it.token_string] = it.rank;
        set_id_hint();
    }
    // Vocabulary files read by the C++ loaders (td_vocab_*): no Python-side VocabItem objects at all.  Empty paths
    // are skipped; `pattern` overrides the pattern a tekken file carries.
    struct FromFiles {};
    CoreBPE(FromFiles, const std::string& pattern, const std::string& tiktoken_model, const std::string& hf_config,
            bool specials_mergeable, const std::string& tekken, const std::string& vocab_json, const std::string& special_json,
            int device) {
        td_vocab* v = nullptr;
        if (td_vocab_create(&v) != TD_OK) throw TiktokenError("td_vocab_create failed");
        int rc = TD_OK;
        std::string err;
        {
            py::gil_scoped_release rel;
            if (rc == TD_OK && !tekken.empty()) rc = td_vocab_load_tekken(v, tekken.c_str());
            if (rc == TD_OK && !tiktoken_model.empty()) rc = td_vocab_load_tiktoken(v, tiktoken_model.c_str());
            if (rc == TD_OK && !hf_config.empty()) rc = td_vocab_load_hf_special(v, hf_config.c_str(), specials_mergeable ? 1 : 0);
            if (rc == TD_OK && (!vocab_json.empty() || !special_json.empty()))
                rc = td_vocab_load_json(v, vocab_json.empty() ? nullptr : vocab_json.c_str(),
                                        special_json.empty() ? nullptr : special_json.c_str());
            if (rc == TD_OK && !pattern.empty()) rc = td_vocab_set_pattern(v, pattern.c_str());
            if (rc != TD_OK) err = td_vocab_error(v);
            if (rc == TD_OK) {
                rc = td_create_from_vocab(v, device, &h_);
                if (rc != TD_OK) err = td_last_error(nullptr);
            }
        }
        if (rc == TD_OK) {
            pattern_ = td_vocab_pattern(v);
            const uint8_t* b;
            const int64_t* o;
            const int32_t* r;
            int64_t n;
            td_vocab_arrays(v, 1, &b, &o, &r, &n);
            for (int64_t i = 0; i < n; ++i) special_ids_[std::string((const char*)b + o[i], (size_t)(o[i + 1] - o[i]))] = r[i];
        }
        td_vocab_destroy(v);
        if (rc != TD_OK) throw TiktokenError(err);
        set_id_hint();
    }
    void set_id_hint() {
        int64_t m = td_info(h_, TD_INFO_MAX_ID);
        for (const auto& kv : special_ids_) m = std::max<int64_t>(m, kv.second);
        ints_.max_id_hint = std::max<int64_t>(m, 0);
    }
    std::string pattern() const { return pattern_; }
    std::map<std::string, int32_t> special_map() const { return special_ids_; }

    ~CoreBPE() { td_destroy(h_); }
    CoreBPE(const CoreBPE&) = delete;
    CoreBPE& operator=(const CoreBPE&) = delete;

    [[noreturn]] void fail() const { throw TiktokenError(td_last_error(h_)); }

    // one document, host buffers
    std::vector<int> encode_mode(const std::string& text, int mode) {
        IdBuf out(text.size());
        int64_t offs[2] = {0, (int64_t)text.size()}, toffs[2], n = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_encode_batch(h_, (const uint8_t*)text.data(), offs, 1, mode, out.data(), out.cap, toffs, &n);
        }
        if (rc != TD_OK) fail();
        return std::vector<int>(out.data(), out.data() + n);
    }

    std::pair<std::vector<int>, int> encode(const std::string& text, const std::set<std::string>& allowed) {
        // the allowed set goes down as the strings themselves: two special strings may share an id
        std::vector<uint8_t> ab;
        std::vector<int64_t> ao;
        pack_allowed(allowed, ab, ao);
        IdBuf out(text.size());
        int64_t n = 0;
        int32_t last = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_encode_with_special_strs(h_, (const uint8_t*)text.data(), (int64_t)text.size(), ab.data(), ao.data(),
                                             (int64_t)allowed.size(), out.data(), out.cap, &n, &last);
        }
        if (rc != TD_OK) fail();
        return {std::vector<int>(out.data(), out.data() + n), (int)last};
    }

    std::vector<unsigned char> decode_bytes(const std::vector<int>& tokens) {
        std::vector<int32_t> t(tokens.begin(), tokens.end());
        std::vector<unsigned char> out(t.size() * 8 + 64);
        int64_t nb = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_decode_bytes(h_, t.data(), (int64_t)t.size(), out.data(), (int64_t)out.size(), &nb);
            if (rc == TD_E_CAPACITY && nb > (int64_t)out.size()) {
                out.resize((size_t)nb);
                rc = td_decode_bytes(h_, t.data(), (int64_t)t.size(), out.data(), (int64_t)out.size(), &nb);
            }
        }
        if (rc != TD_OK) fail();
        out.resize((size_t)nb);
        return out;
    }

    std::vector<std::string> special_tokens() const {
        std::vector<std::string> v;
        for (const auto& kv : special_ids_) v.push_back(kv.first);
        return v;
    }

    std::vector<int> encode_with_special_tokens(const std::string& text) {
        std::set<std::string> all;
        for (const auto& kv : special_ids_) all.insert(kv.first);
        return encode(text, all).first;
    }

    // ---- bulk surface (not in the reference) --------------------------------------------------
    // concatenated UTF-8 + int64 offsets in, (int32 ids, int64 offsets) out
    py::tuple encode_batch_numpy(py::array_t<uint8_t, py::array::c_style | py::array::forcecast> text,
                                 py::array_t<int64_t, py::array::c_style | py::array::forcecast> offsets, int mode) {
        const int64_t n_docs = (int64_t)offsets.size() - 1;
        if (n_docs < 0) throw TiktokenError("offsets must have n_docs+1 entries");
        const int64_t nbytes = offsets.size() ? offsets.data()[n_docs] : 0;
        if (nbytes > (int64_t)text.size()) throw TiktokenError("offsets exceed the text buffer");
        py::array_t<int64_t> toffs(n_docs + 1);
        // the ids land in the numpy array itself (no intermediate vector: at corpus scale that copy costs as much as
        // the tokenization); the array is shrunk in place afterwards
        py::array_t<int32_t> toks(nbytes / 3 + 16);
        int64_t n = 0;
        int rc;
        {
            int32_t* outp = toks.mutable_data();
            int64_t cap = (int64_t)toks.size();
            const uint8_t* tp = text.data();
            const int64_t* op = offsets.data();
            int64_t* top = toffs.mutable_data();
            py::gil_scoped_release rel;
            rc = td_encode_batch(h_, tp, op, n_docs, mode, outp, cap, top, &n);
        }
        if (rc == TD_E_CAPACITY && n > (int64_t)toks.size()) {
            toks = py::array_t<int32_t>(n);
            int32_t* outp = toks.mutable_data();
            const uint8_t* tp = text.data();
            const int64_t* op = offsets.data();
            int64_t* top = toffs.mutable_data();
            py::gil_scoped_release rel;
            rc = td_encode_batch(h_, tp, op, n_docs, mode, outp, n, top, &n);
        }
        if (rc != TD_OK) fail();
        if (n != (int64_t)toks.size()) toks.resize({(py::ssize_t)n}, false);
        return py::make_tuple(toks, toffs);
    }

    // list[str] in, list[list[int]] out through ONE device batch (PackedTexts / IntCache above)
    py::list encode_batch(const py::sequence& texts, int mode) {
        PackedTexts in(texts);
        const int64_t n_docs = (int64_t)in.offs.size() - 1;
        IdBuf out(in.total);
        std::vector<int64_t> toffs((size_t)n_docs + 1);
        int64_t n = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_encode_batch(h_, in.buf.get(), in.offs.data(), n_docs, mode, out.data(), out.cap, toffs.data(), &n);
        }
        if (rc != TD_OK) fail();
        return ints_.lists(out.data(), toffs.data(), n_docs);
    }
    py::bytes decode_to_bytes(py::array_t<int32_t, py::array::c_style | py::array::forcecast> tokens) {
        std::string out((size_t)tokens.size() * 8 + 64, '\0');
        int64_t nb = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_decode_bytes(h_, tokens.data(), (int64_t)tokens.size(), (uint8_t*)&out[0], (int64_t)out.size(), &nb);
            if (rc == TD_E_CAPACITY && nb > (int64_t)out.size()) {
                out.resize((size_t)nb);
                rc = td_decode_bytes(h_, tokens.data(), (int64_t)tokens.size(), (uint8_t*)&out[0], (int64_t)out.size(), &nb);
            }
        }
        if (rc != TD_OK) fail();
        out.resize((size_t)nb);
        return py::bytes(out);
    }

    // list[str] + allowed special strings -> list[list[int]], all ordinary segments of all texts in ONE device batch
    py::list encode_batch_special(const py::sequence& texts, const std::set<std::string>& allowed) {
        std::vector<uint8_t> ab;
        std::vector<int64_t> ao;
        pack_allowed(allowed, ab, ao);
        PackedTexts in(texts);
        const int64_t n_docs = (int64_t)in.offs.size() - 1;
        IdBuf out(in.total);
        std::vector<int64_t> toffs((size_t)n_docs + 1);
        int64_t n = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_encode_batch_with_special_strs(h_, in.buf.get(), in.offs.data(), n_docs, ab.data(), ao.data(), (int64_t)allowed.size(),
                                                   out.data(), out.cap, toffs.data(), &n);
        }
        if (rc != TD_OK) fail();
        return ints_.lists(out.data(), toffs.data(), n_docs);
    }

    // list[list[int]] in, list[bytes] out through ONE device pass (td_decode_batch)
    std::vector<py::bytes> decode_batch(const std::vector<std::vector<int>>& docs) {
        std::vector<int64_t> offs(1, 0), boffs(docs.size() + 1);
        size_t total = 0;
        for (const auto& d : docs) total += d.size();
        std::vector<int32_t> toks;
        toks.reserve(total + 1);
        for (const auto& d : docs) {
            toks.insert(toks.end(), d.begin(), d.end());
            offs.push_back((int64_t)toks.size());
        }
        std::string out(total * 8 + 64, '\0');
        int64_t nb = 0;
        int rc;
        {
            py::gil_scoped_release rel;
            rc = td_decode_batch(h_, toks.data(), offs.data(), (int64_t)docs.size(), (uint8_t*)&out[0], (int64_t)out.size(), boffs.data(), &nb);
            if (rc == TD_E_CAPACITY && nb > (int64_t)out.size()) {
                out.resize((size_t)nb);
                rc = td_decode_batch(h_, toks.data(), offs.data(), (int64_t)docs.size(), (uint8_t*)&out[0], (int64_t)out.size(), boffs.data(), &nb);
            }
        }
        if (rc != TD_OK) fail();
        std::vector<py::bytes> res;
        res.reserve(docs.size());
        for (size_t d = 0; d < docs.size(); ++d) res.emplace_back(out.data() + boffs[d], (size_t)(boffs[d + 1] - boffs[d]));
        return res;
    }

    py::object token_bytes(int id) const {  // None if the id is not in the vocabulary
        const uint8_t* p = nullptr;
        int64_t n = 0;
        if (td_token_bytes(h_, id, &p, &n) != TD_OK) return py::none();
        return py::bytes((const char*)p, (size_t)n);
    }
    py::object single_token(const std::string& bytes) const {  // None if the bytes are not one token
        int32_t id = 0;
        if (bytes.empty() || td_single_token(h_, (const uint8_t*)bytes.data(), (int64_t)bytes.size(), &id) != TD_OK) return py::none();
        return py::int_(id);
    }
    int64_t info(int what) const { return td_info(h_, what); }
    uintptr_t handle() const { return (uintptr_t)h_; }

private:
    td_tokenizer* h_ = nullptr;
    IntCache ints_;
    std::map<std::string, int32_t> special_ids_;
    std::string pattern_;
};

}  // namespace

PYBIND11_MODULE(_tokendagger_core, m) {
    m.doc() = "tokendagger_amd low-level bindings: the TokenDagger CoreBPE surface over the MI355X HIP library";
    // (test hook, no device needed: the list builder of encode_batch alone — ids cut at offsets -> list[list[int]]; returns
    // the lists and the cache's int objects so that a test can look at their reference counts)
    m.def("_ids_to_lists", [](py::array_t<int32_t, py::array::c_style | py::array::forcecast> ids,
                              py::array_t<int64_t, py::array::c_style | py::array::forcecast> offsets) {
        const int64_t n_docs = (int64_t)offsets.size() - 1;
        if (n_docs < 0 || offsets.data()[0] != 0 || offsets.data()[n_docs] > (int64_t)ids.size()) throw TiktokenError("bad offsets");
        for (int64_t d = 0; d < n_docs; ++d)
            if (offsets.data()[d + 1] < offsets.data()[d]) throw TiktokenError("bad offsets");
        IntCache cache;  // (per call here; a CoreBPE keeps its own for its lifetime)
        for (py::ssize_t i = 0; i < ids.size(); ++i) cache.max_id_hint = std::max<int64_t>(cache.max_id_hint, ids.data()[i]);  // (a CoreBPE knows its highest id)
        return cache.lists(ids.data(), offsets.data(), n_docs);
    }, py::arg("ids"), py::arg("offsets"));

    py::class_<VocabItem>(m, "VocabItem")
        .def(py::init<>())
        .def_readwrite("rank", &VocabItem::rank)
        .def_readwrite("token_bytes", &VocabItem::token_bytes)
        .def_readwrite("token_string", &VocabItem::token_string);

    py::register_exception<TiktokenError>(m, "TiktokenError");

    py::class_<CoreBPE>(m, "CoreBPE")
        .def(py::init<const std::string&, const std::vector<VocabItem>&, const std::vector<VocabItem>&, int>(),
             py::arg("pattern"), py::arg("vocab"), py::arg("special_vocab"), py::arg("device") = -1)
        .def_static(
            "from_files",
            [](const std::string& pattern, const std::string& tiktoken_model, const std::string& hf_config, bool specials_mergeable,
               const std::string& tekken, const std::string& vocab_json, const std::string& special_json, int device) {
                return new CoreBPE(CoreBPE::FromFiles{}, pattern, tiktoken_model, hf_config, specials_mergeable, tekken, vocab_json,
                                   special_json, device);
            },
            py::arg("pattern") = "", py::arg("tiktoken_model") = "", py::arg("hf_config") = "", py::arg("specials_mergeable") = false,
            py::arg("tekken") = "", py::arg("vocab_json") = "", py::arg("special_json") = "", py::arg("device") = -1,
            py::return_value_policy::take_ownership)
        .def("pattern", &CoreBPE::pattern)
        .def("special_map", &CoreBPE::special_map)
        .def("encode_ordinary", [](CoreBPE& self, const std::string& text) { return self.encode_mode(text, TD_MODE_ORDINARY); },
             py::arg("text"))
        .def("encode", &CoreBPE::encode, py::arg("text"), py::arg("allowed_special"))
        .def("decode_bytes", &CoreBPE::decode_bytes, py::arg("tokens"))
        .def("special_tokens", &CoreBPE::special_tokens)
        .def("encode_with_special_tokens", &CoreBPE::encode_with_special_tokens, py::arg("text"))
        .def("encode_batch", &CoreBPE::encode_batch, py::arg("texts"), py::arg("mode") = TD_MODE_ENCODE)
        .def("encode_batch_numpy", &CoreBPE::encode_batch_numpy, py::arg("text"), py::arg("offsets"), py::arg("mode") = TD_MODE_ENCODE)
        .def("decode_to_bytes", &CoreBPE::decode_to_bytes, py::arg("tokens"))
        .def("decode_batch", &CoreBPE::decode_batch, py::arg("docs"))
        .def("encode_batch_special", &CoreBPE::encode_batch_special, py::arg("texts"), py::arg("allowed_special"))
        .def("token_bytes", &CoreBPE::token_bytes, py::arg("id"))
        .def("single_token", [](const CoreBPE& self, py::bytes b) { return self.single_token(std::string(b)); }, py::arg("token_bytes"))
        .def("info", &CoreBPE::info, py::arg("what"))
        .def("handle", &CoreBPE::handle);
}
