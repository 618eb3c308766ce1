// C-ABI host library (include/tokendagger_hip.h) over the gfx950 kernels.
// Host code is C++; it owns the device tables, the per-call workspace and stream-ordered launches.
// There is deliberately NO CPU tokenization path in this file: if HIP is unusable, td_create fails.
#include <hip/hip_runtime.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/tokendagger_hip.h"
#include "td_kernels.h"
#include "td_regex.h"
#include "td_tables.h"
#include "td_vocab.h"

namespace td { hipError_t launch_mid_done(const void* ctl, uint32_t ctl_bytes, void* h_ctl, unsigned long long* h_seq, unsigned long long seq, hipStream_t stream); }  // td_special.hip
namespace td { hipError_t launch_pipe_copy_out(const void* src, void* dst, int64_t n_words, int blocks, hipStream_t stream); }  // td_special.hip
namespace td { hipError_t launch_pipe_publish(const void* ctl, uint32_t ctl_bytes, void* h_ctl, const int64_t* d_toff, int64_t n_off, int64_t* h_toff, hipStream_t stream); }  // td_special.hip
using namespace td;

namespace {

thread_local std::string g_create_err;
// td_last_error(t) must not hand out a pointer into t->err, which another thread's call may be rewriting: every failing
// call copies its message (under the handle's lock) into this thread's slot, and td_last_error reads the slot.
thread_local std::string g_thread_err;
thread_local const td_tokenizer* g_thread_err_owner = nullptr;

// Entry points switch to the handle's device for their own duration only: the caller's current HIP device (torch's,
// in a multi-GPU process) is what it was when the call returns, on every path, td_destroy from a finaliser included.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define HIP_TRY(t, expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            (t)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                     \
            return TD_E_HIP;                                                                  \
        }                                                                                     \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

}  // namespace

namespace {
// Host threads that fill and drain the pinned bounce buffers of the td_encode_batch pipeline: a copy job is cut into
// segments that the workers take from one queue; copies into the pipeline and out of it run side by side.
class CopyPool {
public:
    struct Job { std::atomic<int> remaining{0}; };
    explicit CopyPool(int n) {
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& w : workers_) w.join();
    }
    int size() const { return (int)workers_.size(); }
    std::shared_ptr<Job> copy(void* dst, const void* src, size_t bytes) {
        auto job = std::make_shared<Job>();
        if (bytes == 0) return job;
        const size_t seg = std::max<size_t>(1u << 20, ((bytes / (size_t)(2 * std::max(size(), 1))) + 4095) & ~(size_t)4095);
        int nseg = 0;
        for (size_t lo = 0; lo < bytes; lo += seg) ++nseg;
        job->remaining.store(nseg);
        {
            std::lock_guard<std::mutex> g(mu_);
            for (size_t lo = 0; lo < bytes; lo += seg) q_.push_back({(char*)dst + lo, (const char*)src + lo, std::min(seg, bytes - lo), job});
        }
        cv_.notify_all();
        return job;
    }
    void wait(const std::shared_ptr<Job>& job) {  // (the caller helps)
        while (job->remaining.load() > 0) {
            Seg sg;
            {
                std::unique_lock<std::mutex> g(mu_);
                if (q_.empty()) { done_.wait_for(g, std::chrono::microseconds(50)); continue; }
                sg = q_.front(); q_.pop_front();
            }
            memcpy(sg.dst, sg.src, sg.len);
            if (sg.job->remaining.fetch_sub(1) == 1) done_.notify_all();
        }
    }
private:
    struct Seg { char* dst; const char* src; size_t len; std::shared_ptr<Job> job; };
    void run() {
        for (;;) {
            Seg sg;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                sg = q_.front(); q_.pop_front();
            }
            memcpy(sg.dst, sg.src, sg.len);
            if (sg.job->remaining.fetch_sub(1) == 1) done_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    std::deque<Seg> q_;
    bool stop_ = false;
};

struct Ctl {  // small control block in device memory
    int err;
    int pad;
    long long err_pos;
    uint32_t long_count;   // --- from here on: reset before every call
    uint32_t slow_count;
    unsigned long long pool_used;
    uint32_t scan_done;
    uint32_t merge_next;   // td_merge_pieces: next tile nobody has taken yet
    uint32_t miss_count[6];  // entries on the miss lists (K_MISS_CLASSES of them)
    uint32_t flagged_count;  // tiles flagged TILE_HAS_MISS (on flagged_list: td_merge_pieces draws them from there)
    uint32_t gap_count;      // generic split patterns: stretches of text the pattern skipped
    uint32_t deferred_count; // fused tile loop: token tiles left to td_probe_tiles
    uint32_t giant_count;    // long pieces above 1 KiB
    uint32_t tile_draw;      // fused tile loop: tiles handed out beyond every workgroup's first two
    uint32_t direct_tiles;   // (statistics) pre-tokenizer tiles whose ids the fused loop wrote straight to the output
    uint32_t lb_timeouts;    // ... and tiles it staged because their base was not known in time (behind direct_tiles)
    uint32_t ovf_count;      // td_collect_misses: tiles with a length class that found its lists full
    uint32_t dd_stats[2];    // (statistics) repeats, pieces listed for the merge (td_copy_dups)
    uint32_t gp_ctl[4];      // td_giant_pieces over all workgroups: barrier arrivals, pieces listed, a barrier gave up, pieces above the limit
    uint32_t far_tiles;      // pre-tokenizer tiles without a synchronisation point in their left halo (td_split_far_tiles)
    uint32_t ph_bar;         // td_far_probe / td_tail: arrivals at their grid barriers
    uint32_t gs_done;        // td_giant_scan: workgroups that have left the giant pieces
    uint32_t lp_next;        // td_long_pieces: chunks of the long-piece list drawn so far
};
static_assert(K_MISS_CLASSES <= 6, "Ctl::miss_count");
constexpr size_t CTL_BYTES = 256;  // the control block's place in its buffer; behind it: td_giant_pieces' scratch (TD_GP_SCRATCH_BYTES)
static_assert(sizeof(Ctl) <= CTL_BYTES, "Ctl");
}  // namespace

// What td_create builds and no call changes afterwards: the host tables and their copies in HBM (~30 MB for a 200 000-entry
// vocabulary).  Shared by the handles td_clone makes from one another; freed with the last of them.
struct SharedTables {
    HostTables H;
    std::vector<void*> table_allocs;
    int device = 0;
    ~SharedTables() {
        if (table_allocs.empty()) return;
        DeviceGuard dg(device);
        for (void* p : table_allocs) (void)hipFree(p);
    }
};

struct td_tokenizer {
    std::shared_ptr<SharedTables> shared;
    HostTables& H;                           // = shared->H
    Tables dT;  // device pointers
    const Tables* dTp = nullptr;  // the same descriptor, in device memory
    int device = 0;
    std::vector<void*>& table_allocs;        // = shared->table_allocs (filled by td_create only)
    explicit td_tokenizer(std::shared_ptr<SharedTables> s = std::make_shared<SharedTables>())
        : shared(std::move(s)), H(shared->H), table_allocs(shared->table_allocs) {}
    std::string err;
    std::mutex mu;
    // workspace (grown on demand)
    DevBuf dd_table, rest_mask, coll_ctr, tile_state, slab, docbits, startbits, slow_list, tile_flag, tile_carry, stage, stage2, tile_count, tile_extra, miss_list, flagged_list, deferred_list, gap_list, gapbits, gx_exit, gx_state, tile_base, doc_slot, long_list, pool, ctl, tile_first_doc, chunk_pref;
    DevBuf h2d_text, h2d_offs, d_tokens, d_offsets;  // host-API staging
    DevBuf dec_tokens, dec_off, dec_out;
    int64_t pool_bytes_opt = 0;
    bool profile = false;
    int stop_after = 0;
    // The library never touches the legacy (null) stream on its own: a legacy-stream operation is illegal while ANY thread of
    // the process captures a blocking stream, and synchronises with every blocking stream of every other thread.  Copies the
    // host waits for, table uploads and the host-buffer entry points run on `s_own` (non-blocking, private to the handle);
    // small results come back through `h_ctl` (pinned).
    hipStream_t s_own = nullptr;
    hipStream_t s_aux = nullptr;       // the long pieces beside the short ones (LaunchAux, td_kernels.h); with its two events
    hipEvent_t e_fork = nullptr, e_join = nullptr;
    bool overlap = true;               // TD_OPT_OVERLAP (TD_OVERLAP=0 at td_create time turns it off)
    uint32_t gp_coop_min = 16384;      // TD_OPT_GIANT_COOP_MIN (TD_GP_COOP_MIN at td_create time): pieces above this many bytes get all workgroups of td_giant_pieces
    hipStream_t s_cap = nullptr;       // hipGraph capture only (non-blocking: the CALLER's stream is never put into capture)
    void* h_ctl = nullptr;             // pinned, 256 B: the control block / an 8-byte total on their way to the host
    // the last step as a hipGraph (encode_device_locked): opt-in (TD_OPT_GRAPH, TD_GRAPH=1)
    bool graphs = false;
    int graph_failures = 0;
    hipGraphExec_t graph_exec = nullptr;
    EncodeArgs graph_key, last_key;
    hipStream_t graph_stream = nullptr, last_key_stream = nullptr;
    bool has_last_key = false;
    // allowed special tokens on the device (td_encode_device_with_special): the sorted literal table of the last allowed set
    std::vector<int32_t> sp_key;     // the allowed ids it was built for (sorted, unique)
    DevBuf sp_bytes, sp_off, sp_len, sp_id, sp_parent, sp_first2, sp_hit, sp_acc, sp_cpos, sp_clit, sp_ccount;
    uint32_t sp_n = 0, sp_maxlen = 0;
    bool device_specials = true;     // host-buffer batches of a MiB and more search on the device (TD_OPT_DEVICE_SPECIALS)
    bool sp_active = false;          // this call cuts allowed specials (set around encode_device_locked)
    uint32_t dd_replicas = 4;     // (TD_DD_REPLICAS at td_create time, tuning: a power of two)
    uint32_t dd_minlen = 2;       // (TD_DD_MINLEN at td_create time, tuning: pieces below this many bytes are merged without a look at the table)
    uint32_t dd_entries_opt = 0;  // (TD_DD_ENTRIES=<power of two> at td_create time, tests: seats of the table of distinct missed pieces)
    bool dedupe = true;       // a missed piece whose bytes another one of the call has is merged once (TD_OPT_DEDUPE; TD_DEDUPE=0 at td_create time turns it off)
    bool pack_split = true;   // td_pack_plain + td_pack_rest instead of td_pack_tokens (TD_OPT_PACK_SPLIT; TD_PACK_SPLIT=0 at td_create time turns it off)
    int coll_shrink = 1;      // (TD_COLL_SHRINK=<k> at td_create time, tests: td_collect_misses' lists 1/k of their size)
    // generic patterns with left-context assertions behind special cuts: per document of the NEXT host batch, the bytes at its
    // start that are context only (set around encode_batch_locked by encode_special_locked)
    const uint8_t* gx_prefix_host = nullptr;
    DevBuf gx_prefix;
    const uint8_t* gx_prefix_dev = nullptr;
    bool direct = false; // the fused loop places a tile's ids itself when their output base is known in time (TD_OPT_DIRECT; TD_DIRECT=0 turns it off)
    bool fused = true;  // pre-tokenizer and lookup in one pass over the text (TD_OPT_FUSED; TD_FUSED=0 in the environment turns it off)
    struct Ev3 { hipEvent_t e[TD_PROF_EVENTS]; };
    std::vector<Ev3> ev_pending, ev_free;
    int64_t last_repeats = 0, last_listed = 0;
    int64_t last_long = 0, last_far = 0, last_deferred = 0, last_flagged = 0, last_direct = 0, last_timeouts = 0;
    int64_t last_giant = 0;
    bool seen_counters = false;   // td_device_status / a host-buffer entry point has read a call's counters (what the launch sequence is chosen by)
    bool last_sparse = false;     // (TD_INFO_SPARSE)
    int sparse_opt = -1;          // TD_OPT_SPARSE / TD_SPARSE at td_create time: 1 = always the sparse sequence, 0 = never, -1 = by the last counters
    const RxProgram* d_rx = nullptr;      // generic split pattern: the compiled program and its tables in HBM
    const uint16_t* d_rx_s1 = nullptr;
    const uint8_t* d_rx_s2 = nullptr;
    size_t ws_bytes = 0;
    // One workspace per handle: work of this handle may be in flight on one stream at a time.  A call on another
    // stream first waits (on the device, not the host) for the previous call's last kernel.
    hipEvent_t last_done = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    std::vector<void*> graveyard;  // workspace buffers replaced by larger ones; freed at the next synchronisation point
    // host-buffer pipeline (td_encode_batch on large inputs): four slots of pinned bounce buffers + device buffers, three
    // streams (H2D, kernels, D2H)
    struct PipeSlot {
        void* h_text = nullptr; size_t h_text_cap = 0;   // pinned
        void* h_offs = nullptr; size_t h_offs_cap = 0;   // pinned: rebased document offsets in, token offsets out
        void* h_tok = nullptr; size_t h_tok_cap = 0;     // pinned
        DevBuf d_text, d_offs, d_tok, d_toff;
        hipEvent_t ev_h2d = nullptr, ev_k = nullptr, ev_off = nullptr, ev_tok = nullptr;
        Ctl* h_ctl = nullptr;                            // pinned copy of the device control block after the chunk's kernels
    };
    static constexpr int PIPE_SLOTS = 4;
    PipeSlot pipe[PIPE_SLOTS];
    hipStream_t s_h2d = nullptr, s_k = nullptr, s_d2h = nullptr;
    int64_t pipe_chunk_bytes = 64ll << 20;  // (a GiB of English host to host: 16 MiB chunks 33 GB/s, 32 MiB 39, 64 MiB 40.5, profiles/r5_bench/e2e_sweep.txt)
    int pipe_threads = 16;
    std::unique_ptr<CopyPool> pool_threads;
    // one-launch path for inputs of at most 4 KiB: pinned host buffers the kernel reads and writes directly
    void* small_in = nullptr;
    void* small_out = nullptr;
    void* small_dec_in = nullptr;   // td_small_decode: ids in, status + bytes out
    void* small_dec_out = nullptr;
    unsigned long long small_seq = 0;
    // the resident form of the one-launch kernel (td_small_resident): its own stream, the generation of the last launch
    hipStream_t s_res = nullptr;
    unsigned long long res_gen = 0;
    bool small_resident = false;  // (TD_SMALL_RESIDENT=1 at td_create time.  Built for VERDICT r5 item 5a, measured, OFF: 14.1 us against 13.6 for a one-byte call —
                                  // what a small call costs is the body's PCIe round trips and barriers, not the launch; tests/test_gpu_small_resident.py keeps it right)
    unsigned long long small_idle_ticks = 20000;  // (TD_SMALL_IDLE_US at td_create time, default 200 us: 100 MHz ticks the kernel waits for the next request)
    // host batches between the one-launch path and the pipeline (td_encode_batch, 4 KiB .. 4 MiB): pinned in / out buffers
    void* mid_in = nullptr; size_t mid_in_cap = 0;
    void* mid_out = nullptr; size_t mid_out_cap = 0;
    DevBuf mid_dev;               // [offsets | text] on the device
    unsigned long long mid_seq = 0;
    bool mid_enabled = true;      // (TD_MID_PATH=0 at td_create time: the copies-and-synchronise path of rounds 1-5, A/B)
    bool small_enabled = true;
};

namespace {

void drop_graph(td_tokenizer* t) {
    if (t->graph_exec) (void)hipGraphExecDestroy(t->graph_exec);
    t->graph_exec = nullptr;
}

int ensure(td_tokenizer* t, DevBuf& b, size_t bytes) {
    if (b.cap >= bytes && b.p) return TD_OK;
    if (b.p) {  // kernels of an earlier call may still read it: park it until the next synchronisation point
        t->graveyard.push_back(b.p);
        t->ws_bytes -= b.cap;
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes + bytes / 8 + 256;
    HIP_TRY(t, hipMalloc(&b.p, want));
    b.cap = want;
    t->ws_bytes += want;
    return TD_OK;
}

void bury(td_tokenizer* t) {  // caller has synchronised every stream this handle was used on
    for (void* p : t->graveyard) (void)hipFree(p);
    t->graveyard.clear();
}

int own_streams(td_tokenizer* t) {
    if (!t->s_own) HIP_TRY(t, hipStreamCreateWithFlags(&t->s_own, hipStreamNonBlocking));
    if (!t->h_ctl) HIP_TRY(t, hipHostMalloc(&t->h_ctl, 256, hipHostMallocDefault));
    return TD_OK;
}
// A copy (or fill) the host waits for: on the call's stream, then that stream is synchronised.  Never the legacy stream.
int copy_wait(td_tokenizer* t, void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s) {
    if (bytes == 0) return TD_OK;
    HIP_TRY(t, hipMemcpyAsync(dst, src, bytes, kind, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return TD_OK;
}
int zero_wait(td_tokenizer* t, void* dst, size_t bytes, hipStream_t s) {
    HIP_TRY(t, hipMemsetAsync(dst, 0, bytes, s));
    HIP_TRY(t, hipStreamSynchronize(s));
    return TD_OK;
}
// Everything this handle has in flight is done when this returns (its own streams + the last call's event: the caller's
// stream itself may be gone by now, the event is ours).
void drain(td_tokenizer* t) {
    if (t->has_last && t->last_done) (void)hipEventSynchronize(t->last_done);
    for (hipStream_t st : {t->s_own, t->s_h2d, t->s_k, t->s_d2h, t->s_aux}) if (st) (void)hipStreamSynchronize(st);
}

// Runs f() with the handle locked and its device current; a failure's message is published to this thread's slot.
template <class F>
int locked(td_tokenizer* t, F&& f) {
    std::lock_guard<std::mutex> g(t->mu);
    DeviceGuard dg(t->device);
    const int rc = f();
    if (rc != TD_OK) {
        g_thread_err = t->err;
        g_thread_err_owner = t;
    }
    return rc;
}
int fail_unlocked(td_tokenizer* t, int rc, const std::string& msg) {  // argument errors found before taking the lock
    g_thread_err = msg;
    g_thread_err_owner = t;
    return rc;
}

// Stream order across calls: the per-handle workspace and control block are reused by every call, so a call on a
// stream other than the previous call's waits for that call's last kernel (device-side wait, no host stall).
int order_before(td_tokenizer* t, hipStream_t stream) {
    if (t->has_last && stream != t->last_stream) HIP_TRY(t, hipStreamWaitEvent(stream, t->last_done, 0));
    return TD_OK;
}
int order_after(td_tokenizer* t, hipStream_t stream) {
    if (!t->last_done) HIP_TRY(t, hipEventCreateWithFlags(&t->last_done, hipEventDisableTiming));
    HIP_TRY(t, hipEventRecord(t->last_done, stream));
    t->last_stream = stream;
    t->has_last = true;
    return TD_OK;
}

template <class V>
int upload(td_tokenizer* t, const V* src, size_t count, const V** dst) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count * sizeof(V), 16);
    HIP_TRY(t, hipMalloc(&p, bytes + 16));
    t->table_allocs.push_back(p);
    int rc = copy_wait(t, p, src, count * sizeof(V), hipMemcpyHostToDevice, t->s_own);
    if (rc) return rc;
    *dst = (const V*)p;
    return TD_OK;
}


// The miss lists in one buffer: the tile loops' five (room for K_MISS_LISTED_MAX per tile each), then td_collect_misses'
// K_MISS_CLASSES x COLL_SUBS.  A class gets room for the most records n bytes can hold up to 64 Ki, beyond that for a fixed share
// of n: a miss of <= 8 bytes every 24 bytes of text, 9..16 every 48, 17..32 every 96, 33..48 every 128, longer ones every 160 (0.7 bytes
// of list per byte of text — round 4 had 2.1; mixed-script text has a miss every 51 bytes over all classes, the reference's code
// file set one of <= 8 bytes every 33, and since round 5 only the DISTINCT pieces of a call are listed).  What finds no room is merged by the slow scan
// at the end of td_merge_pieces.  dense (encode_ordinary with a vocabulary whose tokens the merge loop does not all reproduce: no
// whole-piece lookup, EVERY piece of two bytes or more is merged): the worst case, 5.8 bytes per byte of text.  TD_COLL_SHRINK=<k>
// (tests): 1/k of that.
struct CollLayout {
    unsigned long long base[K_MISS_CLASSES];
    uint32_t cap[K_MISS_CLASSES];
    unsigned long long dup_base;  // the lists of repeats behind them: COLL_SUBS x dup_cap entries
    uint32_t dup_cap;
    unsigned long long total;
};
static CollLayout coll_layout(const td_tokenizer* t, int64_t n, int64_t n_tiles, bool dense) {
    static const int minlen[K_MISS_CLASSES] = {2, 9, 17, 33, 49}, share[K_MISS_CLASSES] = {24, 48, 96, 128, 160};
    CollLayout L;
    unsigned long long at = (unsigned long long)(n_tiles + 1) * K_MISS_LISTED_MAX * K_MISS_CLASSES;
    for (int c = 0; c < K_MISS_CLASSES; ++c) {
        const int64_t worst = n / minlen[c] + 1;
        int64_t cls = dense ? worst : std::min<int64_t>(worst, std::max<int64_t>(n / share[c] + 1, 65536));
        if (t->coll_shrink > 1) cls = cls / t->coll_shrink + 1;
        const int64_t per_list = std::min<int64_t>((cls + COLL_SUBS - 1) / COLL_SUBS + 64, 0x7FFFFFF0ll);
        L.base[c] = at;
        L.cap[c] = (uint32_t)per_list;
        at += (unsigned long long)per_list * COLL_SUBS;
    }
    // the repeats (round 5; 8 bytes each): room for as many as the lists of the classes hold records — a missed piece is on one or the other
    L.dup_base = at;
    const unsigned long long coll_records = at - (unsigned long long)(n_tiles + 1) * K_MISS_LISTED_MAX * K_MISS_CLASSES;
    L.dup_cap = (uint32_t)std::min<unsigned long long>(coll_records / COLL_SUBS + 64, 0x7FFFFFF0ull);
    at += (unsigned long long)L.dup_cap * COLL_SUBS;
    L.total = at;
    return L;
}

// seats of the table of distinct missed pieces (EncodeArgs::dd_table): a power of two, about one per 128 bytes of text
// ... as many as an entry of the list of repeats can name next to its tile (39 bits for both: 2 Mi seats up to 1 GiB of text)
static uint32_t dd_tile_bits(int64_t n) {
    const int64_t n_tiles = (n + K_TILE - 1) / K_TILE;
    uint32_t b = 1;
    while (b < 39 && (1ll << b) < n_tiles) ++b;
    return b;
}
static uint32_t dd_entries(const td_tokenizer* t, int64_t n) {
    const uint32_t most = 1u << std::min<uint32_t>(39 - dd_tile_bits(n), 21);  // (21 bits: what a TOK_DUPREF slot has for the seat)
    if (t->dd_entries_opt) return std::min(t->dd_entries_opt, most);
    uint32_t e = 4096;
    while (e < (2u << 20) && e < most && (int64_t)e * 128 < n) e <<= 1;
    return std::min(e, most);
}

int reserve_ws(td_tokenizer* t, int64_t n, int64_t n_docs, bool dense = false) {
    const int64_t n_tiles = (n + K_TILE - 1) / K_TILE;
    int rc;
    if ((rc = ensure(t, t->docbits, (size_t)((n + 31) / 32 + 2) * 4))) return rc;
    if ((rc = ensure(t, t->startbits, (size_t)((n + 31) / 32 + 8) * 4))) return rc;
    if ((rc = ensure(t, t->slow_list, (size_t)(n_tiles * 8 + 64) * 8))) return rc;
    if ((rc = ensure(t, t->tile_flag, (size_t)(n_tiles + 2) * 4))) return rc;   // (per pre-tokenizer tile: fewer than n_tiles)
    if ((rc = ensure(t, t->tile_carry, (size_t)(n_tiles + 2) * 8))) return rc;
    if ((rc = ensure(t, t->tile_state, (size_t)(n_tiles + 2) * 8))) return rc;   // (per pre-tokenizer tile too)
    if (t->direct && t->H.pattern_kind != PATTERN_GENERIC && (rc = ensure(t, t->slab, (size_t)direct_grid_blocks() * SLAB_RING * SLAB_WORDS * 4))) return rc;
    if ((rc = ensure(t, t->stage, (size_t)std::max<int64_t>(n_tiles, 1) * K_STAGE * 4))) return rc;
    if ((rc = ensure(t, t->stage2, (size_t)std::max<int64_t>(n_tiles, 1) * K_STAGE * 4))) return rc;  // ids of merged pieces
    if ((rc = ensure(t, t->tile_count, (size_t)(n_tiles + 1) * 4))) return rc;
    if ((rc = ensure(t, t->tile_extra, (size_t)(n_tiles + 1) * 4))) return rc;
    if ((rc = ensure(t, t->rest_mask, (size_t)((n_tiles + 15) / 16 + 4) * 2))) return rc;
    if ((rc = ensure(t, t->flagged_list, (size_t)(n_tiles + 64) * 4))) return rc;
    if ((rc = ensure(t, t->deferred_list, (size_t)(n_tiles + 64) * 4))) return rc;
    if (t->H.pattern_kind == PATTERN_GENERIC) {
        if ((rc = ensure(t, t->gap_list, (size_t)(n / gx_chunk_for(n) + 64) * 4))) return rc;   // chunks that failed the check
        if ((rc = ensure(t, t->gapbits, (size_t)((n + 31) / 32 + 8) * 4))) return rc;
        if ((rc = ensure(t, t->gx_exit, (size_t)(n / gx_chunk_for(n) + 2) * 8))) return rc;
        if ((rc = ensure(t, t->gx_state, (size_t)(n / gx_chunk_for(n) + 2) * 4))) return rc;
    }
    {
        CollLayout L = coll_layout(t, n, n_tiles, dense);
        if ((rc = ensure(t, t->miss_list, (size_t)L.total * 8))) return rc;
        if ((rc = ensure(t, t->coll_ctr, (size_t)(K_MISS_CLASSES + 1) * COLL_SUBS * COLL_STRIDE * 4))) return rc;
    }
    if ((rc = ensure(t, t->dd_table, (size_t)dd_entries(t, n) * 8))) return rc;
    if ((rc = ensure(t, t->tile_base, (size_t)(n_tiles + 2) * 8))) return rc;
    if ((rc = ensure(t, t->doc_slot, (size_t)(n_docs + 1) * 4))) return rc;
    if ((rc = ensure(t, t->tile_first_doc, (size_t)(n_tiles + 1) * 4))) return rc;
    if ((rc = ensure(t, t->chunk_pref, (size_t)(n_tiles / 4096 + 2) * 8))) return rc;
    if ((rc = ensure(t, t->long_list, (size_t)(n / (K_MAXSHORT + 1) + n_tiles + 16) * sizeof(LongEntry)))) return rc;
    const int64_t pool_bytes = t->pool_bytes_opt > 0 ? t->pool_bytes_opt : std::max<int64_t>(64ll << 20, 2 * n);
    if ((rc = ensure(t, t->pool, (size_t)pool_bytes))) return rc;
    if ((rc = ensure(t, t->ctl, CTL_BYTES + TD_GP_SCRATCH_BYTES))) return rc;
    return TD_OK;
}

int encode_device_locked(td_tokenizer* t, const void* d_text, int64_t n, const void* d_offs, int64_t n_docs, int mode,
                         void* d_out, int64_t out_cap, void* d_out_offs, hipStream_t stream) {
    if (n < 0 || n_docs < 0 || (n > 0 && (!d_text || !d_offs)) || !d_out_offs || (mode != TD_MODE_ENCODE && mode != TD_MODE_ORDINARY)) {
        t->err = "td_encode_device: bad argument";
        return TD_E_INVALID;
    }
    int rc;
    if ((rc = order_before(t, stream))) return rc;
    if (n == 0) {  // nothing but empty documents
        HIP_TRY(t, hipMemsetAsync(d_out_offs, 0, (size_t)(n_docs + 1) * 8, stream));
        return order_after(t, stream);
    }
    const bool dense = mode != TD_MODE_ENCODE && !t->H.merge_closed;
    if ((rc = reserve_ws(t, n, n_docs, dense))) return rc;
    const int64_t n_tiles = (n + K_TILE - 1) / K_TILE;
    if (n_tiles > 0x7FFFFFF0ll) { t->err = "input too large"; return TD_E_INVALID; }
    EncodeArgs a;
    memset(&a, 0, sizeof a);
    a.Tp = t->dTp;
    a.text = (const uint8_t*)d_text;
    a.n = n;
    a.doc_offsets = (const int64_t*)d_offs;
    a.n_docs = n_docs;
    a.docbits = (uint32_t*)t->docbits.p;
    a.startbits = (uint32_t*)t->startbits.p;
    a.slow_list = (int64_t*)t->slow_list.p;
    a.tile_flag = (uint32_t*)t->tile_flag.p;
    a.tile_carry = (int64_t*)t->tile_carry.p;
    a.slow_cap = (uint32_t)std::min<size_t>(t->slow_list.cap / 8, 0x7FFFFFF0u);
    a.stage = (uint32_t*)t->stage.p;
    a.merge_out = (uint32_t*)t->stage2.p;
    a.tile_count = (uint32_t*)t->tile_count.p;
    a.tile_extra = (uint32_t*)t->tile_extra.p;
    a.rest_mask = (uint16_t*)t->rest_mask.p;
    a.tile_base = (int64_t*)t->tile_base.p;
    a.doc_slot = (uint32_t*)t->doc_slot.p;
    a.tile_first_doc = (uint32_t*)t->tile_first_doc.p;
    a.long_list = (LongEntry*)t->long_list.p;
    a.long_cap = (uint32_t)std::min<size_t>(t->long_list.cap / sizeof(LongEntry), 0x7FFFFFF0u);
    Ctl* ctl = (Ctl*)t->ctl.p;
    a.long_count = &ctl->long_count;
    a.giant_count = &ctl->giant_count;
    a.gp_ctl = ctl->gp_ctl;
    a.gp_scratch = (uint32_t*)((char*)t->ctl.p + CTL_BYTES);  // (behind the control block; needs no reset)
    a.gp_coop_min = t->gp_coop_min;
    a.tile_draw = &ctl->tile_draw;
    a.far_count = &ctl->far_tiles;
    a.ph_bar = &ctl->ph_bar;
    a.gs_done = &ctl->gs_done;
    a.lp_next = &ctl->lp_next;
    int pack_dense_opt = 1;
    {
        static const int far_light = getenv("TD_FAR_LIGHT") ? atoi(getenv("TD_FAR_LIGHT")) : 1;
        a.far_light = far_light;
        static const int pack_dense = getenv("TD_PACK_DENSE") ? atoi(getenv("TD_PACK_DENSE")) : 1;
        pack_dense_opt = pack_dense;
    }
    a.slow_count = &ctl->slow_count;
    a.pool = (uint32_t*)t->pool.p;
    a.pool_cap = t->pool.cap / 4;
    a.pool_used = &ctl->pool_used;
    a.scan_done = &ctl->scan_done;
    a.merge_next = &ctl->merge_next;
    a.miss_list = (unsigned long long*)t->miss_list.p;
    a.miss_count = ctl->miss_count;
    {
        const CollLayout L = coll_layout(t, n, n_tiles, dense);
        for (int c = 0; c < K_MISS_CLASSES; ++c) { a.coll_base[c] = L.base[c]; a.coll_cap[c] = L.cap[c]; }
        a.dup_list = a.miss_list + L.dup_base;
        a.dup_cap = L.dup_cap;
        a.coll_count = (uint32_t*)t->coll_ctr.p;
        a.ovf_count = &ctl->ovf_count;
    }
    a.dedupe = (t->dedupe && n_tiles < (1ll << 24)) ? 1 : 0;  // (a table entry keeps the tile in 24 bits)
    a.dd_seat_bits = std::min<uint32_t>(39 - dd_tile_bits(n), 21);
    a.dd_minlen = t->dd_minlen;
    a.dd_replicas = t->dd_replicas;
    a.dd_stats = ctl->dd_stats;
    a.dd_table = a.dedupe ? (unsigned long long*)t->dd_table.p : nullptr;
    a.dd_mask = a.dedupe ? dd_entries(t, n) - 1u : 0u;
    a.flagged_count = &ctl->flagged_count;
    a.rx = t->d_rx;
    a.rx_stage1 = t->d_rx_s1;
    a.rx_stage2 = t->d_rx_s2;
    a.gap_list = (int64_t*)t->gap_list.p;
    a.gap_cap = (uint32_t)std::min<size_t>(t->gap_list.cap / 4, 0x7FFFFFF0u);
    a.gapbits = (uint32_t*)t->gapbits.p;
    a.gx_chunk = gx_chunk_for(n);
    a.gx_exit = (int64_t*)t->gx_exit.p;
    a.gx_state = (uint32_t*)t->gx_state.p;
    a.gap_count = &ctl->gap_count;
    a.gx_prefix = t->gx_prefix_dev;
    a.flagged_list = (uint32_t*)t->flagged_list.p;
    a.deferred_list = (uint32_t*)t->deferred_list.p;
    a.deferred_count = &ctl->deferred_count;
    a.fused = t->fused ? 1 : 0;
    a.pack_split = t->pack_split ? 1 : 0;
    a.direct = (t->direct && t->fused && !t->sp_active && t->H.pattern_kind != PATTERN_GENERIC && d_out && !t->stop_after) ? 1 : 0;
    a.tile_state = (unsigned long long*)t->tile_state.p;
    a.slab = (uint32_t*)t->slab.p;
    a.direct_tiles = &ctl->direct_tiles;
    a.probe_deferred = 0;
    a.miss_cap = (uint32_t)((n_tiles + 1) * K_MISS_LISTED_MAX);
    a.chunk_pref = (int64_t*)t->chunk_pref.p;
    a.ctl_reset = &ctl->long_count;  // keep a sticky error (err / err_pos) but reset the per-call counters
    a.ctl_reset_words = (uint32_t)((sizeof(Ctl) - offsetof(Ctl, long_count)) / 4);
    a.out_tokens = (int32_t*)d_out;
    a.out_cap = d_out ? out_cap : 0;
    a.out_offsets = (int64_t*)d_out_offs;
    a.err = &ctl->err;
    a.err_pos = &ctl->err_pos;
    a.n_tiles = (int)n_tiles;
    a.n_stiles = (int)((n + KS_TILE - 1) / KS_TILE);
    a.pat_flags = pattern_flags(t->H.pattern_kind);
    // encode_ordinary has no whole-piece fast path (tiktoken.cpp:156-167); when every token is
    // reproduced by the merge loop the fast path cannot change the result and stays on.
    a.use_fastpath = (mode == TD_MODE_ENCODE) || t->H.merge_closed;
    a.text_aligned = (((uintptr_t)d_text) & 15) == 0;
    a.stop_after = t->stop_after;
    if (t->sp_active) {
        a.sp.bytes = (const uint8_t*)t->sp_bytes.p;
        a.sp.off = (const uint32_t*)t->sp_off.p;
        a.sp.len = (const uint32_t*)t->sp_len.p;
        a.sp.id = (const int32_t*)t->sp_id.p;
        a.sp.parent = (const int32_t*)t->sp_parent.p;
        a.sp.first2 = (const uint32_t*)t->sp_first2.p;
        a.sp.hitbits = (uint32_t*)t->sp_hit.p;
        a.sp.cand_pos = (int64_t*)t->sp_cpos.p;
        a.sp.cand_lit = (int32_t*)t->sp_clit.p;
        a.sp.cand_count = (uint32_t*)t->sp_ccount.p;
        a.sp.cand_cap = (uint32_t)std::min<size_t>(t->sp_clit.cap / 4, 0x7FFFFFF0u);
        a.sp.accbits = (uint32_t*)t->sp_acc.p;
        a.sp.n = t->sp_n;
        a.sp.maxlen = t->sp_maxlen;
    }
    td_tokenizer::Ev3 ev{};
    if (t->profile) {
        if (!t->ev_free.empty()) { ev = t->ev_free.back(); t->ev_free.pop_back(); }
        else for (auto& e : ev.e) HIP_TRY(t, hipEventCreate(&e));
        t->ev_pending.push_back(ev);
    }
    // A call that repeats the previous one exactly (same buffers, sizes, stream: a loop over a resident batch, one rank's step
    // of a multi-GPU job) can replay the step as ONE hipGraph launch instead of its dozen kernel launches (opt-in: TD_OPT_GRAPH):
    // the second such call captures the launches, the following ones replay them.  The capture runs on a PRIVATE non-blocking
    // stream in thread-local mode — nothing executes during a capture, so it needs no ordering with the caller's stream — and
    // the caller's stream only ever sees hipGraphLaunch: no stream of the application is in capture because of this library,
    // other threads' legacy-stream work (torch's default stream) stays legal, and a failed capture costs nothing but itself:
    // it is ended, the error is cleared and the step is launched kernel by kernel.
    // The launch sequence (launch_encode): SPARSE — six launches, everything between the tile loop and the packing in td_tail +
    // td_giant_scan — when the last call whose counters were read had next to nothing for the kernels that plain text leaves idle (no
    // deferred or far tiles, few flagged tiles, few long pieces); DENSE — those kernels on their own, at their own occupancies —
    // otherwise and for a handle whose counters nobody has read yet.  Either sequence is correct for any text: a wrong guess costs time.
    {
        bool sparse = t->seen_counters && t->last_far == 0 && t->last_deferred == 0 && t->last_giant == 0 &&
                      t->last_flagged * 64 <= n_tiles && t->last_long * 16 <= n_tiles;
        if (t->sparse_opt >= 0) sparse = t->sparse_opt != 0;
        a.sparse = (sparse && a.fused && !a.direct && !t->sp_active && t->H.pattern_kind != PATTERN_GENERIC && !t->stop_after) ? 1 : 0;
        t->last_sparse = a.sparse != 0;
        a.pack_dense = (!a.sparse && a.pack_split && pack_dense_opt) ? 1 : 0;  // (the sparse sequence keeps its six launches: td_pack_rest takes the odd tile with many markers)
    }
    LaunchAux aux_v{nullptr, nullptr, nullptr};
    const LaunchAux* aux = nullptr;
    // ... when the handle has SEEN long pieces: the fork and the join cost a step without any ~13 us (English: -2 % at 256 MiB), a step
    // with a hundred thousand of them gains 4-7 %.  What the last call whose counters were read had (td_device_status, every host-buffer
    // entry point) decides; a handle that never reads them stays in line.
    if (t->overlap && t->last_long >= 2048 && !a.sparse) {
        hipError_t ae = hipSuccess;
        if (!t->s_aux) ae = hipStreamCreateWithFlags(&t->s_aux, hipStreamNonBlocking);  // (a stream priority, either way, changes nothing: r6 call 17)
        if (ae == hipSuccess && !t->e_fork) ae = hipEventCreateWithFlags(&t->e_fork, hipEventDisableTiming);
        if (ae == hipSuccess && !t->e_join) ae = hipEventCreateWithFlags(&t->e_join, hipEventDisableTiming);
        if (ae == hipSuccess) { aux_v = LaunchAux{t->s_aux, t->e_fork, t->e_join}; aux = &aux_v; }
        else (void)hipGetLastError();  // (no second stream: everything in line)
    }
    a.overlap = aux ? 1 : 0;
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cap_status) != hipSuccess) { cap_status = hipStreamCaptureStatusNone; (void)hipGetLastError(); }
    // (a caller that is capturing this stream into a graph of its own gets the plain launches captured there)
    if (t->graphs && !t->profile && !t->stop_after && cap_status == hipStreamCaptureStatusNone) {
        const bool same_as_graph = t->graph_exec && t->graph_stream == stream && memcmp(&a, &t->graph_key, sizeof a) == 0;
        if (same_as_graph) {
            HIP_TRY(t, hipGraphLaunch(t->graph_exec, stream));
            return order_after(t, stream);
        }
        const bool repeats = t->has_last_key && t->last_key_stream == stream && memcmp(&a, &t->last_key, sizeof a) == 0;
        if (repeats) {
            drop_graph(t);
            hipGraph_t g = nullptr;
            hipError_t ce = t->s_cap ? hipSuccess : hipStreamCreateWithFlags(&t->s_cap, hipStreamNonBlocking);
            if (ce == hipSuccess) ce = hipStreamBeginCapture(t->s_cap, hipStreamCaptureModeThreadLocal);
            if (ce == hipSuccess) {
                const hipError_t le = launch_encode(a, t->s_cap, nullptr, aux);
                ce = hipStreamEndCapture(t->s_cap, &g);  // (always: also ends a capture that was invalidated)
                if (le != hipSuccess) ce = le;
            }
            if (ce == hipSuccess && g && hipGraphInstantiate(&t->graph_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                (void)hipGraphDestroy(g);
                t->graph_key = a;
                t->graph_stream = stream;
                t->graph_failures = 0;
                HIP_TRY(t, hipGraphLaunch(t->graph_exec, stream));
                return order_after(t, stream);
            }
            if (g) (void)hipGraphDestroy(g);
            t->graph_exec = nullptr;
            (void)hipGetLastError();
            // not this time: plain launches below; two more identical calls try again, three failures in a row give up
            if (++t->graph_failures >= 3) t->graphs = false;
            t->has_last_key = false;
        } else {
            t->last_key = a;
            t->last_key_stream = stream;
            t->has_last_key = true;
        }
    }
    HIP_TRY(t, launch_encode(a, stream, t->profile ? ev.e : nullptr, aux));
    return order_after(t, stream);
}

int absorb_ctl(td_tokenizer* t, Ctl c, hipStream_t stream, int64_t* err_pos);
int device_status_locked(td_tokenizer* t, hipStream_t stream, int64_t* err_pos) {
    if (!t->ctl.p) { HIP_TRY(t, hipStreamSynchronize(stream)); return TD_OK; }
    int rc0 = own_streams(t);
    if (rc0) return rc0;
    static_assert(sizeof(Ctl) <= 256, "h_ctl");
    HIP_TRY(t, hipMemcpyAsync(t->h_ctl, t->ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, stream));  // (behind the call's kernels)
    HIP_TRY(t, hipStreamSynchronize(stream));
    if (!t->has_last || t->last_stream == stream) bury(t);  // nothing of this handle is in flight any more
    return absorb_ctl(t, *(const Ctl*)t->h_ctl, stream, err_pos);
}

// the control block of a finished call, on the host: counters the launch sequence / second stream are chosen by, the device's error (if any)
int absorb_ctl(td_tokenizer* t, Ctl c, hipStream_t stream, int64_t* err_pos) {
    int rc0;
    if (c.err == TD_E_BAD_TOKEN) c.err_pos = 0x7FFFFFFFFFFFFFFFll - c.err_pos;  // (td_decode_len keeps the LOWEST invalid index as a maximum)
    t->last_long = c.long_count;
    t->last_far = (int64_t)c.slow_count + c.far_tiles;
    t->last_giant = c.giant_count;
    t->seen_counters = true;
    t->last_deferred = c.deferred_count;
    t->last_flagged = c.flagged_count;
    t->last_direct = c.direct_tiles;
    t->last_timeouts = c.lb_timeouts;
    t->last_repeats = c.dd_stats[0];
    t->last_listed = c.dd_stats[1];
    if (err_pos) *err_pos = c.err_pos;
    if (c.err != 0) {
        if ((rc0 = zero_wait(t, t->ctl.p, sizeof(Ctl), stream))) return rc0;
        switch (c.err) {
            case TD_E_UNKNOWN_BYTE:
                t->err = "No value found for piece at byte offset " + std::to_string(c.err_pos) + ": byte sequence is not in the vocabulary";
                break;
            case TD_E_CAPACITY:
                t->err = "output capacity too small: " + std::to_string(c.err_pos) + " tokens needed";
                break;
            case TD_E_BAD_TOKEN:
                t->err = "Invalid token for decoding at index " + std::to_string(c.err_pos);
                break;
            case TD_E_SCRATCH:
                t->err = "device scratch exhausted near byte offset " + std::to_string(c.err_pos) +
                         " (pieces above 64 bytes: raise TD_OPT_LONG_POOL_BYTES; allowed special tokens: more candidates than one per 32 bytes of the batch)";
                break;
            default:
                t->err = "device error " + std::to_string(c.err) + " at byte offset " + std::to_string(c.err_pos);
        }
        return c.err;
    }
    return TD_OK;
}

}  // namespace

extern "C" {

int td_create(const char* pat_str, int64_t n_vocab, const uint8_t* token_bytes, const int64_t* token_offsets,
              const int32_t* ranks, int64_t n_special, const uint8_t* special_bytes, const int64_t* special_offsets,
              const int32_t* special_ids, int device, td_tokenizer** out) {
    if (!out) return TD_E_INVALID;
    *out = nullptr;
    if (!pat_str || !token_bytes || !token_offsets || !ranks || n_vocab <= 0 || n_special < 0) {
        g_create_err = "td_create: bad argument";
        return TD_E_INVALID;
    }
    td_tokenizer* t = new td_tokenizer;
    if (const char* e = getenv("TD_FUSED")) t->fused = atoi(e) != 0;
    if (const char* e = getenv("TD_GRAPH")) t->graphs = atoi(e) != 0;
    if (const char* e = getenv("TD_DIRECT")) t->direct = atoi(e) != 0;
    if (const char* e = getenv("TD_PACK_SPLIT")) t->pack_split = atoi(e) != 0;
    if (const char* e = getenv("TD_DEDUPE")) t->dedupe = atoi(e) != 0;
    if (const char* e = getenv("TD_OVERLAP")) t->overlap = atoi(e) != 0;
    if (const char* e = getenv("TD_MID_PATH")) t->mid_enabled = atoi(e) != 0;
    if (const char* e = getenv("TD_SMALL_RESIDENT")) t->small_resident = atoi(e) != 0;
    if (const char* e = getenv("TD_SMALL_IDLE_US")) { const long v = atol(e); if (v >= 1 && v <= 1000000) t->small_idle_ticks = (unsigned long long)v * 100ull; }
    if (const char* e = getenv("TD_SPARSE")) { const int v = atoi(e); if (v >= -1 && v <= 1) t->sparse_opt = v; }
    if (const char* e = getenv("TD_GP_COOP_MIN")) { if (atol(e) >= 1024) t->gp_coop_min = (uint32_t)std::min<long>(atol(e), 0x7FFFFFFF); }
    if (const char* e = getenv("TD_DD_REPLICAS")) { const int v = atoi(e); if (v >= 1 && v <= 16 && !(v & (v - 1))) t->dd_replicas = (uint32_t)v; }
    if (const char* e = getenv("TD_DD_MINLEN")) t->dd_minlen = (uint32_t)std::max(2, atoi(e));
    if (const char* e = getenv("TD_DD_ENTRIES")) { const long v = atol(e); if (v >= 2 && v <= (1l << 24) && !(v & (v - 1))) t->dd_entries_opt = (uint32_t)v; }
    if (const char* e = getenv("TD_COLL_SHRINK")) t->coll_shrink = std::max(1, atoi(e));
    std::string err;
    int rc = build_tables(pat_str, n_vocab, token_bytes, token_offsets, ranks, n_special, special_bytes, special_offsets,
                          special_ids, t->H, err);
    if (rc != TD_OK) {
        g_create_err = err;
        delete t;
        return rc;
    }
    auto fail = [&](int code) {
        g_create_err = t->err;
        td_destroy(t);
        return code;
    };
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev <= 0) {
        t->err = std::string("no usable HIP device (") + hipGetErrorString(he) + "); this library has no CPU path";
        return fail(TD_E_HIP);
    }
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) { t->err = "device index out of range"; return fail(TD_E_INVALID); }
    t->device = device;
    t->shared->device = device;
    DeviceGuard dg(device);  // uploads go to the handle's device; the caller's current device is restored on return
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur != device) { t->err = "hipSetDevice failed"; return fail(TD_E_HIP); }
    }
    if ((rc = own_streams(t))) return fail(rc);
    const HostTables& H = t->H;
    const Tables hv = H.view();
    Tables d;
    memset(&d, 0, sizeof d);
    d.piece_mask = H.piece_mask;
    d.pair_mask = H.pair_mask;
    d.max_id = H.max_id;
    d.pseudo_base = H.pseudo_base;
    d.max_token_len = H.max_token_len;
    d.piece12_mask = H.piece12_mask;
    d.pat_flags = hv.pat_flags;
    if ((rc = upload(t, H.ascii_cls.data(), H.ascii_cls.size(), &d.ascii_cls))) return fail(rc);
    if ((rc = upload(t, hv.ucls1, (size_t)4352, &d.ucls1))) return fail(rc);
    {
        // size of stage 2 = (max block index + 1) * 256
        unsigned maxb = 0;
        for (int i = 0; i < 4352; ++i) maxb = std::max<unsigned>(maxb, hv.ucls1[i]);
        if ((rc = upload(t, hv.ucls2, (size_t)(maxb + 1) * 256, &d.ucls2))) return fail(rc);
    }
    if ((rc = upload(t, H.byte_id.data(), H.byte_id.size(), &d.byte_id))) return fail(rc);
    if ((rc = upload(t, H.byte_pair.data(), H.byte_pair.size(), &d.byte_pair))) return fail(rc);
    if ((rc = upload(t, H.byte_pair_id.data(), H.byte_pair_id.size(), &d.byte_pair_id))) return fail(rc);
    if ((rc = upload(t, H.piece_slots.data(), H.piece_slots.size(), &d.piece_slots))) return fail(rc);
    if ((rc = upload(t, H.piece12_slots.data(), H.piece12_slots.size(), &d.piece12_slots))) return fail(rc);
    if ((rc = upload(t, H.pair_slots.data(), H.pair_slots.size(), &d.pair_slots))) return fail(rc);
    if ((rc = upload(t, H.tok_off.data(), H.tok_off.size(), &d.tok_off))) return fail(rc);
    if ((rc = upload(t, H.tok_bytes.data(), H.tok_bytes.size(), &d.tok_bytes))) return fail(rc);
    d.cseed = nullptr; d.cseed_pm = nullptr; d.cseed_nm = nullptr;
    if (!H.cseed.empty()) {  // characters that enter the merge loop whole (td_common.h: character seeds)
        if ((rc = upload(t, H.cseed.data(), H.cseed.size(), &d.cseed))) return fail(rc);
        if ((rc = upload(t, H.cseed_pm.data(), H.cseed_pm.size(), &d.cseed_pm))) return fail(rc);
        if ((rc = upload(t, H.cseed_nm.data(), H.cseed_nm.size(), &d.cseed_nm))) return fail(rc);
    }
    if (H.pattern_kind == PATTERN_GENERIC) {
        size_t n1 = 0, n2 = 0;
        const uint16_t* s1 = rx_stage1(&n1);
        const uint8_t* s2 = rx_stage2(&n2);
        if ((rc = upload(t, reinterpret_cast<const RxProgram*>(H.rx_program.data()), 1, &t->d_rx))) return fail(rc);
        if ((rc = upload(t, s1, n1, &t->d_rx_s1))) return fail(rc);
        if ((rc = upload(t, s2, n2, &t->d_rx_s2))) return fail(rc);
    }
    t->dT = d;
    if ((rc = upload(t, &t->dT, 1, &t->dTp))) return fail(rc);
    if ((rc = ensure(t, t->ctl, CTL_BYTES + TD_GP_SCRATCH_BYTES))) return fail(rc);
    if ((rc = zero_wait(t, t->ctl.p, sizeof(Ctl), t->s_own))) return fail(rc);
    *out = t;
    return TD_OK;
}

int td_clone(td_tokenizer* src, td_tokenizer** out) {
    if (!out) return TD_E_INVALID;
    *out = nullptr;
    if (!src) { g_create_err = "td_clone: bad argument"; return TD_E_INVALID; }
    td_tokenizer* t = nullptr;
    {
        std::lock_guard<std::mutex> g(src->mu);  // (its options are read)
        t = new td_tokenizer(src->shared);
        t->dT = src->dT; t->dTp = src->dTp; t->device = src->device;
        t->d_rx = src->d_rx; t->d_rx_s1 = src->d_rx_s1; t->d_rx_s2 = src->d_rx_s2;
        t->pool_bytes_opt = src->pool_bytes_opt; t->graphs = src->graphs; t->fused = src->fused; t->direct = src->direct; t->pack_split = src->pack_split; t->dedupe = src->dedupe; t->overlap = src->overlap; t->sparse_opt = src->sparse_opt; t->gp_coop_min = src->gp_coop_min; t->dd_entries_opt = src->dd_entries_opt; t->dd_minlen = src->dd_minlen; t->dd_replicas = src->dd_replicas; t->coll_shrink = src->coll_shrink;
        t->device_specials = src->device_specials; t->small_enabled = src->small_enabled; t->mid_enabled = src->mid_enabled; t->small_resident = src->small_resident; t->small_idle_ticks = src->small_idle_ticks;
        t->pipe_chunk_bytes = src->pipe_chunk_bytes; t->pipe_threads = src->pipe_threads;
    }
    DeviceGuard dg(t->device);
    int rc = own_streams(t);
    if (rc == TD_OK) rc = ensure(t, t->ctl, CTL_BYTES + TD_GP_SCRATCH_BYTES);
    if (rc == TD_OK) rc = zero_wait(t, t->ctl.p, sizeof(Ctl), t->s_own);
    if (rc != TD_OK) {
        g_create_err = t->err;
        td_destroy(t);
        return rc;
    }
    *out = t;
    return TD_OK;
}

void td_destroy(td_tokenizer* t) {
    if (!t) return;
    {
        DeviceGuard dg(t->device);  // reached from finalisers at arbitrary points: the caller's device must survive
        drain(t);  // (not hipDeviceSynchronize: other handles' and the application's streams are none of this handle's business)
        bury(t);
        drop_graph(t);
        for (auto& ev : t->ev_pending) for (auto e : ev.e) (void)hipEventDestroy(e);
        for (auto& ev : t->ev_free) for (auto e : ev.e) (void)hipEventDestroy(e);
        if (t->last_done) (void)hipEventDestroy(t->last_done);
        for (auto& sl : t->pipe) {
            for (void* hp : {sl.h_text, sl.h_offs, sl.h_tok, (void*)sl.h_ctl}) if (hp) (void)hipHostFree(hp);
            for (DevBuf* b : {&sl.d_text, &sl.d_offs, &sl.d_tok, &sl.d_toff}) if (b->p) (void)hipFree(b->p);
            for (hipEvent_t e : {sl.ev_h2d, sl.ev_k, sl.ev_off, sl.ev_tok}) if (e) (void)hipEventDestroy(e);
        }
        for (hipStream_t st : {t->s_h2d, t->s_k, t->s_d2h, t->s_own, t->s_cap, t->s_aux}) if (st) (void)hipStreamDestroy(st);
        for (hipEvent_t e : {t->e_fork, t->e_join}) if (e) (void)hipEventDestroy(e);
        if (t->h_ctl) (void)hipHostFree(t->h_ctl);
        if (t->s_res) { (void)hipStreamSynchronize(t->s_res); (void)hipStreamDestroy(t->s_res); }  // (the resident small-call kernel leaves by itself within its idle time)
        if (t->small_in) (void)hipHostFree(t->small_in);
        if (t->mid_in) (void)hipHostFree(t->mid_in);
        if (t->mid_out) (void)hipHostFree(t->mid_out);
        if (t->small_dec_in) (void)hipHostFree(t->small_dec_in);
        if (t->small_dec_out) (void)hipHostFree(t->small_dec_out);
        if (t->small_out) (void)hipHostFree(t->small_out);
        DevBuf* bufs[] = {&t->dd_table, &t->rest_mask, &t->coll_ctr, &t->gx_prefix, &t->tile_state, &t->slab, &t->docbits, &t->startbits, &t->slow_list, &t->tile_flag, &t->tile_carry, &t->stage, &t->stage2, &t->tile_count, &t->tile_extra, &t->miss_list, &t->flagged_list, &t->deferred_list, &t->gap_list, &t->gapbits, &t->gx_exit, &t->gx_state, &t->sp_bytes, &t->sp_off, &t->sp_len, &t->sp_id, &t->sp_parent, &t->sp_first2, &t->sp_hit, &t->sp_acc, &t->sp_cpos, &t->sp_clit, &t->sp_ccount, &t->tile_base, &t->doc_slot, &t->long_list,
                          &t->pool, &t->ctl, &t->tile_first_doc, &t->chunk_pref, &t->h2d_text, &t->h2d_offs, &t->d_tokens, &t->d_offsets, &t->dec_tokens,
                          &t->dec_off, &t->dec_out};
        for (DevBuf* b : bufs)
            if (b->p) (void)hipFree(b->p);
    }
    if (g_thread_err_owner == t) g_thread_err_owner = nullptr;
    delete t;
}

const char* td_last_error(const td_tokenizer* t) {
    if (!t) return g_create_err.c_str();
    // this thread's copy of the message of ITS last failing call on the handle (never a pointer into storage that
    // another thread's call may be rewriting)
    return g_thread_err_owner == t ? g_thread_err.c_str() : "";
}

int td_reserve(td_tokenizer* t, int64_t max_bytes, int64_t max_docs) {
    if (!t || max_bytes < 0 || max_docs < 0) return TD_E_INVALID;
    // (an encode_ordinary call on a vocabulary the merge loop does not reproduce merges EVERY piece: reserve for that, so that no
    // call after td_reserve allocates — ADVICE r4)
    return locked(t, [&] { return reserve_ws(t, std::max<int64_t>(max_bytes, 1), max_docs, !t->H.merge_closed); });
}

int td_encode_device(td_tokenizer* t, const void* d_text, int64_t n_bytes, const void* d_doc_offsets, int64_t n_docs,
                     int mode, void* d_out_tokens, int64_t out_capacity, void* d_out_offsets, void* hip_stream) {
    if (!t) return TD_E_INVALID;
    return locked(t, [&] {
        return encode_device_locked(t, d_text, n_bytes, d_doc_offsets, n_docs, mode, d_out_tokens, out_capacity,
                                    d_out_offsets, (hipStream_t)hip_stream);
    });
}

// The allowed literals (every special string that carries one of the ids) as td_special.hip wants them: sorted bytewise, each
// with the longest other literal that is a proper prefix of it, and the bitmap of their first two bytes.
static int build_special_table(td_tokenizer* t, const int32_t* allowed_ids, int64_t n_allowed) {
    std::vector<int32_t> key(allowed_ids, allowed_ids + n_allowed);
    std::sort(key.begin(), key.end());
    key.erase(std::unique(key.begin(), key.end()), key.end());
    if (key == t->sp_key && t->sp_n) return TD_OK;
    const HostTables& H = t->H;
    std::vector<std::pair<std::string, int32_t>> lits;
    for (int32_t id : key) {
        bool found = false;
        for (size_t k = 0; k < H.special_ids.size(); ++k)
            if (H.special_ids[k] == id && !H.special_strs[k].empty()) { lits.emplace_back(H.special_strs[k], id); found = true; }
        if (!found) { t->err = "Special token id " + std::to_string(id) + " not found in special encoder"; return TD_E_SPECIAL; }
    }
    std::sort(lits.begin(), lits.end(), [](const auto& x, const auto& y) { return x.first < y.first; });  // (bytewise: std::string compares as unsigned char)
    lits.erase(std::unique(lits.begin(), lits.end(), [](const auto& x, const auto& y) { return x.first == y.first; }), lits.end());
    const size_t n = lits.size();
    std::vector<uint8_t> bytes;
    std::vector<uint32_t> off(n + 1, 0), lens(n + 1, 0), first2(2048, 0);
    std::vector<int32_t> ids(n), parent(n, -1);
    uint32_t maxlen = 0;
    for (size_t i = 0; i < n; ++i) {
        const std::string& x = lits[i].first;
        off[i] = (uint32_t)bytes.size();
        lens[i] = (uint32_t)x.size();
        bytes.insert(bytes.end(), x.begin(), x.end());
        while (bytes.size() % 4) bytes.push_back(0);
        if (x.size() > 48) {
            t->err = "td_encode_device_with_special: the allowed special token '" + x + "' is " + std::to_string(x.size()) +
                     " bytes long; the device search takes literals of at most 48 bytes (td_encode_batch_with_special searches on the host)";
            return TD_E_INVALID;
        }
        ids[i] = lits[i].second;
        maxlen = std::max<uint32_t>(maxlen, (uint32_t)x.size());
        // longest proper prefix that is a literal: in sorted order a prefix stands in front of its extensions
        for (size_t j = i; j-- > 0;) {
            const std::string& y = lits[j].first;
            if (y.size() < x.size() && x.compare(0, y.size(), y) == 0) { parent[i] = (int32_t)j; break; }
            if (y.empty() || (uint8_t)y[0] != (uint8_t)x[0]) break;
        }
        const uint32_t b0 = (uint8_t)x[0];
        if (x.size() == 1) for (uint32_t b1 = 0; b1 < 256; ++b1) first2[(b0 << 8 | b1) >> 5] |= 1u << ((b0 << 8 | b1) & 31);
        else { const uint32_t kk = b0 << 8 | (uint8_t)x[1]; first2[kk >> 5] |= 1u << (kk & 31); }
    }
    if (bytes.empty()) bytes.push_back(0);
    int rc;
    if ((rc = ensure(t, t->sp_bytes, bytes.size() + 16))) return rc;
    if ((rc = ensure(t, t->sp_off, (n + 1) * 4))) return rc;
    if ((rc = ensure(t, t->sp_len, (n + 1) * 4))) return rc;
    if ((rc = ensure(t, t->sp_id, std::max<size_t>(n, 1) * 4))) return rc;
    if ((rc = ensure(t, t->sp_parent, std::max<size_t>(n, 1) * 4))) return rc;
    if ((rc = ensure(t, t->sp_first2, 2048 * 4))) return rc;
    if ((rc = own_streams(t))) return rc;
    // (the callers have waited for the kernels that read the previous table; the call that uses this one is ordered behind
    // these copies by the host: each is waited for)
    if ((rc = copy_wait(t, t->sp_bytes.p, bytes.data(), bytes.size(), hipMemcpyHostToDevice, t->s_own))) return rc;
    if ((rc = copy_wait(t, t->sp_off.p, off.data(), (n + 1) * 4, hipMemcpyHostToDevice, t->s_own))) return rc;
    if ((rc = copy_wait(t, t->sp_len.p, lens.data(), (n + 1) * 4, hipMemcpyHostToDevice, t->s_own))) return rc;
    if ((rc = copy_wait(t, t->sp_id.p, ids.data(), n * 4, hipMemcpyHostToDevice, t->s_own))) return rc;
    if ((rc = copy_wait(t, t->sp_parent.p, parent.data(), n * 4, hipMemcpyHostToDevice, t->s_own))) return rc;
    if ((rc = copy_wait(t, t->sp_first2.p, first2.data(), 2048 * 4, hipMemcpyHostToDevice, t->s_own))) return rc;
    t->sp_key = key;
    t->sp_n = (uint32_t)n;
    t->sp_maxlen = maxlen;
    return TD_OK;
}

int td_encode_device_with_special(td_tokenizer* t, const void* d_text, int64_t n_bytes, const void* d_doc_offsets, int64_t n_docs,
                                  const int32_t* allowed_ids, int64_t n_allowed, void* d_out_tokens, int64_t out_capacity,
                                  void* d_out_offsets, void* hip_stream) {
    if (!t || n_allowed < 0 || (n_allowed > 0 && !allowed_ids)) return TD_E_INVALID;
    return locked(t, [&] {
        if (n_allowed == 0 || n_bytes == 0)
            return encode_device_locked(t, d_text, n_bytes, d_doc_offsets, n_docs, TD_MODE_ENCODE, d_out_tokens, out_capacity, d_out_offsets,
                                        (hipStream_t)hip_stream);
        if (t->H.pattern_kind == PATTERN_GENERIC) {
            t->err = "td_encode_device_with_special: generic split patterns take their subjects from the document offsets; use td_encode_batch_with_special";
            return (int)TD_E_PATTERN;
        }
        int rc;
        // (the table of the previous call may still be read by its kernels: a different allowed set waits for them)
        {
            std::vector<int32_t> key(allowed_ids, allowed_ids + n_allowed);
            std::sort(key.begin(), key.end());
            key.erase(std::unique(key.begin(), key.end()), key.end());
            if (key != t->sp_key && t->has_last) HIP_TRY(t, hipEventSynchronize(t->last_done));
        }
        if ((rc = build_special_table(t, allowed_ids, n_allowed))) return rc;
        if ((rc = ensure(t, t->sp_hit, (size_t)((n_bytes + 31) / 32 + 8) * 4))) return rc;
        if ((rc = ensure(t, t->sp_acc, (size_t)((n_bytes + 31) / 32 + 8) * 4))) return rc;
        if ((rc = ensure(t, t->sp_cpos, (size_t)(n_bytes / 32 + 4096) * 8))) return rc;   // candidates: room for one per 32 bytes
        if ((rc = ensure(t, t->sp_clit, (size_t)(n_bytes / 32 + 4096) * 4))) return rc;
        if ((rc = ensure(t, t->sp_ccount, 64))) return rc;
        t->sp_active = t->sp_n != 0;
        rc = encode_device_locked(t, d_text, n_bytes, d_doc_offsets, n_docs, TD_MODE_ENCODE, d_out_tokens, out_capacity, d_out_offsets,
                                  (hipStream_t)hip_stream);
        t->sp_active = false;
        return rc;
    });
}

int td_device_status(td_tokenizer* t, void* hip_stream, int64_t* err_pos) {
    if (!t) return TD_E_INVALID;
    return locked(t, [&] { return device_status_locked(t, (hipStream_t)hip_stream, err_pos); });
}

}  // extern "C"

namespace {

int check_offsets(td_tokenizer* t, const char* what, const int64_t* offs, int64_t n_docs, const void* payload) {
    if (offs[0] != 0) { t->err = std::string(what) + " must start at 0"; return TD_E_INVALID; }
    for (int64_t d = 0; d < n_docs; ++d)
        if (offs[d + 1] < offs[d]) { t->err = std::string(what) + " must be non-decreasing"; return TD_E_INVALID; }
    if (offs[n_docs] > 0 && !payload) { t->err = "null buffer with non-empty documents"; return TD_E_INVALID; }
    return TD_OK;
}


// ---- td_encode_batch on large inputs: chunks of documents through pinned bounce buffers, H2D || kernels || D2H -------
// A plain hipMemcpy from pageable memory runs at 9-10 GB/s (the runtime stages it through one pinned buffer on one thread)
// and round 1's td_encode_batch did copy in, kernels, copy out one after the other: 29 ms for 256 MiB of which 1.7 ms were
// kernels.  Here several host threads copy a chunk into a pinned buffer while the previous chunk is on the wire, the kernels
// of chunk i run while chunk i + 1 goes down and the ids of chunk i - 1 come up, and the ids are copied out of their pinned
// buffer by the same threads.
void parallel_memcpy(void* dst, const void* src, size_t bytes, int threads) {
    if (bytes < (4u << 20) || threads <= 1) { memcpy(dst, src, bytes); return; }
    std::vector<std::thread> th;
    const size_t part = ((bytes / (size_t)threads) + 4095) & ~(size_t)4095;
    for (int k = 1; k < threads; ++k) {
        const size_t lo = part * (size_t)k;
        if (lo >= bytes) break;
        const size_t len = std::min(part, bytes - lo);
        th.emplace_back([=] { memcpy((char*)dst + lo, (const char*)src + lo, len); });
    }
    memcpy(dst, src, std::min(part, bytes));
    for (auto& x : th) x.join();
}

int pinned_ensure(td_tokenizer* t, void*& p, size_t& cap, size_t bytes) {
    if (cap >= bytes && p) return TD_OK;
    if (p) HIP_TRY(t, hipHostFree(p));
    p = nullptr; cap = 0;
    const size_t want = bytes + bytes / 8 + 4096;
    HIP_TRY(t, hipHostMalloc(&p, want, hipHostMallocDefault));
    cap = want;
    return TD_OK;
}

int pipe_init(td_tokenizer* t) {
    if (t->s_h2d) return TD_OK;
    HIP_TRY(t, hipStreamCreateWithFlags(&t->s_h2d, hipStreamNonBlocking));
    HIP_TRY(t, hipStreamCreateWithFlags(&t->s_k, hipStreamNonBlocking));
    HIP_TRY(t, hipStreamCreateWithFlags(&t->s_d2h, hipStreamNonBlocking));
    for (auto& sl : t->pipe) {
        for (hipEvent_t* e : {&sl.ev_h2d, &sl.ev_k, &sl.ev_off, &sl.ev_tok}) HIP_TRY(t, hipEventCreateWithFlags(e, hipEventDisableTiming));
        HIP_TRY(t, hipHostMalloc((void**)&sl.h_ctl, sizeof(Ctl), hipHostMallocDefault));
    }
    return TD_OK;
}

int encode_batch_pipelined(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                           int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    int rc;
    if ((rc = pipe_init(t))) return rc;
    if (!t->pool_threads || t->pool_threads->size() != std::max(t->pipe_threads - 1, 1)) t->pool_threads.reset(new CopyPool(std::max(t->pipe_threads - 1, 1)));
    if ((rc = order_before(t, t->s_k))) return rc;
    // chunks: whole documents, about pipe_chunk_bytes each (inputs of less than six such chunks: a sixth of the input, down to an
    // eighth of pipe_chunk_bytes — the pipeline needs a few chunks in flight to hide anything)
    const int64_t n_all = doc_offsets[n_docs] - doc_offsets[0];
    const int64_t chunk_bytes = std::min(t->pipe_chunk_bytes, std::max<int64_t>(t->pipe_chunk_bytes / 8, n_all / 6));
    std::vector<int64_t> cd{0};
    for (int64_t d = 0; d < n_docs;) {
        const int64_t lo = doc_offsets[d];
        int64_t e = d + 1;
        // (binary search for the last document that still fits)
        int64_t a = d + 1, b = n_docs;
        while (a < b) { const int64_t mid = (a + b + 1) >> 1; if (doc_offsets[mid] - lo <= chunk_bytes) a = mid; else b = mid - 1; }
        e = std::max(e, a);
        cd.push_back(e);
        d = e;
    }
    const int nchunks = (int)cd.size() - 1;
    t->last_direct = t->last_timeouts = 0;
    struct Pending { int64_t d0, d1, b0, nbytes, ntok; };
    std::vector<Pending> pend((size_t)nchunks);
    int64_t tok_base = 0;
    bool capacity_miss = false;
    int first_err = TD_OK;
    constexpr int NS = td_tokenizer::PIPE_SLOTS;
    const bool timing = getenv("TD_PIPE_TIMING") != nullptr;
    static const bool publish = !(getenv("TD_PIPE_PUBLISH") && atoi(getenv("TD_PIPE_PUBLISH")) == 0);
    // A chunk's ids leave the device by a KERNEL that stores them into the pinned buffer (32 workgroups; TD_PIPE_D2H_KERNEL=0: by
    // hipMemcpyAsync).  As SDMA copies on their own stream they did not run beside the H2D copies of the next chunks on this box — a
    // GiB of English took the SUM of the two directions, 37 ms, whatever the chunk size, the copy threads, a second pair of streams,
    // HSA_ENABLE_SDMA_GANG=0 or the small dependent copies (profiles/r5_bench/e2e_sweep.txt) — although two streams of queued copies
    // alone do overlap (tools/gpu_pcie_duplex.py: 20 ms).  Stores over PCIe from a kernel do: 27 ms.
    static const int d2h_blocks = getenv("TD_PIPE_D2H_KERNEL") ? atoi(getenv("TD_PIPE_D2H_KERNEL")) : 32;
    double tm[6] = {0, 0, 0, 0, 0, 0};  // wait for the text copy | enqueue | wait for a chunk's kernels | wait for an out-copy | wait for its ids | start copies
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto lap = [&](int k, std::chrono::steady_clock::time_point t0) { if (timing) tm[k] += std::chrono::duration<double, std::milli>(now() - t0).count(); };
    std::shared_ptr<CopyPool::Job> in_job[NS], out_job[NS];
    // chunk i's text starts its way into the slot's pinned buffer (the pool copies; nobody waits here).  The slot was chunk
    // i - NS's, whose H2D copy was over before its kernels were, and those were waited for in fetch(i - NS)
    auto start_in = [&](int i) -> int {
        td_tokenizer::PipeSlot& sl = t->pipe[i % NS];
        const int64_t d0 = cd[(size_t)i], d1 = cd[(size_t)i + 1], b0 = doc_offsets[d0], nb = doc_offsets[d1] - b0;
        const auto t0 = now();
        int r;
        if ((r = pinned_ensure(t, sl.h_text, sl.h_text_cap, (size_t)nb + 64))) return r;
        in_job[i % NS] = t->pool_threads->copy(sl.h_text, text + b0, (size_t)nb);
        lap(5, t0);
        return TD_OK;
    };
    auto submit = [&](int i) -> int {
        td_tokenizer::PipeSlot& sl = t->pipe[i % NS];
        const int64_t d0 = cd[(size_t)i], d1 = cd[(size_t)i + 1], b0 = doc_offsets[d0], nb = doc_offsets[d1] - b0, nd = d1 - d0;
        pend[(size_t)i] = {d0, d1, b0, nb, 0};
        int r;
        auto t0 = now();
        if ((r = pinned_ensure(t, sl.h_offs, sl.h_offs_cap, (size_t)(nd + 1) * 8))) return r;
        if ((r = ensure(t, sl.d_text, (size_t)nb + 64))) return r;
        if ((r = ensure(t, sl.d_offs, (size_t)(nd + 1) * 8))) return r;
        if ((r = ensure(t, sl.d_toff, (size_t)(nd + 1) * 8))) return r;
        if ((r = ensure(t, sl.d_tok, (size_t)std::max<int64_t>(nb, 1) * 4))) return r;  // worst case one id per byte
        int64_t* ho = (int64_t*)sl.h_offs;
        for (int64_t k = 0; k <= nd; ++k) ho[k] = doc_offsets[d0 + k] - b0;
        lap(1, t0);
        t0 = now();
        if (in_job[i % NS]) { t->pool_threads->wait(in_job[i % NS]); in_job[i % NS].reset(); }
        lap(0, t0);
        t0 = now();
        if (nb > 0) HIP_TRY(t, hipMemcpyAsync(sl.d_text.p, sl.h_text, (size_t)nb, hipMemcpyHostToDevice, t->s_h2d));
        HIP_TRY(t, hipMemcpyAsync(sl.d_offs.p, sl.h_offs, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, t->s_h2d));
        HIP_TRY(t, hipEventRecord(sl.ev_h2d, t->s_h2d));
        HIP_TRY(t, hipStreamWaitEvent(t->s_k, sl.ev_h2d, 0));
        if ((r = encode_device_locked(t, sl.d_text.p, nb, sl.d_offs.p, nd, mode, sl.d_tok.p, std::max<int64_t>(nb, 1), sl.d_toff.p, t->s_k))) return r;
        // the chunk's error word and the workspace counters travel with its offsets (the next chunk resets the counters)
        if (publish) {
            HIP_TRY(t, launch_pipe_publish(t->ctl.p, (uint32_t)sizeof(Ctl), sl.h_ctl, (const int64_t*)sl.d_toff.p, nd + 1, (int64_t*)sl.h_offs, t->s_k));
        } else {
            HIP_TRY(t, hipMemcpyAsync(sl.h_ctl, t->ctl.p, sizeof(Ctl), hipMemcpyDeviceToHost, t->s_k));
            HIP_TRY(t, hipMemcpyAsync(sl.h_offs, sl.d_toff.p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToHost, t->s_k));
        }
        HIP_TRY(t, hipEventRecord(sl.ev_off, t->s_k));
        lap(1, t0);
        return TD_OK;
    };
    auto fetch = [&](int i) -> int {  // chunk i's kernels are done: its total is known, its ids start their way up
        td_tokenizer::PipeSlot& sl = t->pipe[i % NS];
        Pending& P = pend[(size_t)i];
        auto t0 = now();
        HIP_TRY(t, hipEventSynchronize(sl.ev_off));
        lap(2, t0);
        const int64_t nd = P.d1 - P.d0;
        const int64_t* to = (const int64_t*)sl.h_offs;
        P.ntok = to[nd];
        t->last_direct += sl.h_ctl->direct_tiles;  // (td_info: sums over the call's chunks)
        t->last_timeouts += sl.h_ctl->lb_timeouts;
        if (sl.h_ctl->err != 0 && first_err == TD_OK) {
            first_err = sl.h_ctl->err;
            const long long pos = sl.h_ctl->err_pos + (first_err == TD_E_UNKNOWN_BYTE || first_err == TD_E_SCRATCH ? P.b0 : 0);
            t->err = first_err == TD_E_UNKNOWN_BYTE ? "No value found for piece at byte offset " + std::to_string(pos) + ": byte sequence is not in the vocabulary"
                                                    : "device error " + std::to_string(first_err) + " near byte offset " + std::to_string(pos);
        }
        for (int64_t k = 0; k < nd; ++k) out_offsets[P.d0 + k] = tok_base + to[k];
        if (tok_base + P.ntok > out_capacity) capacity_miss = true;
        int r;
        // the slot's pinned id buffer was chunk i - NS's: its ids have to have left it (the pool's copy, started three rounds ago)
        t0 = now();
        if (out_job[i % NS]) { t->pool_threads->wait(out_job[i % NS]); out_job[i % NS].reset(); }
        lap(3, t0);
        if ((r = pinned_ensure(t, sl.h_tok, sl.h_tok_cap, (size_t)std::max<int64_t>(P.ntok, 1) * 4))) return r;
        if (P.ntok > 0 && !capacity_miss && first_err == TD_OK)
        {
            if (d2h_blocks > 0) HIP_TRY(t, launch_pipe_copy_out(sl.d_tok.p, sl.h_tok, P.ntok, d2h_blocks, t->s_d2h));
            else HIP_TRY(t, hipMemcpyAsync(sl.h_tok, sl.d_tok.p, (size_t)P.ntok * 4, hipMemcpyDeviceToHost, t->s_d2h));
        }
        HIP_TRY(t, hipEventRecord(sl.ev_tok, t->s_d2h));
        const int64_t base = tok_base;
        tok_base += P.ntok;
        P.nbytes = base;  // (reused: where the chunk's ids go in the caller's buffer)
        return TD_OK;
    };
    auto deliver = [&](int i) -> int {  // chunk i's ids are in its pinned buffer: the pool copies them out while the next chunks go in
        td_tokenizer::PipeSlot& sl = t->pipe[i % NS];
        const Pending& P = pend[(size_t)i];
        auto t0 = now();
        HIP_TRY(t, hipEventSynchronize(sl.ev_tok));
        lap(4, t0);
        t0 = now();
        if (P.ntok > 0 && !capacity_miss && first_err == TD_OK) out_job[i % NS] = t->pool_threads->copy(out_tokens + P.nbytes, sl.h_tok, (size_t)P.ntok * 4);
        lap(5, t0);
        return TD_OK;
    };
    // Round 5: the host thread no longer WAITS for a copy it has just started.  The text of chunk i + 1 goes into its pinned buffer
    // while chunk i is enqueued and chunks i - 1, i - 2 are collected, and a slot's ids have three rounds to leave it (four slots):
    // with three slots and the text copied inside submit() a round was out-copy + in-copy back to back on this thread (1.1 ms per
    // 32 MiB chunk: 37 ms per GiB of English = 0.50 of what the two PCIe directions allow side by side).
    if (nchunks > 0) rc = start_in(0);
    for (int i = 0; rc == TD_OK && i < nchunks + 2; ++i) {
        if (i < nchunks && (rc = submit(i))) break;
        if (i + 1 < nchunks && (rc = start_in(i + 1))) break;
        if (i - 1 >= 0 && i - 1 < nchunks && (rc = fetch(i - 1))) break;
        if (i - 2 >= 0 && i - 2 < nchunks && (rc = deliver(i - 2))) break;
    }
    for (auto& j : in_job) if (j) t->pool_threads->wait(j);
    for (auto& j : out_job) if (j) t->pool_threads->wait(j);
    if (timing)
        fprintf(stderr, "[tokendagger] pipeline: %d chunks; host thread ms: wait text copy %.2f | enqueue %.2f | wait kernels %.2f | wait out-copy %.2f | wait ids %.2f | start copies %.2f\n",
                nchunks, tm[0], tm[1], tm[2], tm[3], tm[4], tm[5]);
    if (rc != TD_OK) {
        // a chunk failed on the host side (allocation, a HIP call): nothing of this call may still be reading the caller's
        // text or writing its output buffers when the error is returned — the copy jobs are done (above), the three streams
        // are drained here
        (void)hipStreamSynchronize(t->s_h2d);
        (void)hipStreamSynchronize(t->s_k);
        (void)hipStreamSynchronize(t->s_d2h);
        return rc;
    }
    out_offsets[n_docs] = tok_base;
    if (n_tokens) *n_tokens = tok_base;
    HIP_TRY(t, hipStreamSynchronize(t->s_k));
    if (first_err != TD_OK) {
        if ((rc = zero_wait(t, t->ctl.p, sizeof(Ctl), t->s_k))) return rc;
        return first_err;
    }
    if (capacity_miss) {
        t->err = "output capacity too small: " + std::to_string(tok_base) + " tokens needed";
        return TD_E_CAPACITY;
    }
    if (tok_base > 0 && !out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
    return TD_OK;
}

// ---- td_encode_batch on tiny inputs: ONE launch, no hipMemcpy, no stream synchronisation ------------------------------
constexpr int64_t SMALL_MAX_BYTES = 4096, SMALL_MAX_DOCS = 1024;
static_assert(SMALL_MAX_DOCS == SM_MAXDOCS, "td_small_encode keeps the document offsets in LDS");
constexpr size_t SMALL_IN_BYTES = 64 + (SMALL_MAX_DOCS + 2) * 8 + SMALL_MAX_BYTES + 256;  // (64: td_small_resident's request header)
constexpr size_t SMALL_OUT_BYTES = 64 + (SMALL_MAX_DOCS + 2) * 8 + SMALL_MAX_BYTES * 4 + 256;
// returns TD_OK, a TD_E_* code, or -1: the kernel handed the call back (a piece above 64 bytes)
int encode_batch_small(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                       int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    const int64_t n = doc_offsets[n_docs];
    if (!t->small_in) {
        HIP_TRY(t, hipHostMalloc(&t->small_in, SMALL_IN_BYTES, hipHostMallocDefault));
        HIP_TRY(t, hipHostMalloc(&t->small_out, SMALL_OUT_BYTES, hipHostMallocDefault));
        memset(t->small_out, 0, SMALL_OUT_BYTES);
    }
    int rc;
    if ((rc = own_streams(t))) return rc;
    const size_t offs_bytes = (((size_t)(n_docs + 1) * 8) + 15) & ~(size_t)15;
    uint8_t* in = (uint8_t*)t->small_in;
    memcpy(in + 64, doc_offsets, (size_t)(n_docs + 1) * 8);
    memcpy(in + 64 + offs_bytes, text, (size_t)n);
    uint8_t* out = (uint8_t*)t->small_out;
    SmallArgs a;
    a.Tp = t->dTp;
    a.doc_offsets = (const int64_t*)(in + 64);
    a.text = in + 64 + offs_bytes;
    a.status = (SmallStatus*)out;
    a.out_offsets = (int64_t*)(out + 64);
    a.out_tokens = (int32_t*)(out + 64 + offs_bytes);
    a.seq = ++t->small_seq;
    a.n = (int)n;
    a.n_docs = (int)n_docs;
    a.use_fastpath = (mode == TD_MODE_ENCODE) || t->H.merge_closed;
    volatile unsigned long long* seqp = &a.status->seq;
    hipStream_t s = t->s_own;
    if (t->small_resident) {
        // the request for td_small_resident: header fields, then the sequence number (release); the kernel is launched when the last one
        // has left (its generation stands at out + 40 then) — it reads tables and pinned buffers only, so it needs no ordering with the
        // handle's other work
        if (!t->s_res) HIP_TRY(t, hipStreamCreateWithFlags(&t->s_res, hipStreamNonBlocking));
        SmallMailbox* mb = (SmallMailbox*)in;
        mb->n = a.n; mb->n_docs = a.n_docs; mb->use_fastpath = a.use_fastpath; mb->offs_bytes = (int)offs_bytes;
        __atomic_store_n(&mb->seq, a.seq, __ATOMIC_RELEASE);
        volatile unsigned long long* exitp = (volatile unsigned long long*)(out + 40);
        auto launch = [&]() -> int {
            ++t->res_gen;
            HIP_TRY(t, launch_small_resident(t->dTp, in, out, t->res_gen, t->small_idle_ticks, t->s_res));
            return TD_OK;
        };
        if (t->res_gen == 0 || __atomic_load_n(exitp, __ATOMIC_ACQUIRE) == t->res_gen) { if ((rc = launch())) return rc; }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 0;; ++spins) {
            if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
            if ((spins & 0x3Fu) == 0x3Fu && __atomic_load_n(exitp, __ATOMIC_ACQUIRE) == t->res_gen) {
                // the kernel left (idle time over) without having seen this request: the next generation answers it
                if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
                if ((rc = launch())) return rc;
            }
            if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
                HIP_TRY(t, hipStreamSynchronize(t->s_res));  // (a launch failure surfaces here)
                if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
                t->err = "td_small_resident did not answer";
                return TD_E_HIP;
            }
        }
    } else {
    if ((rc = order_before(t, s))) return rc;
    HIP_TRY(t, launch_small_encode(a, s));
    // the kernel releases its sequence number (system scope) after everything else it wrote: spin on it
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            HIP_TRY(t, hipStreamSynchronize(s));  // (a launch failure surfaces here)
            if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
            t->err = "td_small_encode did not complete";
            return TD_E_HIP;
        }
    }
    }
    const SmallStatus st = *a.status;
    if (st.fallback) return -1;
    if (st.err) {
        t->err = st.err == TD_E_UNKNOWN_BYTE ? "No value found for piece at byte offset " + std::to_string(st.err_pos) + ": byte sequence is not in the vocabulary"
                                             : "device error " + std::to_string(st.err);
        return st.err;
    }
    memcpy(out_offsets, a.out_offsets, (size_t)(n_docs + 1) * 8);
    if (n_tokens) *n_tokens = st.n_tokens;
    if ((int64_t)st.n_tokens > out_capacity) { t->err = "output capacity too small: " + std::to_string(st.n_tokens) + " tokens needed"; return TD_E_CAPACITY; }
    if (st.n_tokens) {
        if (!out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
        memcpy(out_tokens, a.out_tokens, (size_t)st.n_tokens * 4);
    }
    return TD_OK;
}

// Host batches of 4 KiB .. 4 MiB (a document, a file, a chat transcript — the calls /root/reference/tests/code_performance_benchmark.py:338-396
// times one by one).  Rounds 1-5: two pageable H2D copies, the step, then THREE copy-and-synchronise round trips (control block, offsets, ids):
// 64 KB of English took 226 us of which the kernels' work was under 20.  Round 6: text and offsets go through ONE pinned buffer and one
// asynchronous copy; the step's pack kernels write ids and offsets STRAIGHT into pinned host memory (they are its output buffers); a last
// one-workgroup kernel copies the control block there and releases a sequence number (system scope) the host spins on — no
// hipStreamSynchronize, no D2H copy on the way back.
constexpr int64_t MID_MAX_BYTES = 4ll << 20, MID_MAX_DOCS = 1ll << 18;
int encode_batch_mid(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                     int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    const int64_t n = doc_offsets[n_docs];
    int rc;
    const size_t offs_bytes = (((size_t)(n_docs + 1) * 8) + 63) & ~(size_t)63;
    const size_t in_bytes = offs_bytes + (size_t)n + 64;
    const size_t out_bytes = 512 + offs_bytes + (size_t)n * 4 + 64;
    if ((rc = pinned_ensure(t, t->mid_in, t->mid_in_cap, in_bytes))) return rc;
    if (!t->mid_out || t->mid_out_cap < out_bytes) {
        if ((rc = pinned_ensure(t, t->mid_out, t->mid_out_cap, out_bytes))) return rc;
        memset(t->mid_out, 0, 512);
    }
    if ((rc = ensure(t, t->mid_dev, in_bytes))) return rc;
    if ((rc = own_streams(t))) return rc;
    hipStream_t s = t->s_own;
    if ((rc = order_before(t, s))) return rc;
    uint8_t* in = (uint8_t*)t->mid_in;
    memcpy(in, doc_offsets, (size_t)(n_docs + 1) * 8);
    memcpy(in + offs_bytes, text, (size_t)n);
    HIP_TRY(t, hipMemcpyAsync(t->mid_dev.p, in, offs_bytes + (size_t)n, hipMemcpyHostToDevice, s));
    uint8_t* out = (uint8_t*)t->mid_out;  // [0, 256): control block | [256]: sequence number | 512: offsets | ids
    int64_t* h_offs = (int64_t*)(out + 512);
    int32_t* h_tok = (int32_t*)(out + 512 + offs_bytes);
    rc = encode_device_locked(t, (uint8_t*)t->mid_dev.p + offs_bytes, n, t->mid_dev.p, n_docs, mode, h_tok, std::max<int64_t>(n, 1), h_offs, s);
    if (rc) return rc;
    const unsigned long long seq = ++t->mid_seq;
    HIP_TRY(t, launch_mid_done(t->ctl.p, (uint32_t)sizeof(Ctl), out, (unsigned long long*)(out + 256), seq, s));
    volatile unsigned long long* seqp = (volatile unsigned long long*)(out + 256);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == seq) break;
        if ((spins & 0xFFFFu) == 0xFFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            HIP_TRY(t, hipStreamSynchronize(s));  // (a launch failure surfaces here)
            if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == seq) break;
            t->err = "td_encode_batch: the step did not complete";
            return TD_E_HIP;
        }
    }
    if (!t->has_last || t->last_stream == s) bury(t);  // (the sequence number is written behind the step's last kernel: nothing of this handle is in flight)
    if ((rc = absorb_ctl(t, *(const Ctl*)out, s, nullptr))) return rc;
    memcpy(out_offsets, h_offs, (size_t)(n_docs + 1) * 8);
    const int64_t total = out_offsets[n_docs];
    if (n_tokens) *n_tokens = total;
    if (total > out_capacity) {
        t->err = "output capacity too small: " + std::to_string(total) + " tokens needed";
        return TD_E_CAPACITY;
    }
    if (total > 0) {
        if (!out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
        memcpy(out_tokens, h_tok, (size_t)total * 4);
    }
    return TD_OK;
}

int encode_batch_locked(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                        int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    int rc;
    if ((rc = check_offsets(t, "doc_offsets", doc_offsets, n_docs, text))) return rc;
    const int64_t n = doc_offsets[n_docs];
    const bool with_prefix = t->gx_prefix_host != nullptr;  // (generic pattern: the plain path below, whatever the size)
    if (n > 0 && n <= SMALL_MAX_BYTES && n_docs <= SMALL_MAX_DOCS && t->small_enabled && t->H.pattern_kind != PATTERN_GENERIC) {  // (the one-launch kernel knows the family's scanners only)
        rc = encode_batch_small(t, text, doc_offsets, n_docs, mode, out_tokens, out_capacity, out_offsets, n_tokens);
        if (rc != -1) return rc;  // (-1: a piece above 64 bytes; the general path below handles it)
    }
    if (n > 0 && n <= MID_MAX_BYTES && n_docs <= MID_MAX_DOCS && t->mid_enabled && !with_prefix)
        return encode_batch_mid(t, text, doc_offsets, n_docs, mode, out_tokens, out_capacity, out_offsets, n_tokens);
    if (n >= t->pipe_chunk_bytes / 2 && out_tokens && !with_prefix)  // (default: from 32 MiB on)
        return encode_batch_pipelined(t, text, doc_offsets, n_docs, mode, out_tokens, out_capacity, out_offsets, n_tokens);
    if ((rc = ensure(t, t->h2d_text, (size_t)n + 64))) return rc;
    if ((rc = ensure(t, t->h2d_offs, (size_t)(n_docs + 1) * 8))) return rc;
    if ((rc = ensure(t, t->d_offsets, (size_t)(n_docs + 1) * 8))) return rc;
    // worst case one token per byte; typical text needs a quarter of that
    const int64_t dev_cap = std::max<int64_t>(n, 1);
    if ((rc = ensure(t, t->d_tokens, (size_t)dev_cap * 4))) return rc;
    if ((rc = own_streams(t))) return rc;
    hipStream_t s = t->s_own;
    if ((rc = order_before(t, s))) return rc;
    if (n > 0) HIP_TRY(t, hipMemcpyAsync(t->h2d_text.p, text, (size_t)n, hipMemcpyHostToDevice, s));
    HIP_TRY(t, hipMemcpyAsync(t->h2d_offs.p, doc_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s));
    if (with_prefix) {
        if ((rc = ensure(t, t->gx_prefix, (size_t)n_docs + 16))) return rc;
        HIP_TRY(t, hipMemcpyAsync(t->gx_prefix.p, t->gx_prefix_host, (size_t)n_docs, hipMemcpyHostToDevice, s));
        t->gx_prefix_dev = (const uint8_t*)t->gx_prefix.p;
    }
    rc = encode_device_locked(t, t->h2d_text.p, n, t->h2d_offs.p, n_docs, mode, t->d_tokens.p, dev_cap, t->d_offsets.p, s);
    t->gx_prefix_dev = nullptr;
    if (rc) return rc;
    rc = device_status_locked(t, s, nullptr);
    if (rc) return rc;
    if ((rc = copy_wait(t, out_offsets, t->d_offsets.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s))) return rc;
    const int64_t total = out_offsets[n_docs];
    if (n_tokens) *n_tokens = total;
    if (total > out_capacity) {
        t->err = "output capacity too small: " + std::to_string(total) + " tokens needed";
        return TD_E_CAPACITY;
    }
    if (total > 0) {
        if (!out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
        if ((rc = copy_wait(t, out_tokens, t->d_tokens.p, (size_t)total * 4, hipMemcpyDeviceToHost, s))) return rc;
    }
    return TD_OK;
}

int decode_args(td_tokenizer* t, const void* d_tokens, int64_t n_tokens, void* d_out, int64_t out_cap, void* d_n_bytes,
                hipStream_t stream, DecodeArgs& a) {
    int rc;
    const int64_t npref = ((n_tokens / 4096 + 4) + 1) & ~1ll;  // even: the offsets behind it stay 16-byte aligned
    if ((rc = ensure(t, t->dec_off, (size_t)(n_tokens + 4) * 4 + (size_t)npref * 8))) return rc;
    memset(&a, 0, sizeof a);
    a.Tp = t->dTp;
    a.tokens = (const int32_t*)d_tokens;
    a.n = n_tokens;
    a.chunk_pref = (int64_t*)t->dec_off.p;                       // 8-byte aligned part first
    a.local_off = (uint32_t*)(a.chunk_pref + npref);
    a.out = (uint8_t*)d_out;
    a.out_cap = out_cap;
    a.n_bytes = (int64_t*)d_n_bytes;
    Ctl* ctl = (Ctl*)t->ctl.p;
    a.scan_done = &ctl->scan_done;
    a.err = &ctl->err;
    a.err_pos = &ctl->err_pos;
    if ((rc = order_before(t, stream))) return rc;
    HIP_TRY(t, hipMemsetAsync(&ctl->scan_done, 0, 4, stream));
    return TD_OK;
}

}  // namespace

extern "C" {

int td_encode_batch(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs, int mode,
                    int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    if (!t || !doc_offsets || n_docs < 0 || !out_offsets || out_capacity < 0) return TD_E_INVALID;
    if ((mode == TD_MODE_ENCODE || mode == TD_MODE_ORDINARY) && doc_offsets[0] == 0 && doc_offsets[n_docs] == 0) {
        // nothing but empty documents: no ids, and no reason to wake the device (enc.encode("") was 48 us)
        for (int64_t d = 0; d <= n_docs; ++d) out_offsets[d] = 0;
        if (n_tokens) *n_tokens = 0;
        return TD_OK;
    }
    return locked(t, [&] { return encode_batch_locked(t, text, doc_offsets, n_docs, mode, out_tokens, out_capacity, out_offsets, n_tokens); });
}

int td_decode_device(td_tokenizer* t, const void* d_tokens, int64_t n_tokens, void* d_out, int64_t out_capacity, void* d_n_bytes,
                     void* hip_stream) {
    if (!t || n_tokens < 0 || (n_tokens > 0 && (!d_tokens || !d_out)) || out_capacity < 0) return TD_E_INVALID;
    return locked(t, [&] {
        hipStream_t s = (hipStream_t)hip_stream;
        if (n_tokens == 0) {
            if (d_n_bytes) HIP_TRY(t, hipMemsetAsync(d_n_bytes, 0, 8, s));
            return (int)TD_OK;
        }
        DecodeArgs a;
        int rc = decode_args(t, d_tokens, n_tokens, d_out, out_capacity, d_n_bytes, s, a);
        if (rc) return rc;
        HIP_TRY(t, launch_decode(a, s, 3));
        return order_after(t, s);
    });
}

// decode_bytes on at most SMALL_DEC_MAX_TOKENS ids: ONE launch over pinned host buffers (td_small_decode).  Returns TD_OK, a
// TD_E_* code, or -1: more bytes than the kernel's window holds (the general path takes the call).
static int decode_bytes_small(td_tokenizer* t, const int32_t* tokens, int64_t n_tokens, uint8_t* out, int64_t out_capacity, int64_t* n_bytes) {
    if (!t->small_dec_in) {
        HIP_TRY(t, hipHostMalloc(&t->small_dec_in, SMALL_DEC_MAX_TOKENS * 4 + 64, hipHostMallocDefault));
        HIP_TRY(t, hipHostMalloc(&t->small_dec_out, 64 + SMALL_DEC_MAX_BYTES, hipHostMallocDefault));
        memset(t->small_dec_out, 0, 64 + SMALL_DEC_MAX_BYTES);
    }
    int rc;
    if ((rc = own_streams(t))) return rc;
    hipStream_t s = t->s_own;
    if ((rc = order_before(t, s))) return rc;
    memcpy(t->small_dec_in, tokens, (size_t)n_tokens * 4);
    SmallDecArgs a;
    a.Tp = t->dTp;
    a.tokens = (const int32_t*)t->small_dec_in;
    a.status = (SmallStatus*)t->small_dec_out;
    a.out = (uint8_t*)t->small_dec_out + 64;
    a.seq = ++t->small_seq;
    a.n = (int)n_tokens;
    HIP_TRY(t, launch_small_decode(a, s));
    volatile unsigned long long* seqp = &a.status->seq;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; ++spins) {
        if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
        if ((spins & 0xFFFu) == 0xFFFu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(5)) {
            HIP_TRY(t, hipStreamSynchronize(s));  // (a launch failure surfaces here)
            if (__atomic_load_n(seqp, __ATOMIC_ACQUIRE) == a.seq) break;
            t->err = "td_small_decode did not complete";
            return TD_E_HIP;
        }
    }
    const SmallStatus st = *a.status;
    if (st.err == TD_E_BAD_TOKEN) {
        const long long ep = st.err_pos;
        t->err = "Invalid token for decoding: " + std::to_string(ep >= 0 && ep < n_tokens ? tokens[ep] : -1);  // reference: tiktoken.cpp:249
        return TD_E_BAD_TOKEN;
    }
    if (st.err) { t->err = "device error " + std::to_string(st.err); return st.err; }
    if (st.fallback) return -1;
    if (n_bytes) *n_bytes = st.n_tokens;
    if ((int64_t)st.n_tokens > out_capacity) { t->err = "decode capacity too small"; return TD_E_CAPACITY; }
    if (st.n_tokens > 0 && !out) { t->err = "null out"; return TD_E_INVALID; }
    if (st.n_tokens) memcpy(out, a.out, st.n_tokens);
    return TD_OK;
}

int td_decode_bytes(td_tokenizer* t, const int32_t* tokens, int64_t n_tokens, uint8_t* out, int64_t out_capacity,
                    int64_t* n_bytes) {
    if (!t || n_tokens < 0 || (n_tokens > 0 && !tokens)) return TD_E_INVALID;
    if (n_bytes) *n_bytes = 0;
    if (n_tokens == 0) return TD_OK;
    return locked(t, [&] {
        int rc;
        if (n_tokens <= SMALL_DEC_MAX_TOKENS && t->small_enabled) {
            rc = decode_bytes_small(t, tokens, n_tokens, out, out_capacity, n_bytes);
            if (rc != -1) return rc;
        }
        if ((rc = ensure(t, t->dec_tokens, (size_t)n_tokens * 4 + 16))) return rc;
        if ((rc = own_streams(t))) return rc;
        hipStream_t s = t->s_own;
        if ((rc = order_before(t, s))) return rc;
        if ((rc = copy_wait(t, t->dec_tokens.p, tokens, (size_t)n_tokens * 4, hipMemcpyHostToDevice, s))) return rc;
        // lengths and offsets first: the byte total sizes the device buffer of the gather
        DecodeArgs a;
        if ((rc = decode_args(t, t->dec_tokens.p, n_tokens, nullptr, INT64_MAX, nullptr, s, a))) return rc;
        HIP_TRY(t, launch_decode(a, s, 1));
        if ((rc = order_after(t, s))) return rc;
        int64_t err_pos = 0;
        rc = device_status_locked(t, s, &err_pos);
        if (rc == TD_E_BAD_TOKEN && err_pos >= 0 && err_pos < n_tokens)
            t->err = "Invalid token for decoding: " + std::to_string(tokens[err_pos]);  // reference: tiktoken.cpp:249
        if (rc) return rc;
        if ((rc = copy_wait(t, t->h_ctl, a.chunk_pref + (n_tokens + 4095) / 4096, 8, hipMemcpyDeviceToHost, s))) return rc;
        const int64_t total = *(const int64_t*)t->h_ctl;
        if (n_bytes) *n_bytes = total;
        if (total > out_capacity) { t->err = "decode capacity too small"; return (int)TD_E_CAPACITY; }
        if (total > 0 && !out) { t->err = "null out"; return (int)TD_E_INVALID; }
        if ((rc = ensure(t, t->dec_out, (size_t)total + 16))) return rc;
        a.out = (uint8_t*)t->dec_out.p;
        a.out_cap = total;
        HIP_TRY(t, launch_decode(a, s, 2));
        if ((rc = order_after(t, s))) return rc;
        rc = device_status_locked(t, s, nullptr);
        if (rc) return rc;
        if ((rc = copy_wait(t, out, t->dec_out.p, (size_t)total, hipMemcpyDeviceToHost, s))) return rc;
        return (int)TD_OK;
    });
}

int td_decode_batch(td_tokenizer* t, const int32_t* tokens, const int64_t* tok_offsets, int64_t n_docs, uint8_t* out,
                    int64_t out_capacity, int64_t* out_offsets, int64_t* n_bytes) {
    if (!t || !tok_offsets || n_docs < 0 || !out_offsets) return TD_E_INVALID;
    if (n_bytes) *n_bytes = 0;
    return locked(t, [&] {
        int rc;
        if ((rc = check_offsets(t, "tok_offsets", tok_offsets, n_docs, tokens))) return rc;
        const int64_t n_tokens = tok_offsets[n_docs];
        if (n_tokens == 0) {
            for (int64_t d = 0; d <= n_docs; ++d) out_offsets[d] = 0;
            return (int)TD_OK;
        }
        if ((rc = ensure(t, t->dec_tokens, (size_t)n_tokens * 4 + 16))) return rc;
        if ((rc = ensure(t, t->h2d_offs, (size_t)(n_docs + 1) * 8))) return rc;
        if ((rc = ensure(t, t->d_offsets, (size_t)(n_docs + 1) * 8))) return rc;
        if ((rc = own_streams(t))) return rc;
        hipStream_t s = t->s_own;
        if ((rc = order_before(t, s))) return rc;
        HIP_TRY(t, hipMemcpyAsync(t->dec_tokens.p, tokens, (size_t)n_tokens * 4, hipMemcpyHostToDevice, s));
        if ((rc = copy_wait(t, t->h2d_offs.p, tok_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s))) return rc;
        DecodeArgs a;
        if ((rc = decode_args(t, t->dec_tokens.p, n_tokens, nullptr, INT64_MAX, nullptr, s, a))) return rc;
        a.doc_tok_offsets = (const int64_t*)t->h2d_offs.p;
        a.n_docs = n_docs;
        a.doc_byte_offsets = (int64_t*)t->d_offsets.p;
        HIP_TRY(t, launch_decode(a, s, 1));
        if ((rc = order_after(t, s))) return rc;
        int64_t err_pos = 0;
        rc = device_status_locked(t, s, &err_pos);
        if (rc == TD_E_BAD_TOKEN && err_pos >= 0 && err_pos < n_tokens) t->err = "Invalid token for decoding: " + std::to_string(tokens[err_pos]);
        if (rc) return rc;
        if ((rc = copy_wait(t, out_offsets, t->d_offsets.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s))) return rc;
        const int64_t total = out_offsets[n_docs];
        if (n_bytes) *n_bytes = total;
        if (total > out_capacity) { t->err = "decode capacity too small"; return (int)TD_E_CAPACITY; }
        if ((rc = ensure(t, t->dec_out, (size_t)total + 16))) return rc;
        a.out = (uint8_t*)t->dec_out.p;
        a.out_cap = total;
        HIP_TRY(t, launch_decode(a, s, 2));
        if ((rc = order_after(t, s))) return rc;
        rc = device_status_locked(t, s, nullptr);
        if (rc) return rc;
        if (total > 0) {
            if (!out) { t->err = "null out"; return (int)TD_E_INVALID; }
            if ((rc = copy_wait(t, out, t->dec_out.p, (size_t)total, hipMemcpyDeviceToHost, s))) return rc;
        }
        return (int)TD_OK;
    });
}

}  // extern "C" (reopened below)

namespace {
// Allowed special tokens indexed by their first two bytes: one pass over the text finds, at every position, the
// longest allowed special that starts there (tiktoken semantics: cut at the EARLIEST occurrence; longest on ties).
struct SpecialIndex {
    struct Ent { const std::string* s; int32_t id; };
    bool first[256] = {};
    std::vector<Ent> ents;       // sorted by (first byte, second byte or -1, longer first)
    uint32_t lo[257] = {};       // ents[lo[b0] .. lo[b0 + 1]): the literals that start with byte b0
    size_t count = 0;
    static int second(const std::string& x) { return x.size() > 1 ? (uint8_t)x[1] : -1; }
    void add(const std::string* s, int32_t id) {
        if (s->empty()) return;
        ++count;
        ents.push_back({s, id});
    }
    void finish() {  // (cost proportional to the allowed set: nothing for an empty one)
        std::sort(ents.begin(), ents.end(), [](const Ent& x, const Ent& y) {
            const uint8_t a0 = (uint8_t)(*x.s)[0], b0 = (uint8_t)(*y.s)[0];
            if (a0 != b0) return a0 < b0;
            const int a1 = second(*x.s), b1 = second(*y.s);
            if (a1 != b1) return a1 < b1;
            return x.s->size() > y.s->size();
        });
        uint32_t k = 0;
        for (int b = 0; b < 256; ++b) {
            lo[b] = k;
            while (k < ents.size() && (uint8_t)(*ents[k].s)[0] == b) ++k;
            first[b] = k > lo[b];
        }
        lo[256] = k;
    }
    // longest special starting at text[p] (p < hi), or nullptr
    const Ent* match(const uint8_t* text, int64_t p, int64_t hi) const {
        const uint8_t b0 = text[p];
        if (!first[b0]) return nullptr;
        const Ent* single = nullptr;
        const Ent* e = ents.data() + lo[b0];
        const Ent* end = ents.data() + lo[b0 + 1];
        if (e < end && e->s->size() == 1) { single = e; ++e; }  // (second byte -1 sorts first)
        if (p + 1 < hi) {
            const int b1 = text[p + 1];
            // first literal whose second byte is b1 (binary search over the literals of this first byte)
            const Ent* a = e;
            const Ent* z = end;
            while (a < z) { const Ent* m = a + (z - a) / 2; if (second(*m->s) < b1) a = m + 1; else z = m; }
            for (; a < end && second(*a->s) == b1; ++a)
                if (p + (int64_t)a->s->size() <= hi && memcmp(text + p, a->s->data(), a->s->size()) == 0) return a;
        }
        return single;
    }
};

// The allowed set arrives either as special-token STRINGS (exactly those literals are cut out, tiktoken's
// allowed_special) or as ids (every special string that carries one of the ids — two strings may share an id).
int build_special_index(td_tokenizer* t, const uint8_t* allowed_bytes, const int64_t* allowed_offsets, const int32_t* allowed_ids,
                        int64_t n_allowed, SpecialIndex& ix) {
    const HostTables& H = t->H;
    for (int64_t k = 0; k < n_allowed; ++k) {
        bool found = false;
        if (allowed_offsets) {
            const int64_t lo = allowed_offsets[k], hi = allowed_offsets[k + 1];
            if (hi < lo) { t->err = "allowed_offsets must be non-decreasing"; return TD_E_INVALID; }
            const std::string want((const char*)allowed_bytes + lo, (size_t)(hi - lo));
            for (size_t s = 0; s < H.special_strs.size(); ++s)
                if (H.special_strs[s] == want) { ix.add(&H.special_strs[s], H.special_ids[s]); found = true; break; }
            if (!found) { t->err = "Special token '" + want + "' not found in special encoder"; return TD_E_SPECIAL; }  // tiktoken.cpp:178-180
        } else {
            for (size_t s = 0; s < H.special_ids.size(); ++s)
                if (H.special_ids[s] == allowed_ids[k]) { ix.add(&H.special_strs[s], allowed_ids[k]); found = true; }
            if (!found) { t->err = "Special token id " + std::to_string(allowed_ids[k]) + " not found in special encoder"; return TD_E_SPECIAL; }
        }
    }
    ix.finish();
    return TD_OK;
}

struct Segments {  // ordinary text between allowed special tokens, as one batch for the device
    std::vector<int64_t> seg_offs{0};   // into the ORIGINAL text when `compact` is false, else into seg_text
    std::vector<int64_t> seg_src;       // start of each segment in the original text
    std::vector<int32_t> seg_special;   // special id that follows each segment, -1 after a document's last one
    std::vector<int64_t> doc_seg{0};    // first segment of each document
};

// document text[lo, hi) -> (start, end) of its ordinary segments + the special id that follows each
void segment_document(const SpecialIndex& ix, const uint8_t* text, int64_t lo, int64_t hi, std::vector<int64_t>& starts,
                      std::vector<int64_t>& ends, std::vector<int32_t>& seg_special) {
    int64_t start = lo;
    if (ix.count)
        for (int64_t p = lo; p < hi;) {
            // memchr-speed skip to the next byte that can begin an allowed special
            const SpecialIndex::Ent* e = ix.match(text, p, hi);
            if (!e) { ++p; continue; }
            starts.push_back(start); ends.push_back(p); seg_special.push_back(e->id);
            p += (int64_t)e->s->size();
            start = p;
        }
    starts.push_back(start); ends.push_back(hi); seg_special.push_back(-1);
}

// Shared body of the two *_with_special entry points (handle locked by the caller).
int encode_special_locked(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                          const uint8_t* allowed_bytes, const int64_t* allowed_offsets, const int32_t* allowed_ids, int64_t n_allowed,
                          int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens,
                          int64_t* last_seg_lo, int64_t* last_seg_hi) {
    int rc;
    if ((rc = check_offsets(t, "doc_offsets", doc_offsets, n_docs, text))) return rc;
    if (n_allowed == 0) {  // nothing to cut out: the documents are the segments
        if (last_seg_lo) { *last_seg_lo = n_docs ? doc_offsets[n_docs - 1] : 0; *last_seg_hi = doc_offsets[n_docs]; }
        return encode_batch_locked(t, text, doc_offsets, n_docs, TD_MODE_ENCODE, out_tokens, out_capacity, out_offsets, n_tokens);
    }
    SpecialIndex ix;
    if ((rc = build_special_index(t, allowed_bytes, allowed_offsets, allowed_ids, n_allowed, ix))) return rc;
    // Batches of a MiB and more: the search runs on the device (td_special.hip; the same cuts, td_encode_device_with_special)
    // when the allowed set can be named by ids (no other special string shares an allowed one's id) and the caller does not
    // ask for the last segment (the single-string entry points do, for last_piece_token_len).
    if (t->device_specials && !last_seg_lo && doc_offsets[n_docs] >= (1ll << 20) && t->H.pattern_kind != PATTERN_GENERIC && ix.count > 0) {
        std::vector<int32_t> ids;
        bool nameable = true;
        for (const auto& e : ix.ents) {
            ids.push_back(e.id);
            size_t carriers = 0;
            for (size_t k2 = 0; k2 < t->H.special_ids.size(); ++k2) carriers += t->H.special_ids[k2] == e.id && !t->H.special_strs[k2].empty();
            size_t listed = 0;
            for (const auto& e2 : ix.ents) listed += e2.id == e.id;
            if (carriers != listed) { nameable = false; break; }
            if (e.s->size() > 48) { nameable = false; break; }
        }
        if (nameable) {
            const int64_t n = doc_offsets[n_docs];
            if ((rc = ensure(t, t->h2d_text, (size_t)n + 64))) return rc;
            if ((rc = ensure(t, t->h2d_offs, (size_t)(n_docs + 1) * 8))) return rc;
            if ((rc = ensure(t, t->d_offsets, (size_t)(n_docs + 1) * 8))) return rc;
            const int64_t dev_cap = std::max<int64_t>(n, 1);
            if ((rc = ensure(t, t->d_tokens, (size_t)dev_cap * 4))) return rc;
            if ((rc = own_streams(t))) return rc;
            hipStream_t s = t->s_own;
            if ((rc = order_before(t, s))) return rc;
            {
                std::vector<int32_t> key(ids);
                std::sort(key.begin(), key.end());
                key.erase(std::unique(key.begin(), key.end()), key.end());
                if (key != t->sp_key && t->has_last) HIP_TRY(t, hipEventSynchronize(t->last_done));
            }
            if ((rc = build_special_table(t, ids.data(), (int64_t)ids.size()))) return rc;
            if ((rc = ensure(t, t->sp_hit, (size_t)((n + 31) / 32 + 8) * 4))) return rc;
            if ((rc = ensure(t, t->sp_acc, (size_t)((n + 31) / 32 + 8) * 4))) return rc;
            if ((rc = ensure(t, t->sp_cpos, (size_t)(n / 32 + 4096) * 8))) return rc;
            if ((rc = ensure(t, t->sp_clit, (size_t)(n / 32 + 4096) * 4))) return rc;
            if ((rc = ensure(t, t->sp_ccount, 64))) return rc;
            HIP_TRY(t, hipMemcpyAsync(t->h2d_text.p, text, (size_t)n, hipMemcpyHostToDevice, s));
            HIP_TRY(t, hipMemcpyAsync(t->h2d_offs.p, doc_offsets, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice, s));
            t->sp_active = t->sp_n != 0;
            rc = encode_device_locked(t, t->h2d_text.p, n, t->h2d_offs.p, n_docs, TD_MODE_ENCODE, t->d_tokens.p, dev_cap, t->d_offsets.p, s);
            t->sp_active = false;
            if (rc) return rc;
            rc = device_status_locked(t, s, nullptr);
            if (rc == TD_E_SCRATCH) rc = TD_OK + 1000;  // (more candidates than the device list holds: the host search below)
            if (rc == TD_OK) {
                if ((rc = copy_wait(t, out_offsets, t->d_offsets.p, (size_t)(n_docs + 1) * 8, hipMemcpyDeviceToHost, s))) return rc;
                const int64_t total = out_offsets[n_docs];
                if (n_tokens) *n_tokens = total;
                if (total > out_capacity) { t->err = "output capacity too small: " + std::to_string(total) + " tokens needed"; return TD_E_CAPACITY; }
                if (total > 0) {
                    if (!out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
                    if ((rc = copy_wait(t, out_tokens, t->d_tokens.p, (size_t)total * 4, hipMemcpyDeviceToHost, s))) return rc;
                }
                return TD_OK;
            }
            if (rc != TD_OK + 1000) return rc;
        }
    }
    // 1. host: cut every document at the earliest occurrences of allowed special strings (tiktoken semantics; the
    //    reference's own loop, tiktoken.cpp:130-154,187-231, has iterator-invalidation UB).  Documents are independent:
    //    a few host threads take contiguous document ranges.
    std::vector<int64_t> starts, ends, doc_seg{0};
    std::vector<int32_t> seg_special;
    {
        const int64_t n = doc_offsets[n_docs];
        unsigned hw = std::thread::hardware_concurrency();
        int nth = (int)std::min<int64_t>(hw ? std::min(hw, 32u) : 4, std::max<int64_t>(1, n >> 22));  // one thread per 4 MiB, at most 32
        if (nth <= 1 || n_docs < 2 * nth || ix.count == 0) {
            for (int64_t d = 0; d < n_docs; ++d) {
                segment_document(ix, text, doc_offsets[d], doc_offsets[d + 1], starts, ends, seg_special);
                doc_seg.push_back((int64_t)seg_special.size());
            }
        } else {
            struct Part { std::vector<int64_t> starts, ends, per_doc; std::vector<int32_t> sp; };
            std::vector<Part> parts((size_t)nth);
            std::vector<std::thread> th;
            for (int k = 0; k < nth; ++k)
                th.emplace_back([&, k] {
                    Part& P = parts[(size_t)k];
                    const int64_t da = n_docs * k / nth, db = n_docs * (k + 1) / nth;
                    for (int64_t d = da; d < db; ++d) {
                        segment_document(ix, text, doc_offsets[d], doc_offsets[d + 1], P.starts, P.ends, P.sp);
                        P.per_doc.push_back((int64_t)P.sp.size());
                    }
                });
            for (auto& x : th) x.join();
            for (Part& P : parts) {
                const int64_t base = (int64_t)seg_special.size();
                starts.insert(starts.end(), P.starts.begin(), P.starts.end());
                ends.insert(ends.end(), P.ends.begin(), P.ends.end());
                seg_special.insert(seg_special.end(), P.sp.begin(), P.sp.end());
                for (int64_t v : P.per_doc) doc_seg.push_back(base + v);
            }
        }
    }
    const int64_t nseg = (int64_t)seg_special.size();
    if (last_seg_lo && nseg) { *last_seg_lo = starts[(size_t)nseg - 1]; *last_seg_hi = ends[(size_t)nseg - 1]; }
    // 2. device: all ordinary segments of all documents as ONE batch.  No special was cut out: the segments are the
    //    documents and the text goes down as it is; otherwise the segments are packed (the specials drop out).
    int64_t n_special = 0;
    for (int32_t v : seg_special) n_special += v >= 0;
    std::vector<int64_t> toffs((size_t)nseg + 1);
    int64_t ntok = 0;
    if (n_special == 0) {
        rc = encode_batch_locked(t, text, doc_offsets, n_docs, TD_MODE_ENCODE, out_tokens, out_capacity, out_offsets, &ntok);
        if (n_tokens) *n_tokens = ntok;
        return rc;
    }
    // The reference matches every segment with the text in front of it as left context (pcre2_match on text[0, end) from
    // start_offset, tiktoken.cpp:86-93): behind a special token \\A and ^ cannot match, \\b and a one-character look-behind see the
    // special's last character.  For a pattern with such assertions (rx_left_context) every segment that stands behind a
    // special token is sent down WITH that character in front of it, marked as context (gx_prefix): the matcher starts behind
    // it, sees it, and its bytes get no tokens.  (Round 3 refused the cut.)
    const bool ctx = t->H.rx_left_context && t->H.pattern_kind == PATTERN_GENERIC;
    std::vector<uint8_t> seg_text, prefix;
    std::vector<int64_t> seg_offs((size_t)nseg + 1, 0);
    {
        if (ctx) prefix.assign((size_t)nseg, 0);
        int64_t tot = 0;
        for (int64_t d = 0; d < n_docs; ++d)
            for (int64_t k = doc_seg[(size_t)d]; k < doc_seg[(size_t)d + 1]; ++k) {
                const int64_t lo = starts[(size_t)k], hi = ends[(size_t)k];
                if (ctx && k > doc_seg[(size_t)d] && hi > lo) {  // behind a special token of the same document
                    int64_t c = 1;
                    while (c < 4 && lo - c > doc_offsets[d] && (text[lo - c] & 0xC0u) == 0x80u) ++c;
                    prefix[(size_t)k] = (uint8_t)c;
                }
                tot += hi - lo + (ctx ? prefix[(size_t)k] : 0);
                seg_offs[(size_t)k + 1] = tot;
            }
        seg_text.resize((size_t)std::max<int64_t>(tot, 1));
        for (int64_t k = 0; k < nseg; ++k) {
            const int64_t pre = ctx ? prefix[(size_t)k] : 0, lo = starts[(size_t)k] - pre, hi = ends[(size_t)k];
            if (hi > lo) memcpy(seg_text.data() + seg_offs[(size_t)k], text + lo, (size_t)(hi - lo));
        }
    }
    std::vector<int32_t> toks((size_t)std::max<int64_t>(seg_offs[(size_t)nseg], 1));
    if (ctx) t->gx_prefix_host = prefix.data();
    rc = encode_batch_locked(t, seg_text.data(), seg_offs.data(), nseg, TD_MODE_ENCODE, toks.data(), (int64_t)toks.size(), toffs.data(), &ntok);
    t->gx_prefix_host = nullptr;
    if (rc) return rc;
    // 3. stitch: offsets first (they do not need the capacity), then the ids
    const int64_t need = ntok + n_special;
    if (n_tokens) *n_tokens = need;
    int64_t k = 0;
    for (int64_t d = 0; d < n_docs; ++d) {
        out_offsets[d] = k;
        for (int64_t sg = doc_seg[(size_t)d]; sg < doc_seg[(size_t)d + 1]; ++sg) k += toffs[(size_t)sg + 1] - toffs[(size_t)sg] + (seg_special[(size_t)sg] >= 0);
    }
    out_offsets[n_docs] = k;
    if (need > out_capacity) { t->err = "output capacity too small: " + std::to_string(need) + " tokens needed"; return TD_E_CAPACITY; }
    if (need > 0 && !out_tokens) { t->err = "null out_tokens"; return TD_E_INVALID; }
    k = 0;
    for (int64_t sg = 0; sg < nseg; ++sg) {
        const int64_t cnt = toffs[(size_t)sg + 1] - toffs[(size_t)sg];
        if (cnt) memcpy(out_tokens + k, toks.data() + toffs[(size_t)sg], (size_t)cnt * 4);
        k += cnt;
        if (seg_special[(size_t)sg] >= 0) out_tokens[k++] = seg_special[(size_t)sg];
    }
    return TD_OK;
}

// Second element of the reference's return pair (tiktoken.cpp:185,213,218,225): number of ids of the last regex piece
// of the trailing ordinary segment text[s_lo, s_hi), 0 after a special.  Metadata only, derived on the host tables.
int32_t last_piece_token_len_host(td_tokenizer* t, const uint8_t* text, int64_t s_lo, int64_t s_hi) {
    if (s_hi <= s_lo) return 0;
    struct HostAcc {
        using pos_t = int64_t;
        const Tables* T; const uint8_t* p; int64_t lo, hi, lim;
        uint32_t byte(int64_t i) const { return i < hi ? p[i] : 0u; }
        bool doc(int64_t i) const { return i == lo; }
        uint32_t cf(int64_t i) const {
            if (i >= hi) return F_DOC;
            uint32_t v = classify_at(*T, *this, i);
            if (i == lo) v |= F_DOC;
            return v;
        }
    };
    const Tables hv = t->H.view();
    if (t->H.pattern_kind == PATTERN_GENERIC) {
        // the compiled pattern over the whole segment (no provable restart points): its last piece
        // (a segment behind a special token is matched with the special's last character in front of it, like the batch path)
        int64_t pre = 0;
        if (t->H.rx_left_context && s_lo > 0) {
            pre = 1;
            while (pre < 4 && s_lo - pre > 0 && (text[s_lo - pre] & 0xC0u) == 0x80u) ++pre;
        }
        struct SegAcc { const uint8_t* p; uint32_t byte(int64_t i) const { return p[i]; } } S{text + s_lo - pre};
        const RxProgram& P = *reinterpret_cast<const RxProgram*>(t->H.rx_program.data());
        const RxTables RT = rx_host_tables();
        const int64_t n = s_hi - s_lo + pre;
        int64_t ms = pre, me = pre;
        for (int64_t pos = pre; pos < n; pos = me) rx_next_piece(P, RT, S, pos, n, ms, me);
        const uint32_t len = (uint32_t)(me - ms);
        const uint8_t* pb = text + s_lo - pre + ms;
        std::vector<int32_t> tmp;
        const int32_t whole = (len == 1) ? t->H.byte_id[pb[0]] : piece_lookup(hv, piece_key_host(pb, len), len, [pb](uint32_t i) { return (uint32_t)pb[i]; });
        if (whole != NO_RANK) return 1;
        return merge_piece_host(hv, pb, len, tmp) == TD_OK ? (int32_t)tmp.size() : 0;
    }
    HostAcc A{&hv, text, s_lo, s_hi, s_hi + 4};
    // the last piece starts at or behind the last provable sync point of the segment: walk back to it instead of scanning
    // the whole segment (this runs on the host for every CoreBPE.encode call)
    int64_t p = s_lo;
    for (int64_t q = s_hi - 1; q > s_lo; --q)
        if (is_sync(A.cf(q - 1), A.cf(q), hv.pat_flags)) { p = q; break; }
    int64_t last = p;
    while (p < s_hi) { last = p; p = scan_piece(A, p, hv.pat_flags); }
    std::vector<int32_t> tmp;
    const uint32_t len = (uint32_t)(s_hi - last);
    const uint8_t* pb = text + last;
    const int32_t whole = (len == 1) ? t->H.byte_id[pb[0]]
                                     : piece_lookup(hv, piece_key_host(pb, len), len, [pb](uint32_t i) { return (uint32_t)pb[i]; });
    if (whole != NO_RANK) return 1;
    if (merge_piece_host(hv, pb, len, tmp) == TD_OK) return (int32_t)tmp.size();
    return 0;
}
}  // namespace

extern "C" {

int td_encode_batch_with_special(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                                 const int32_t* allowed_ids, int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity,
                                 int64_t* out_offsets, int64_t* n_tokens) {
    if (!t || !doc_offsets || n_docs < 0 || n_allowed < 0 || (n_allowed > 0 && !allowed_ids) || !out_offsets || out_capacity < 0) return TD_E_INVALID;
    return locked(t, [&] {
        return encode_special_locked(t, text, doc_offsets, n_docs, nullptr, nullptr, allowed_ids, n_allowed, out_tokens, out_capacity,
                                     out_offsets, n_tokens, nullptr, nullptr);
    });
}

int td_encode_batch_with_special_strs(td_tokenizer* t, const uint8_t* text, const int64_t* doc_offsets, int64_t n_docs,
                                      const uint8_t* allowed_bytes, const int64_t* allowed_offsets, int64_t n_allowed,
                                      int32_t* out_tokens, int64_t out_capacity, int64_t* out_offsets, int64_t* n_tokens) {
    if (!t || !doc_offsets || n_docs < 0 || n_allowed < 0 || (n_allowed > 0 && (!allowed_bytes || !allowed_offsets)) || !out_offsets ||
        out_capacity < 0)
        return TD_E_INVALID;
    static const int64_t no_offs[1] = {0};
    return locked(t, [&] {
        return encode_special_locked(t, text, doc_offsets, n_docs, allowed_bytes, n_allowed ? allowed_offsets : no_offs, nullptr, n_allowed,
                                     out_tokens, out_capacity, out_offsets, n_tokens, nullptr, nullptr);
    });
}

int td_encode_with_special(td_tokenizer* t, const uint8_t* text, int64_t n_bytes, const int32_t* allowed_ids,
                           int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity, int64_t* n_tokens,
                           int32_t* last_piece_token_len) {
    if (!t || n_bytes < 0 || (n_bytes > 0 && !text) || n_allowed < 0 || (n_allowed > 0 && !allowed_ids) || out_capacity < 0) return TD_E_INVALID;
    return locked(t, [&] {
        const int64_t doc[2] = {0, n_bytes};
        int64_t offs[2] = {0, 0}, lo = 0, hi = 0;
        const int rc = encode_special_locked(t, text, doc, 1, nullptr, nullptr, allowed_ids, n_allowed, out_tokens, out_capacity, offs,
                                             n_tokens, &lo, &hi);
        if (rc == TD_OK && last_piece_token_len) *last_piece_token_len = last_piece_token_len_host(t, text, lo, hi);
        return rc;
    });
}

int td_encode_with_special_strs(td_tokenizer* t, const uint8_t* text, int64_t n_bytes, const uint8_t* allowed_bytes,
                                const int64_t* allowed_offsets, int64_t n_allowed, int32_t* out_tokens, int64_t out_capacity,
                                int64_t* n_tokens, int32_t* last_piece_token_len) {
    if (!t || n_bytes < 0 || (n_bytes > 0 && !text) || n_allowed < 0 || (n_allowed > 0 && (!allowed_bytes || !allowed_offsets)) || out_capacity < 0)
        return TD_E_INVALID;
    if (n_bytes == 0 && n_allowed == 0) {  // (no text, nothing allowed to validate: no ids, and no reason to wake the device)
        if (n_tokens) *n_tokens = 0;
        if (last_piece_token_len) *last_piece_token_len = 0;
        return TD_OK;
    }
    static const int64_t no_offs[1] = {0};
    return locked(t, [&] {
        const int64_t doc[2] = {0, n_bytes};
        int64_t offs[2] = {0, 0}, lo = 0, hi = 0;
        const int rc = encode_special_locked(t, text, doc, 1, allowed_bytes, n_allowed ? allowed_offsets : no_offs, nullptr, n_allowed,
                                             out_tokens, out_capacity, offs, n_tokens, &lo, &hi);
        if (rc == TD_OK && last_piece_token_len) *last_piece_token_len = last_piece_token_len_host(t, text, lo, hi);
        return rc;
    });
}

int64_t td_info(const td_tokenizer* t, int what) {
    if (!t) return -1;
    switch (what) {
        case TD_INFO_N_PAIRS: return (int64_t)t->H.n_pairs;
        case TD_INFO_MERGE_CLOSED: return t->H.merge_closed ? 1 : 0;
        case TD_INFO_MAX_ID: return t->H.max_id;
        case TD_INFO_TILE_BYTES: return K_TILE;
        case TD_INFO_WORKSPACE_BYTES: return (int64_t)t->ws_bytes;
        case TD_INFO_N_SPECIAL: return (int64_t)t->H.special_ids.size();
        case TD_INFO_LONG_PIECES: return t->last_long;
        case TD_INFO_FAR_PIECES: return t->last_far;
        case TD_INFO_DEFERRED_TILES: return t->last_deferred;
        case TD_INFO_FLAGGED_TILES: return t->last_flagged;
        case TD_INFO_DIRECT_TILES: return t->last_direct;
        case TD_INFO_LB_TIMEOUTS: return t->last_timeouts;
        case TD_INFO_REPEATS: return t->last_repeats;
        case TD_INFO_LISTED_PIECES: return t->last_listed;
        case TD_INFO_CHAR_SEEDS: return (int64_t)t->H.n_char_seeds;
        case TD_INFO_SPARSE: return t->last_sparse ? 1 : 0;
    }
    return -1;
}

int td_set_option(td_tokenizer* t, int what, int64_t value) {
    if (!t) return TD_E_INVALID;
    std::lock_guard<std::mutex> g(t->mu);
    if (what == TD_OPT_LONG_POOL_BYTES && value >= (1 << 20)) {
        t->pool_bytes_opt = value;
        return TD_OK;
    }
#ifdef TD_ABLATE
    if (what == 99) {  // tuning builds only (tools/build_variant.sh -DTD_ABLATE): phase ablation, results are garbage when != 0
        t->stop_after = (int)value;
        return TD_OK;
    }
#endif
    if (what == TD_OPT_FUSED) {
        t->fused = value != 0;
        return TD_OK;
    }
    if (what == TD_OPT_DIRECT) {
        t->direct = value != 0;
        return TD_OK;
    }
    if (what == TD_OPT_DEVICE_SPECIALS) {
        t->device_specials = value != 0;
        return TD_OK;
    }
    if (what == TD_OPT_PACK_SPLIT) {
        t->pack_split = value != 0;
        drop_graph(t); t->has_last_key = false;
        return TD_OK;
    }
    if (what == TD_OPT_SPARSE && value >= -1 && value <= 1) {
        t->sparse_opt = (int)value;
        drop_graph(t); t->has_last_key = false;
        return TD_OK;
    }
    if (what == TD_OPT_OVERLAP) {
        t->overlap = value != 0;
        drop_graph(t); t->has_last_key = false;
        return TD_OK;
    }
    if (what == TD_OPT_GIANT_COOP_MIN && value >= 1024) {
        t->gp_coop_min = (uint32_t)std::min<int64_t>(value, 0x7FFFFFFF);
        drop_graph(t); t->has_last_key = false;
        return TD_OK;
    }
    if (what == TD_OPT_DEDUPE) {
        t->dedupe = value != 0;
        drop_graph(t); t->has_last_key = false;
        return TD_OK;
    }
    if (what == TD_OPT_GRAPH) {
        t->graphs = value != 0;
        if (!t->graphs) { drop_graph(t); t->has_last_key = false; }
        return TD_OK;
    }
    if (what == TD_OPT_PROFILE) {
        t->profile = value != 0;
        return TD_OK;
    }
    if (what == TD_OPT_PIPE_CHUNK_BYTES && value >= 4096) {
        t->pipe_chunk_bytes = value;
        return TD_OK;
    }
    if (what == TD_OPT_SMALL_PATH) {
        t->small_enabled = value != 0;
        return TD_OK;
    }
    if (what == TD_OPT_PIPE_THREADS && value >= 1 && value <= 256) {
        t->pipe_threads = (int)value;
        return TD_OK;
    }
    return TD_E_INVALID;
}

static const char* const kProfSegments[TD_PROF_EVENTS - 1] = {"td_prepare+td_mark_docs", "td_split_tiles", "td_probe_tiles", "td_merge_pieces",
                                                              "td_long_pieces+td_giant_pieces+td_scan_tiles", "td_pack_tokens"};

int td_profile_read_ex(td_tokenizer* t, double* ms_sums, int n_segments, int64_t* launches) {
    if (!t || n_segments < 0 || (n_segments > 0 && !ms_sums)) return TD_E_INVALID;
    return locked(t, [&] {
        double sum[TD_PROF_EVENTS - 1] = {};
        int64_t n = 0;
        for (auto& ev : t->ev_pending) {
            HIP_TRY(t, hipEventSynchronize(ev.e[TD_PROF_EVENTS - 1]));
            for (int k = 0; k + 1 < TD_PROF_EVENTS; ++k) {
                float ms = 0;
                HIP_TRY(t, hipEventElapsedTime(&ms, ev.e[k], ev.e[k + 1]));
                sum[k] += ms;
            }
            ++n;
            t->ev_free.push_back(ev);
        }
        t->ev_pending.clear();
        for (int k = 0; k < n_segments; ++k) ms_sums[k] = k + 1 < TD_PROF_EVENTS ? sum[k] : 0.0;
        if (launches) *launches = n;
        return (int)TD_OK;
    });
}

const char* td_profile_segment_name(int i) { return (i >= 0 && i + 1 < TD_PROF_EVENTS) ? kProfSegments[i] : ""; }

int td_profile_read(td_tokenizer* t, double* split_ms_sum, double* encode_ms_sum, int64_t* launches) {
    double s[TD_PROF_EVENTS - 1] = {};
    const int rc = td_profile_read_ex(t, s, TD_PROF_EVENTS - 1, launches);
    if (split_ms_sum) *split_ms_sum = s[1];
    if (encode_ms_sum) *encode_ms_sum = s[2] + s[3];
    return rc;
}

int64_t td_special_count(const td_tokenizer* t) { return t ? (int64_t)t->H.special_ids.size() : 0; }

int td_special_get(const td_tokenizer* t, int64_t i, const char** str, int64_t* len, int32_t* id) {
    if (!t || i < 0 || i >= (int64_t)t->H.special_ids.size()) return TD_E_INVALID;
    if (str) *str = t->H.special_strs[(size_t)i].data();
    if (len) *len = (int64_t)t->H.special_strs[(size_t)i].size();
    if (id) *id = t->H.special_ids[(size_t)i];
    return TD_OK;
}

int td_token_bytes(const td_tokenizer* t, int32_t id, const uint8_t** bytes, int64_t* len) {
    if (!t) return TD_E_INVALID;
    const HostTables& H = t->H;
    if (id < 0 || id > H.max_id || H.tok_off[(size_t)id + 1] == H.tok_off[(size_t)id]) return TD_E_BAD_TOKEN;
    if (bytes) *bytes = H.tok_bytes.data() + H.tok_off[(size_t)id];
    if (len) *len = (int64_t)H.tok_off[(size_t)id + 1] - (int64_t)H.tok_off[(size_t)id];
    return TD_OK;
}

int td_single_token(const td_tokenizer* t, const uint8_t* bytes, int64_t len, int32_t* id) {
    if (!t || !bytes || len <= 0 || !id) return TD_E_INVALID;
    const HostTables& H = t->H;
    const Tables hv = H.view();
    int32_t r = NO_RANK;
    if (len == 1) {
        r = H.byte_id[bytes[0]];
        if (r >= H.pseudo_base) r = NO_RANK;
    } else if (len <= (int64_t)H.max_token_len) {
        r = piece_lookup(hv, piece_key_host(bytes, (uint32_t)len), (uint32_t)len, [bytes](uint32_t i) { return (uint32_t)bytes[i]; });
    }
    if (r == NO_RANK)
        for (size_t i = 0; i < H.special_ids.size(); ++i)
            if ((int64_t)H.special_strs[i].size() == len && memcmp(H.special_strs[i].data(), bytes, (size_t)len) == 0) { r = H.special_ids[i]; break; }
    if (r == NO_RANK) return TD_E_UNKNOWN_BYTE;
    *id = r;
    return TD_OK;
}

// ---- vocabulary files ------------------------------------------------------------------------------------------
}  // extern "C"

struct td_vocab {
    td::VocabData d;
};

extern "C" {

int td_vocab_create(td_vocab** out) {
    if (!out) return TD_E_INVALID;
    *out = new td_vocab;
    return TD_OK;
}
void td_vocab_destroy(td_vocab* v) { delete v; }
const char* td_vocab_error(const td_vocab* v) { return v ? v->d.err.c_str() : "null td_vocab"; }

int td_vocab_load_tiktoken(td_vocab* v, const char* path) {
    if (!v || !path) return TD_E_INVALID;
    return load_tiktoken_model(path, v->d) ? TD_OK : TD_E_VOCAB;
}
int td_vocab_load_hf_special(td_vocab* v, const char* path, int also_mergeable) {
    if (!v || !path) return TD_E_INVALID;
    return load_hf_added_tokens(path, v->d, also_mergeable != 0) ? TD_OK : TD_E_VOCAB;
}
int td_vocab_load_tekken(td_vocab* v, const char* path) {
    if (!v || !path) return TD_E_INVALID;
    return load_tekken_json(path, v->d) ? TD_OK : TD_E_VOCAB;
}
int td_vocab_load_json(td_vocab* v, const char* vocab_json_path, const char* special_json_path) {
    if (!v || (!vocab_json_path && !special_json_path)) return TD_E_INVALID;
    return load_wrapper_json(vocab_json_path ? vocab_json_path : "", special_json_path ? special_json_path : "", v->d) ? TD_OK
                                                                                                                      : TD_E_VOCAB;
}
int td_vocab_set_pattern(td_vocab* v, const char* pat_str) {
    if (!v || !pat_str) return TD_E_INVALID;
    v->d.pattern = pat_str;
    return TD_OK;
}
const char* td_vocab_pattern(const td_vocab* v) { return v ? v->d.pattern.c_str() : ""; }

int td_vocab_arrays(const td_vocab* v, int which, const uint8_t** bytes, const int64_t** offsets, const int32_t** ranks,
                    int64_t* n) {
    if (!v || (which != 0 && which != 1)) return TD_E_INVALID;
    const td::TokenList& l = which ? v->d.special : v->d.regular;
    static const uint8_t none = 0;
    if (bytes) *bytes = l.bytes.empty() ? &none : l.bytes.data();
    if (offsets) *offsets = l.offsets.data();
    if (ranks) *ranks = l.ranks.data();
    if (n) *n = l.size();
    return TD_OK;
}

int td_create_from_vocab(const td_vocab* v, int device, td_tokenizer** out) {
    if (!v || !out) return TD_E_INVALID;
    const uint8_t *b = nullptr, *sb = nullptr;
    const int64_t *o = nullptr, *so = nullptr;
    const int32_t *r = nullptr, *sr = nullptr;
    int64_t n = 0, ns = 0;
    td_vocab_arrays(v, 0, &b, &o, &r, &n);
    td_vocab_arrays(v, 1, &sb, &so, &sr, &ns);
    return td_create(v->d.pattern.c_str(), n, b, o, r, ns, sb, so, sr, device, out);
}

}  // extern "C"
