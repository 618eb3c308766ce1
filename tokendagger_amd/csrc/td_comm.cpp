// Multi-GPU epilogue of the tokenizer path behind the C ABI (include/tokendagger_hip.h, td_comm_*): documents shard across
// ranks with no data-path collective (the reference's own parallelism is independent texts on a thread pool,
// /root/reference/tokendagger/wrapper.py:231-235), so the only exchange is the gather of every rank's {tokens, documents}
// -> global token / document bases, and — optionally — the token ids themselves to one rank.  RCCL over xGMI:
//   td_comm_gather_counts   ncclAllGather of two int64 per rank, straight out of the offsets buffer the encode step wrote
//   td_comm_gather_tokens   grouped ncclSend / ncclRecv of variable-length int32 id arrays into one root buffer
// RCCL is opened at run time (dlopen "librccl.so.1"): the tokenizer library itself links against HIP only, and a process
// that already holds an RCCL (PyTorch ships one) gets that very copy.  td_comm_bases is the host half (prefix sums over the
// gathered table) and needs no device.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>  // types and prototypes only

#include <mutex>
#include <string>
#include <vector>

#include "tokendagger_hip.h"

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};

thread_local std::string g_comm_err;

Rccl* rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.lib) break;
        }
        if (!R.lib) { R.err = std::string("RCCL is not available: ") + (dlerror() ? dlerror() : "dlopen failed"); return; }
#define TD_SYM(n) R.n = reinterpret_cast<decltype(R.n)>(dlsym(R.lib, "nccl" #n)); if (!R.n) { R.err = "RCCL symbol nccl" #n " missing"; return; }
        TD_SYM(GetUniqueId) TD_SYM(CommInitRank) TD_SYM(CommDestroy) TD_SYM(AllGather) TD_SYM(GroupStart) TD_SYM(GroupEnd) TD_SYM(Send) TD_SYM(Recv)
        TD_SYM(GetErrorString)
#undef TD_SYM
    });
    return &R;
}

int fail(int rc, const std::string& msg) {
    g_comm_err = msg;
    return rc;
}
int nccl_fail(Rccl* R, const char* what, ncclResult_t r) {
    return fail(TD_E_HIP, std::string(what) + ": " + (R->GetErrorString ? R->GetErrorString(r) : "RCCL error"));
}

struct DeviceScope {  // the caller's current device survives the call
    int prev = -1;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (dev >= 0 && dev != prev) (void)hipSetDevice(dev);
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct td_comm {
    ncclComm_t comm = nullptr;
    int world = 1, rank = 0, device = 0;
};

extern "C" {

const char* td_comm_last_error(void) { return g_comm_err.c_str(); }

int td_comm_unique_id(uint8_t id[TD_COMM_ID_BYTES]) {
    static_assert(TD_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's");
    if (!id) return fail(TD_E_INVALID, "td_comm_unique_id: null id");
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(TD_E_HIP, R->err);
    ncclUniqueId u;
    const ncclResult_t r = R->GetUniqueId(&u);
    if (r != ncclSuccess) return nccl_fail(R, "ncclGetUniqueId", r);
    for (int i = 0; i < TD_COMM_ID_BYTES; ++i) id[i] = (uint8_t)u.internal[i];
    return TD_OK;
}

int td_comm_create(const uint8_t id[TD_COMM_ID_BYTES], int world, int rank, int device, td_comm** out) {
    if (!out) return fail(TD_E_INVALID, "td_comm_create: null out");
    *out = nullptr;
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(TD_E_INVALID, "td_comm_create: bad argument");
    Rccl* R = rccl();
    if (!R->err.empty()) return fail(TD_E_HIP, R->err);
    int dev = device;
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) return fail(TD_E_HIP, "td_comm_create: no HIP device");
    DeviceScope ds(dev);
    ncclUniqueId u;
    for (int i = 0; i < TD_COMM_ID_BYTES; ++i) u.internal[i] = (char)id[i];
    td_comm* c = new td_comm;
    c->world = world; c->rank = rank; c->device = dev;
    const ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) { delete c; return nccl_fail(R, "ncclCommInitRank", r); }
    *out = c;
    return TD_OK;
}

void td_comm_destroy(td_comm* c) {
    if (!c) return;
    Rccl* R = rccl();
    if (c->comm && R->CommDestroy) {
        DeviceScope ds(c->device);
        (void)R->CommDestroy(c->comm);
    }
    delete c;
}

int td_comm_gather_counts(td_comm* c, const int64_t* d_counts, int64_t* d_table, void* stream) {
    if (!c || !d_counts || !d_table) return fail(TD_E_INVALID, "td_comm_gather_counts: bad argument");
    Rccl* R = rccl();
    DeviceScope ds(c->device);
    const ncclResult_t r = R->AllGather(d_counts, d_table, 2, ncclInt64, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) return nccl_fail(R, "ncclAllGather", r);
    return TD_OK;
}

int td_comm_bases(const int64_t* table, int world, int rank, int64_t* token_base, int64_t* doc_base, int64_t* token_total, int64_t* doc_total) {
    if (!table || world < 1 || rank < 0 || rank >= world) return fail(TD_E_INVALID, "td_comm_bases: bad argument");
    int64_t tb = 0, db = 0, tt = 0, dt = 0;
    for (int r = 0; r < world; ++r) {
        const int64_t tk = table[2 * r], dc = table[2 * r + 1];
        if (tk < 0 || dc < 0) return fail(TD_E_INVALID, "td_comm_bases: negative count in the gathered table");
        if (r < rank) { tb += tk; db += dc; }
        tt += tk; dt += dc;
    }
    if (token_base) *token_base = tb;
    if (doc_base) *doc_base = db;
    if (token_total) *token_total = tt;
    if (doc_total) *doc_total = dt;
    return TD_OK;
}

int td_comm_gather_tokens(td_comm* c, const int32_t* d_tokens, const int64_t* table, int root, int32_t* d_root_tokens, int64_t root_capacity,
                          void* stream) {
    if (!c || !table || root < 0 || root >= c->world) return fail(TD_E_INVALID, "td_comm_gather_tokens: bad argument");
    Rccl* R = rccl();
    DeviceScope ds(c->device);
    const int64_t mine = table[2 * c->rank];
    if (mine > 0 && !d_tokens) return fail(TD_E_INVALID, "td_comm_gather_tokens: null token buffer");
    int64_t total = 0;
    for (int r = 0; r < c->world; ++r) total += table[2 * r];
    // A root buffer that is too small must not leave the other ranks' sends without a receive (they cannot see root_capacity,
    // ADVICE r3): the root still takes part — it receives into a scratch buffer of the right size, and reports TD_E_CAPACITY
    // once the exchange is complete on its stream.  Every other rank's call succeeds.
    hipStream_t s = (hipStream_t)stream;
    int32_t* dst = d_root_tokens;
    void* scratch = nullptr;
    const bool short_root = c->rank == root && (total > root_capacity || (total > 0 && !d_root_tokens));
    if (short_root) {
        if (hipMalloc(&scratch, (size_t)total * 4) != hipSuccess) {
            (void)hipGetLastError();
            return fail(TD_E_HIP, "td_comm_gather_tokens: root buffer too small (" + std::to_string(total) +
                                      " ids) and no memory for a scratch buffer to complete the exchange: the other ranks' sends are pending");
        }
        dst = (int32_t*)scratch;
    }
    auto done = [&](int rc) {
        if (scratch) {
            (void)hipStreamSynchronize(s);
            (void)hipFree(scratch);
        }
        return rc;
    };
    ncclResult_t r = R->GroupStart();
    if (r != ncclSuccess) return done(nccl_fail(R, "ncclGroupStart", r));
    hipError_t he = hipSuccess;
    if (c->rank == root) {
        int64_t base = 0;
        for (int p = 0; p < c->world; ++p) {
            const int64_t cnt = table[2 * p];
            if (p == root) {
                if (cnt > 0 && !short_root && dst + base != d_tokens) {
                    const hipError_t e1 = hipMemcpyAsync(dst + base, d_tokens, (size_t)cnt * 4, hipMemcpyDeviceToDevice, s);
                    if (he == hipSuccess) he = e1;
                }
            } else if (cnt > 0) {
                r = R->Recv(dst + base, (size_t)cnt, ncclInt32, p, c->comm, s);
                if (r != ncclSuccess) { (void)R->GroupEnd(); return done(nccl_fail(R, "ncclRecv", r)); }
            }
            base += cnt;
        }
    } else if (mine > 0) {
        r = R->Send(d_tokens, (size_t)mine, ncclInt32, root, c->comm, s);
        if (r != ncclSuccess) { (void)R->GroupEnd(); return nccl_fail(R, "ncclSend", r); }
    }
    r = R->GroupEnd();
    if (r != ncclSuccess) return done(nccl_fail(R, "ncclGroupEnd", r));
    if (he != hipSuccess) return done(fail(TD_E_HIP, std::string("td_comm_gather_tokens: hipMemcpyAsync: ") + hipGetErrorString(he)));
    if (short_root) return done(fail(TD_E_CAPACITY, "td_comm_gather_tokens: root buffer too small: " + std::to_string(total) + " ids"));
    return TD_OK;
}

}  // extern "C"
