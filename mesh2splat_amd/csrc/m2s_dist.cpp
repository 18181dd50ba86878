// m2s_dist.cpp — the multi-GPU exchange behind the C ABI (include/m2s.h, "multi-GPU"): one process per GPU, triangle-range
// shards, RCCL over xGMI.
//
// The reference is a single-GPU program (no distributed backend), so nothing here replaces reference code;
// it implements BASELINE.json's north star: "work shards naturally by triangle range across up to 8 MI355X with a single
// RCCL all-gather over xGMI to concatenate per-rank splat buffers".  The conversion needs no data-path collective: every
// triangle is independent (no depth test, no blending, framebuffer unused: ConversionPass.cpp:45-48) and the only shared
// state of the reference, the append cursor (converterFS.glsl:46), becomes one counter per rank.  What ranks exchange:
//   * counts  — 8 bytes per rank per conversion (ncclAllGather): every rank learns its offset in the merged buffer;
//   * records — optional: every rank's block to every rank (or to one root) at its final offset, exact sizes.  RCCL has
//               no all-gather-v, and xGMI is point-to-point (7 links per GPU): the exchange is one ncclGroup of
//               ncclSend/ncclRecv pairs, peer order staggered per rank so that no two ranks start on the same link.
// A consumer that writes files does not need the records on one GPU at all: see m2s_export_ply_slice (each rank writes its
// rows at the right byte offset of one .ply).
//
// Two transports behind the same entry points: RCCL (one PROCESS per GPU: the id comes from m2s_dist_unique_id) and an
// in-process one (one THREAD per GPU inside one process — the shape of the reference, a single executable: the id comes from
// m2s_dist_local_id; all-gathers go through the group's shared table, record transfers are hipMemcpyPeerAsync, which takes
// the same xGMI links).  Everything above the transport — shard plan, counter exchange, record exchange, the sample sort
// of m2s_dist_sort_by_depth — is the same code for both.
//
// librccl is opened at run time (dlopen), not linked: a Python process that already carries PyTorch's bundled RCCL must
// not end up with a second copy, and single-GPU users of libm2s_hip.so need no RCCL at all.
#include "../../include/m2s.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <random>
#include <string>
#include <vector>

namespace m2s {   // m2s_sort.hip
void launch_pick_samples(const uint32_t* keys, uint64_t n, uint32_t s, unsigned long long* out, hipStream_t st);
void launch_lower_bounds(const uint32_t* keys, uint64_t n, const unsigned long long* splitters, uint32_t m, unsigned long long* out, hipStream_t st);
}

namespace {

// ---- the slice of the RCCL API this file uses (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220, ncclAllGather :678,
// ncclSend :700, ncclRecv :722, ncclGroupStart/End :923/:933) ----
typedef struct { char internal[128]; } ncclUniqueId_t;
typedef void* ncclComm_p;
enum { kNcclSuccess = 0, kNcclUint8 = 1, kNcclUint64 = 5 };
struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_p*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_p) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_p, hipStream_t) = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_p, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
    std::string path;       // M2S_RCCL_PATH if that is what was loaded (m2s_dist_transport reports it)
};
Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string g_dist_error;

void load_rccl() {
    Rccl& r = g_rccl;
    // an explicitly named library first (M2S_RCCL_PATH: a site's own build — or the test stand-in of tests/stub_rccl, which lets
    // several PROCESSES share one GPU; loaded RTLD_LOCAL so that its nccl* symbols never interpose on a real RCCL in the same
    // process), then an RCCL that is already in the process (PyTorch's), then the system one
    if (const char* e = std::getenv("M2S_RCCL_PATH")) {
        r.handle = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        // (dlerror() clears the error it returns: ask once — ADVICE r4: the second call returned NULL into std::string's operator+)
        if (!r.handle) { const char* de = dlerror(); r.error = std::string("M2S_RCCL_PATH: ") + (de ? de : "could not be loaded"); return; }
        r.path = e;
    }
    const char* names[] = { "librccl.so.1", "librccl.so" };
    for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    const char* paths[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for (const char* n : paths) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) { const char* de = dlerror(); r.error = std::string("librccl could not be loaded: ") + (de ? de : "?"); return; }
    auto sym = [&](const char* s) { void* p = dlsym(r.handle, s); if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + s; return p; };
    r.GetUniqueId = reinterpret_cast<int (*)(ncclUniqueId_t*)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<int (*)(ncclComm_p*, int, ncclUniqueId_t, int)>(sym("ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<int (*)(ncclComm_p)>(sym("ncclCommDestroy"));
    r.AllGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, ncclComm_p, hipStream_t)>(sym("ncclAllGather"));
    r.Send = reinterpret_cast<int (*)(const void*, size_t, int, int, ncclComm_p, hipStream_t)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<int (*)(void*, size_t, int, int, ncclComm_p, hipStream_t)>(sym("ncclRecv"));
    r.GroupStart = reinterpret_cast<int (*)()>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<int (*)()>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<const char* (*)(int)>(sym("ncclGetErrorString"));
}

bool rccl_ready() {
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.error.empty() || !g_rccl.handle) { g_dist_error = g_rccl.error.empty() ? "librccl unavailable" : g_rccl.error; return false; }
    return true;
}

constexpr int kRing = 8;   // counter exchanges in flight
constexpr uint32_t kSortSamples = 256;   // samples per rank for the splitters of m2s_dist_sort_by_depth

// ---- the in-process transport: the ranks are threads of this process ----
struct LocalGroup {
    int world = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, joined = 0, left = 0;
    uint64_t generation = 0;
    bool broken = false;
    struct Post { const void* ptr = nullptr; int device = 0; size_t bytes = 0; };
    std::vector<Post> gathered;                    // all-gather: what every rank contributes
    struct Send { int dst = 0; const void* ptr = nullptr; size_t bytes = 0; int device = 0; };
    std::vector<std::vector<Send>> sends;          // record exchange: what every rank sends, in its own order
    std::vector<char> rank_joined;                 // a rank joins once
};
// In-process groups live in a process-wide registry and are named by a random token: the 128-byte id carries the token, not a
// pointer, so an id whose group has been released (or a forged one) is refused instead of dereferenced, and a rank cannot join
// twice.  A group is released when every rank that joined has left and all `world` ranks had joined; a group some rank never
// joined (it failed before m2s_dist_create) stays registered — a few hundred bytes — rather than being freed under a late joiner.
std::mutex g_groups_lock;
std::map<unsigned long long, LocalGroup*> g_groups;
constexpr char kLocalMagic[8] = { 'M', '2', 'S', 'L', 'O', 'C', 'A', 'L' };
constexpr int kLocalBarrierSeconds = 120;

// all ranks of the group; false if one of them never arrives (the group is then unusable for everybody)
bool local_barrier(LocalGroup* g) {
    std::unique_lock<std::mutex> l(g->m);
    if (g->broken) return false;
    const uint64_t gen = g->generation;
    if (++g->arrived == g->world) {
        g->arrived = 0;
        ++g->generation;
        g->cv.notify_all();
        return true;
    }
    if (!g->cv.wait_for(l, std::chrono::seconds(kLocalBarrierSeconds), [&] { return g->generation != gen || g->broken; })) {
        g->broken = true;
        g->cv.notify_all();
        return false;
    }
    return !g->broken;
}

struct Xfer { int peer; void* ptr; size_t bytes; };   // one message of a record exchange

}  // namespace

struct m2s_dist {
    int device = 0, rank = 0, world = 1;
    ncclComm_p comm = nullptr;                 // RCCL transport ...
    LocalGroup* local = nullptr;               // ... or the in-process one
    hipStream_t stream = nullptr;              // counter exchanges run here, off the conversion stream
    unsigned long long* d_sort = nullptr;      // m2s_dist_sort_by_depth: samples | all samples | splitters, bounds | send counts | matrix
    unsigned long long* h_sort = nullptr;      // pinned mirror
    unsigned long long* d_mine = nullptr;      // [kRing]
    unsigned long long* d_all = nullptr;       // [kRing][world]
    unsigned long long* h_mine = nullptr;      // pinned [kRing]
    unsigned long long* h_all = nullptr;       // pinned [kRing][world]
    hipEvent_t done[kRing] = {};
    uint64_t published = 0, collected = 0;
    std::string err;
    // The four runtime calls of one counter exchange (H2D, ncclAllGather, D2H, event) cost ~10 us of host time: too much to sit
    // in the thread that submits 0.13 ms conversions back to back.  m2s_dist_publish_count only queues the value; this worker
    // issues the calls.  Every use of the communicator (worker, m2s_dist_gather_records) holds comm_lock.
    std::thread worker;
    std::mutex q_lock, comm_lock;
    std::condition_variable q_cv, issued_cv;
    std::deque<std::pair<uint64_t, unsigned long long>> queue;   // (exchange number, value)
    uint64_t issued = 0;                       // exchanges whose calls have been issued (guarded by q_lock)
    bool stop = false;
    std::string worker_err;                    // first failure inside the worker (guarded by q_lock)
};

#define DCHK(d, call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { (d)->err = std::string(#call) + ": " + hipGetErrorString(e_); return M2S_ERR_HIP; } \
    } while (0)
#define NCHK(d, call)                                                                      \
    do {                                                                                   \
        int r_ = (call);                                                                   \
        if (r_ != kNcclSuccess) {                                                          \
            (d)->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); \
            return M2S_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)

// ---- transport: all-gather of 64-bit words, exchange of byte ranges (both on device memory) ----
#define TH(call)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return M2S_ERR_HIP; } \
    } while (0)
#define TN(call)                                                                           \
    do {                                                                                   \
        int r_ = (call);                                                                   \
        if (r_ != kNcclSuccess) {                                                          \
            err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "RCCL error"); \
            return M2S_ERR_HIP;                                                            \
        }                                                                                  \
    } while (0)
// RCCL: enqueued on `st`, returns at once.  In-process: completed when it returns.
// (errors go to `err`, not to d->err: the exchange thread calls these too)
static m2s_status t_all_gather(m2s_dist* d, const unsigned long long* d_send, unsigned long long* d_recv, size_t count, hipStream_t st, std::string& err) {
    if (!d->local) {
        TN(g_rccl.AllGather(d_send, d_recv, count, kNcclUint64, d->comm, st));
        return M2S_OK;
    }
    LocalGroup* g = d->local;
    // Every rank passes BOTH barriers on every path: a rank that hit an error still arrives (its peers would otherwise stall
    // for kLocalBarrierSeconds and the group would be left broken), and reports the error afterwards.
    m2s_status result = M2S_OK;
    if (hipStreamSynchronize(st) != hipSuccess) { err = "all-gather: stream failed before the exchange"; result = M2S_ERR_HIP; }   // what I contribute is complete
    {
        std::lock_guard<std::mutex> l(g->m);
        g->gathered[(size_t)d->rank] = LocalGroup::Post{ d_send, d->device, result == M2S_OK ? count * sizeof(unsigned long long) : ~(size_t)0 };
    }
    if (!local_barrier(g)) { err = "a rank of the in-process group did not arrive"; return M2S_ERR_STATE; }
    for (int r = 0; r < d->world && result == M2S_OK; ++r) {
        const LocalGroup::Post& p = g->gathered[(size_t)r];   // (stable between the two barriers)
        if (p.bytes != count * sizeof(unsigned long long)) { err = p.bytes == ~(size_t)0 ? "all-gather: a rank failed before the exchange" : "all-gather sizes differ between ranks"; result = M2S_ERR_STATE; break; }
        if (hipMemcpyPeerAsync(d_recv + (size_t)r * count, d->device, p.ptr, p.device, p.bytes, st) != hipSuccess) { err = "all-gather: hipMemcpyPeerAsync failed"; result = M2S_ERR_HIP; }
    }
    if (hipStreamSynchronize(st) != hipSuccess && result == M2S_OK) { err = "all-gather: stream failed"; result = M2S_ERR_HIP; }
    if (!local_barrier(g) && result == M2S_OK) { err = "a rank of the in-process group did not arrive"; result = M2S_ERR_STATE; }   // nobody overwrites its contribution earlier
    return result;
}

// Every message is described on both sides (exact sizes).  sends[i].ptr is read, recvs[i].ptr written; messages between one
// pair of ranks match in the order given.  `steps` interleaves them the way the caller staggered its peers.
static m2s_status t_exchange(m2s_dist* d, const std::vector<Xfer>& sends, const std::vector<Xfer>& recvs, hipStream_t st, std::string& err) {
    if (!d->local) {
        TN(g_rccl.GroupStart());
        const size_t n = std::max(sends.size(), recvs.size());
        for (size_t i = 0; i < n; ++i) {
            if (i < sends.size() && sends[i].bytes) TN(g_rccl.Send(sends[i].ptr, sends[i].bytes, kNcclUint8, sends[i].peer, d->comm, st));
            if (i < recvs.size() && recvs[i].bytes) TN(g_rccl.Recv(recvs[i].ptr, recvs[i].bytes, kNcclUint8, recvs[i].peer, d->comm, st));
        }
        TN(g_rccl.GroupEnd());
        return M2S_OK;
    }
    LocalGroup* g = d->local;
    m2s_status result = M2S_OK;
    if (hipStreamSynchronize(st) != hipSuccess) { err = "record exchange: stream failed before the exchange"; result = M2S_ERR_HIP; }   // what I send is complete
    {
        std::lock_guard<std::mutex> l(g->m);
        auto& mine = g->sends[(size_t)d->rank];
        mine.clear();
        if (result == M2S_OK)
            for (const Xfer& x : sends) if (x.bytes) mine.push_back(LocalGroup::Send{ x.peer, x.ptr, x.bytes, d->device });
    }
    if (!local_barrier(g)) { err = "a rank of the in-process group did not arrive"; return M2S_ERR_STATE; }
    std::vector<size_t> next((size_t)d->world, 0);
    for (const Xfer& x : recvs) {   // (both barriers are passed on every path: see t_all_gather)
        if (!x.bytes || result != M2S_OK) continue;
        const auto& theirs = g->sends[(size_t)x.peer];
        size_t& k = next[(size_t)x.peer];
        while (k < theirs.size() && theirs[k].dst != d->rank) ++k;
        if (k == theirs.size() || theirs[k].bytes != x.bytes) { err = "record exchange: the two sides of a message disagree"; result = M2S_ERR_STATE; break; }
        if (hipMemcpyPeerAsync(x.ptr, d->device, theirs[k].ptr, theirs[k].device, x.bytes, st) != hipSuccess) { err = "hipMemcpyPeerAsync failed"; result = M2S_ERR_HIP; break; }
        ++k;
    }
    if (hipStreamSynchronize(st) != hipSuccess && result == M2S_OK) { err = "record exchange: stream failed"; result = M2S_ERR_HIP; }
    if (!local_barrier(g) && result == M2S_OK) { err = "a rank of the in-process group did not arrive"; result = M2S_ERR_STATE; }
    return result;
}
#undef TH
#undef TN

// issues the runtime calls of queued counter exchanges, in order
static void dist_worker(m2s_dist* d) {
    (void)hipSetDevice(d->device);
    for (;;) {
        std::pair<uint64_t, unsigned long long> job;
        {
            std::unique_lock<std::mutex> l(d->q_lock);
            d->q_cv.wait(l, [&] { return d->stop || !d->queue.empty(); });
            if (d->queue.empty()) return;       // stop requested and nothing left
            job = d->queue.front();
            d->queue.pop_front();
        }
        const int k = (int)(job.first % kRing);
        std::string err;
        {
            std::lock_guard<std::mutex> c(d->comm_lock);
            d->h_mine[k] = job.second;
            hipError_t e = hipMemcpyAsync(d->d_mine + k, d->h_mine + k, 8, hipMemcpyHostToDevice, d->stream);
            m2s_status r = M2S_OK;
            std::string terr;
            if (e == hipSuccess) r = t_all_gather(d, d->d_mine + k, d->d_all + (size_t)k * d->world, 1, d->stream, terr);
            if (e == hipSuccess && r == M2S_OK)
                e = hipMemcpyAsync(d->h_all + (size_t)k * d->world, d->d_all + (size_t)k * d->world, (size_t)d->world * 8, hipMemcpyDeviceToHost, d->stream);
            if (e == hipSuccess && r == M2S_OK) e = hipEventRecord(d->done[k], d->stream);
            if (r != M2S_OK) err = std::string("counter all-gather: ") + terr;
            else if (e != hipSuccess) err = std::string("counter exchange: ") + hipGetErrorString(e);
        }
        {
            std::lock_guard<std::mutex> l(d->q_lock);
            if (!err.empty() && d->worker_err.empty()) d->worker_err = err;
            d->issued = job.first + 1;
        }
        d->issued_cv.notify_all();
    }
}

extern "C" {

const char* m2s_dist_last_error(const m2s_dist* d) { return d ? d->err.c_str() : g_dist_error.c_str(); }

m2s_status m2s_dist_unique_id(uint8_t out_id[M2S_DIST_ID_BYTES]) {
    if (!out_id) { g_dist_error = "out_id is NULL"; return M2S_ERR_INVALID; }
    if (!rccl_ready()) return M2S_ERR_STATE;
    ncclUniqueId_t id;
    const int r = g_rccl.GetUniqueId(&id);
    if (r != kNcclSuccess) { g_dist_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(r); return M2S_ERR_HIP; }
    static_assert(sizeof id == M2S_DIST_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(out_id, &id, sizeof id);
    return M2S_OK;
}

// The id of an in-process group of `world` ranks (threads of this process, one context each).  Every rank must then call
// m2s_dist_create with it (the group is released by the last m2s_dist_destroy).
m2s_status m2s_dist_local_id(int world, uint8_t out_id[M2S_DIST_ID_BYTES]) {
    if (!out_id || world < 1) { g_dist_error = "bad argument"; return M2S_ERR_INVALID; }
    LocalGroup* g = new (std::nothrow) LocalGroup();
    if (!g) { g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    try {
        g->world = world;
        g->gathered.resize((size_t)world);
        g->sends.resize((size_t)world);
    } catch (...) { delete g; g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    try { g->rank_joined.assign((size_t)world, 0); } catch (...) { delete g; g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    unsigned long long token = 0;
    {
        std::lock_guard<std::mutex> l(g_groups_lock);
        std::random_device rd;
        do { token = ((unsigned long long)rd() << 32) ^ rd(); } while (!token || g_groups.count(token));
        try { g_groups[token] = g; } catch (...) { delete g; g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    }
    memset(out_id, 0, M2S_DIST_ID_BYTES);
    memcpy(out_id, kLocalMagic, sizeof kLocalMagic);
    memcpy(out_id + 8, &token, sizeof token);
    memcpy(out_id + 16, &world, sizeof world);
    return M2S_OK;
}

m2s_status m2s_dist_create(int device, const uint8_t id[M2S_DIST_ID_BYTES], int rank, int world, m2s_dist** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) { g_dist_error = "bad argument"; return M2S_ERR_INVALID; }
    *out = nullptr;
    const bool is_local = memcmp(id, kLocalMagic, sizeof kLocalMagic) == 0;
    if (!is_local && !rccl_ready()) return M2S_ERR_STATE;
    m2s_dist* d = new (std::nothrow) m2s_dist();
    if (!d) { g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    d->device = device; d->rank = rank; d->world = world;
    auto bail = [&](m2s_status s) { g_dist_error = d->err; m2s_dist_destroy(d); return s; };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) { d->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return bail(M2S_ERR_NO_DEVICE); }
    if (is_local) {
        unsigned long long token = 0;
        int gw = 0;
        memcpy(&token, id + 8, sizeof token);
        memcpy(&gw, id + 16, sizeof gw);
        LocalGroup* g = nullptr;
        {
            std::lock_guard<std::mutex> reg(g_groups_lock);
            auto it = g_groups.find(token);
            if (it == g_groups.end()) { d->err = "unknown in-process group (released already, or not an id of m2s_dist_local_id)"; return bail(M2S_ERR_INVALID); }
            g = it->second;
            std::lock_guard<std::mutex> l(g->m);
            if (gw != world || g->world != world) { d->err = "the in-process group was made for another world size"; return bail(M2S_ERR_INVALID); }
            if (g->rank_joined[(size_t)rank]) { d->err = "this rank has already joined the in-process group"; return bail(M2S_ERR_INVALID); }
            g->rank_joined[(size_t)rank] = 1;
            ++g->joined;
        }
        d->local = g;
    } else {
        ncclUniqueId_t uid;
        memcpy(&uid, id, sizeof uid);
        const int r = g_rccl.CommInitRank(&d->comm, world, uid, rank);
        if (r != kNcclSuccess) { d->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); d->comm = nullptr; return bail(M2S_ERR_HIP); }
    }
    const size_t sort_words = (size_t)(kSortSamples + 2) * (1 + (size_t)world) + 2 * (size_t)world + ((size_t)world + 1) * (1 + (size_t)world) + 1 + (size_t)world;   // see m2s_dist_sort_by_depth
    if ((e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_mine, kRing * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_all, (size_t)kRing * world * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipHostMalloc((void**)&d->h_mine, kRing * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void**)&d->h_all, (size_t)kRing * world * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_sort, sort_words * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipHostMalloc((void**)&d->h_sort, sort_words * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) {
        d->err = std::string("allocation: ") + hipGetErrorString(e);
        return bail(M2S_ERR_HIP);
    }
    for (auto& ev : d->done)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) { d->err = std::string("hipEventCreate: ") + hipGetErrorString(e); return bail(M2S_ERR_HIP); }
    try { d->worker = std::thread(dist_worker, d); } catch (...) { d->err = "could not start the exchange thread"; return bail(M2S_ERR_OOM); }
    *out = d;
    return M2S_OK;
}

void m2s_dist_destroy(m2s_dist* d) {
    if (!d) return;
    if (d->worker.joinable()) {
        { std::lock_guard<std::mutex> l(d->q_lock); d->stop = true; }
        d->q_cv.notify_all();
        d->worker.join();
    }
    (void)hipSetDevice(d->device);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(d->comm);
    for (auto& ev : d->done) if (ev) (void)hipEventDestroy(ev);
    if (d->d_mine) (void)hipFree(d->d_mine);
    if (d->d_all) (void)hipFree(d->d_all);
    if (d->h_mine) (void)hipHostFree(d->h_mine);
    if (d->h_all) (void)hipHostFree(d->h_all);
    if (d->d_sort) (void)hipFree(d->d_sort);
    if (d->h_sort) (void)hipHostFree(d->h_sort);
    if (d->stream) (void)hipStreamDestroy(d->stream);
    if (LocalGroup* g = d->local) {   // the group goes with its last member, once every rank of it has joined and left
        std::lock_guard<std::mutex> reg(g_groups_lock);
        bool last;
        { std::lock_guard<std::mutex> l(g->m); last = ++g->left == g->world && g->joined == g->world; }
        if (last) {
            for (auto it = g_groups.begin(); it != g_groups.end(); ++it) if (it->second == g) { g_groups.erase(it); break; }
            delete g;
        }
    }
    delete d;
}

int m2s_dist_rank(const m2s_dist* d) { return d ? d->rank : -1; }
// what moves the bytes of this communicator: "in-process" (ranks are threads: m2s_dist_local_id), "rccl" (the process's or the
// system's librccl) or "rccl:<path>" (the library M2S_RCCL_PATH named)
const char* m2s_dist_transport(const m2s_dist* d) {
    static thread_local std::string s;
    if (!d) return "";
    if (d->local) return "in-process";
    s = g_rccl.path.empty() ? "rccl" : "rccl:" + g_rccl.path;
    return s.c_str();
}
int m2s_dist_world(const m2s_dist* d) { return d ? d->world : 0; }

// Starts the exchange of this rank's counter for one conversion; returns at once (the value is queued for the exchange
// thread).  Exchanges complete in order.
m2s_status m2s_dist_publish_count(m2s_dist* d, uint64_t my_total) {
    if (!d) return M2S_ERR_INVALID;
    if (d->published - d->collected >= (uint64_t)kRing) { d->err = "too many counter exchanges in flight: m2s_dist_collect_counts first"; return M2S_ERR_STATE; }
    {
        std::lock_guard<std::mutex> l(d->q_lock);
        if (!d->worker_err.empty()) { d->err = d->worker_err; return M2S_ERR_HIP; }
        d->queue.emplace_back(d->published, (unsigned long long)my_total);
    }
    d->q_cv.notify_one();
    ++d->published;
    return M2S_OK;
}

// Waits for the OLDEST exchange started with m2s_dist_publish_count: counts[r] = counter of rank r; offsets (exclusive
// prefix, world + 1 entries, may be NULL) = where each rank's block starts in the merged buffer.
m2s_status m2s_dist_collect_counts(m2s_dist* d, uint64_t* counts, uint64_t* offsets) {
    if (!d || !counts) return M2S_ERR_INVALID;
    if (d->collected == d->published) { d->err = "no counter exchange in flight"; return M2S_ERR_STATE; }
    {
        std::unique_lock<std::mutex> l(d->q_lock);
        d->issued_cv.wait(l, [&] { return d->issued > d->collected; });     // its calls have been issued
        if (!d->worker_err.empty()) { d->err = d->worker_err; ++d->collected; return M2S_ERR_HIP; }
    }
    DCHK(d, hipSetDevice(d->device));
    const int k = (int)(d->collected % kRing);
    DCHK(d, hipEventSynchronize(d->done[k]));
    uint64_t run = 0;
    for (int r = 0; r < d->world; ++r) {
        counts[r] = d->h_all[(size_t)k * d->world + r];
        if (offsets) offsets[r] = run;
        run += counts[r];
    }
    if (offsets) offsets[d->world] = run;
    ++d->collected;
    return M2S_OK;
}

m2s_status m2s_dist_all_gather_counts(m2s_dist* d, uint64_t my_total, uint64_t* counts, uint64_t* offsets) {
    if (!d) return M2S_ERR_INVALID;
    while (d->collected != d->published) {   // keep the order: drain older exchanges first
        std::vector<uint64_t> tmp((size_t)d->world);
        const m2s_status s = m2s_dist_collect_counts(d, tmp.data(), nullptr);
        if (s != M2S_OK) return s;
    }
    const m2s_status s = m2s_dist_publish_count(d, my_total);
    return s != M2S_OK ? s : m2s_dist_collect_counts(d, counts, offsets);
}

// Global u_maxGaussians semantics for a sharded conversion (converterFS.glsl:46-51): the merged buffer keeps the first
// `cap` records in rank order (0 = unlimited); keep[r] = how many records rank r contributes.
void m2s_dist_clamp_to_cap(const uint64_t* counts, int world, uint64_t cap, uint64_t* keep) {
    uint64_t left = cap;
    for (int r = 0; r < world; ++r) {
        keep[r] = cap ? std::min(counts[r], left) : counts[r];
        if (cap) left -= keep[r];
    }
}

// The shard plan (host only, deterministic: every rank computes the same one).  The flattened (mesh-major) triangle list
// is cut into `world` contiguous ranges of about equal COST = estimated fragments + 0.25 per triangle (emission costs per
// fragment, setup per triangle).  The estimate is the area of the triangle projected on its dominant axis plane, in
// pixels of the R x R viewport — what the rasteriser covers up to boundary effects; it uses each mesh's (cumulative,
// SceneManager.cpp:476-527) bounding box exactly like the geometry shader does (converterGS.glsl:353-399).
m2s_status m2s_dist_shard_ranges(const m2s_mesh* meshes, uint32_t n_meshes, uint32_t R, int world, uint64_t* first, uint64_t* count) {
    if ((n_meshes && !meshes) || world < 1 || !first || !count) { g_dist_error = "bad argument"; return M2S_ERR_INVALID; }
    try {
        uint64_t T = 0;
        for (uint32_t i = 0; i < n_meshes; ++i) {
            if (meshes[i].stride_floats < 12 || meshes[i].n_vertices % 3 || (meshes[i].n_vertices && !meshes[i].vertices)) { g_dist_error = "bad mesh"; return M2S_ERR_INVALID; }
            T += meshes[i].n_vertices / 3;
        }
        std::vector<double> cum((size_t)T);
        double run = 0.0;
        size_t k = 0;
        const double RR = (double)R * (double)R;
        for (uint32_t i = 0; i < n_meshes; ++i) {
            const m2s_mesh& m = meshes[i];
            const double ex = (double)m.bbox_max[0] - (double)m.bbox_min[0], ey = (double)m.bbox_max[1] - (double)m.bbox_min[1],
                         ez = (double)m.bbox_max[2] - (double)m.bbox_min[2];
            const size_t st = m.stride_floats;
            for (uint32_t t = 0; t < m.n_vertices / 3; ++t, ++k) {
                const float* v0 = m.vertices + (size_t)t * 3 * st;
                const float *v1 = v0 + st, *v2 = v1 + st;
                const double ax = (double)v1[0] - v0[0], ay = (double)v1[1] - v0[1], az = (double)v1[2] - v0[2];
                const double bx = (double)v2[0] - v0[0], by = (double)v2[1] - v0[1], bz = (double)v2[2] - v0[2];
                const double nx = std::fabs(ay * bz - az * by), ny = std::fabs(az * bx - ax * bz), nz = std::fabs(ax * by - ay * bx);
                double rng, proj2;
                if (nx > ny && nx > nz) { rng = std::max(ey, ez); proj2 = nx; }
                else if (ny > nz) { rng = std::max(ex, ez); proj2 = ny; }
                else { rng = std::max(ex, ey); proj2 = nz; }
                double px = 0.5 * proj2 / (rng * rng) * RR;
                if (!(px == px) || px > 1.0e300 || px < 0.0) px = 0.0;      // NaN / inf (degenerate box) -> 0
                px = (double)(float)px;                                      // the estimate is kept in fp32
                run += px + 0.25;
                cum[k] = run;
            }
        }
        std::vector<uint64_t> cuts((size_t)world + 1, 0);
        cuts[(size_t)world] = T;
        for (int r = 1; r < world; ++r) {
            const double target = T ? cum[(size_t)T - 1] * (double)r / (double)world : 0.0;
            cuts[(size_t)r] = (uint64_t)(std::lower_bound(cum.begin(), cum.end(), target) - cum.begin());
        }
        for (int r = 1; r <= world; ++r) cuts[(size_t)r] = std::max(cuts[(size_t)r], cuts[(size_t)r - 1]);
        for (int r = 0; r < world; ++r) { first[r] = cuts[(size_t)r]; count[r] = cuts[(size_t)r + 1] - cuts[(size_t)r]; }
        return M2S_OK;
    } catch (...) { g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
}

// The record exchange: every rank's block (counts[r] records of 96 bytes at d_mine on rank r) lands at record offset
// sum(counts[0..r)) of d_merged on every rank (root < 0) or on `root` only.  One RCCL group of exact-size sends and
// receives, enqueued on hip_stream (conversion stream: the records must be complete); the rank's own block is a device copy.
m2s_status m2s_dist_gather_records(m2s_dist* d, const void* d_mine, const uint64_t* counts, void* d_merged, int root, void* hip_stream) {
    if (!d || !counts) return M2S_ERR_INVALID;
    if (root >= d->world) { d->err = "root out of range"; return M2S_ERR_INVALID; }
    DCHK(d, hipSetDevice(d->device));
    hipStream_t st = (hipStream_t)hip_stream;
    const int W = d->world, me = d->rank;
    std::vector<uint64_t> off((size_t)W + 1, 0);
    for (int r = 0; r < W; ++r) off[r + 1] = off[r] + counts[r];
    const bool receives = root < 0 || root == me;
    if (receives && !d_merged && off[W]) { d->err = "d_merged is NULL"; return M2S_ERR_INVALID; }
    if (counts[me] && !d_mine) { d->err = "d_mine is NULL"; return M2S_ERR_INVALID; }
    const size_t rec = sizeof(m2s_gaussian);
    char* merged = static_cast<char*>(d_merged);
    if (receives && counts[me] && merged + off[me] * rec != d_mine)
        DCHK(d, hipMemcpyAsync(merged + off[me] * rec, d_mine, counts[me] * rec, hipMemcpyDeviceToDevice, st));
    if (W == 1) return M2S_OK;
    // message sizes are in bytes (ncclUint8); a block beyond 2^31 records still fits size_t
    // Operations on a communicator must be issued in the same order on every rank: first let the exchange thread issue every
    // counter exchange published so far (program order on each rank), then the record exchange.
    {
        std::unique_lock<std::mutex> l(d->q_lock);
        d->issued_cv.wait(l, [&] { return d->issued == d->published; });
    }
    std::lock_guard<std::mutex> comm_guard(d->comm_lock);
    std::vector<Xfer> sends, recvs;
    for (int step = 1; step < W; ++step) {
        const int dst = (me + step) % W, src = (me - step + W) % W;
        sends.push_back(Xfer{ dst, const_cast<void*>(d_mine), (counts[me] && (root < 0 || root == dst)) ? (size_t)(counts[me] * rec) : 0 });
        recvs.push_back(Xfer{ src, merged ? merged + off[src] * rec : nullptr, (counts[src] && receives) ? (size_t)(counts[src] * rec) : 0 });
    }
    return t_exchange(d, sends, recvs, st, d->err);
}

// Depth sort of records that are spread over the ranks (BASELINE config 5: "final radix sort of the merged splat buffer";
// the single-GPU pass is m2s_sort_by_depth == RadixSortPass.cpp:8-90).  Sample sort with ONE record exchange:
//   local sort -> 256 evenly spaced keys per rank, all-gathered -> world - 1 splitters (every rank picks the same) ->
//   rank j receives the keys in [splitter j-1, splitter j) from everybody (equal keys never straddle ranks), exact sizes,
//   all pairs at once -> local sort of what arrived (runs arrive in source-rank order, each sorted, and the local sort is
//   stable, so ties keep (source rank, original position) order).
// Concatenating the ranks' results in rank order IS the stable sort of the rank-major concatenation of the inputs, i.e. what
// one GPU produces from the merged buffer.  Keys are never sent: they are a function of the record (key kernel of m2s_sort.hip).
// Collective: every rank calls it, in the same order relative to the other m2s_dist_* calls.  Afterwards the context's
// current records are the received ones, its sorted buffer (m2s_device_sorted_records / m2s_download_sorted) holds this
// rank's slice of the sorted sequence: *out_n records starting at position *out_offset.
m2s_status m2s_dist_sort_by_depth(m2s_dist* d, m2s_ctx* ctx, const float world_to_view[16], uint64_t* out_n, uint64_t* out_offset) {
    if (!d || !ctx || !world_to_view) return M2S_ERR_INVALID;
    // A collective must be entered by EVERY rank or by none (RCCL peers of a rank that returned early would block forever in
    // ncclAllGather / Send / Recv; in-process peers would stall for kLocalBarrierSeconds and break the group).  So a rank
    // that fails locally — its own sort, a device call, the receive buffer — does not return: it carries on to the next
    // collective, contributes a non-zero STATUS word to it, and all ranks leave together right after.  Three
    // agreement points: the sample all-gather, the send-count all-gather, and a one-word all-gather just before the record
    // exchange (the only step whose failure — the receive buffer — is discovered after the counts are known).
    m2s_status mine = M2S_OK;                              // this rank's first failure
    auto note = [&](m2s_status st, const std::string& msg) { if (mine == M2S_OK) { mine = st; d->err = msg; } };
    auto hipok = [&](hipError_t e, const char* what) { if (e != hipSuccess) note(M2S_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };
    hipok(hipSetDevice(d->device), "hipSetDevice");
    uint64_t n = 0;
    if (mine == M2S_OK) {
        const m2s_status ls = m2s_sort_by_depth(ctx, world_to_view, &n);
        if (ls != M2S_OK) { note(ls, std::string("local sort: ") + m2s_last_error(ctx)); n = 0; }
    }
    const int W = d->world, me = d->rank;
    if (out_n) *out_n = n;
    if (out_offset) *out_offset = 0;
    if (W == 1) return mine;
    const uint32_t R = m2s_last_resolution(ctx);
    const uint32_t* keys = mine == M2S_OK ? static_cast<const uint32_t*>(m2s_device_sorted_keys(ctx)) : nullptr;
    const char* sorted = mine == M2S_OK ? static_cast<const char*>(m2s_device_sorted_records(ctx)) : nullptr;
    const size_t S = kSortSamples, S2 = S + 2 /* samples | how many are valid | status */, rec = sizeof(m2s_gaussian);
    const size_t W1 = (size_t)W + 1;                        /* send counts | status */
    unsigned long long *d_samples = d->d_sort, *d_all = d_samples + S2, *d_split = d_all + (size_t)W * S2, *d_bounds = d_split + W,
                       *d_sendc = d_bounds + W, *d_matrix = d_sendc + W1, *d_flag = d_matrix + (size_t)W * W1, *d_flags = d_flag + 1;
    unsigned long long *h_samples = d->h_sort, *h_all = h_samples + S2, *h_split = h_all + (size_t)W * S2, *h_bounds = h_split + W,
                       *h_sendc = h_bounds + W, *h_matrix = h_sendc + W1, *h_flag = h_matrix + (size_t)W * W1, *h_flags = h_flag + 1;
    {   // same order of communicator operations on every rank: counter exchanges published so far go first
        std::unique_lock<std::mutex> l(d->q_lock);
        d->issued_cv.wait(l, [&] { return d->issued == d->published; });
    }
    std::lock_guard<std::mutex> comm_guard(d->comm_lock);
    hipStream_t st = d->stream;
    // everybody's status words, gathered: all fine -> carry on; otherwise every rank returns here, with the same verdict
    auto agreed = [&](const unsigned long long* status_of_rank, size_t stride, const char* where) -> bool {
        int bad = -1;
        for (int r = 0; r < W; ++r) if (status_of_rank[(size_t)r * stride] != 0 && bad < 0) bad = r;
        if (bad < 0) return true;
        if (mine == M2S_OK) { mine = M2S_ERR_STATE; d->err = std::string("distributed sort abandoned ") + where + ": rank " + std::to_string(bad) + " reported a failure"; }
        return false;
    };
    try {
        // 1. samples of my sorted keys (+ my status), gathered
        if (mine == M2S_OK && n) {
            m2s::launch_pick_samples(keys, n, (uint32_t)S, d_samples, st);          // writes S samples and the number of valid ones
            h_flag[0] = 0;
            hipok(hipMemcpyAsync(d_samples + S + 1, h_flag, 8, hipMemcpyHostToDevice, st), "status upload");
        } else {
            for (size_t i = 0; i < S; ++i) h_samples[i] = ~0ull;
            h_samples[S] = 0;
            h_samples[S + 1] = mine == M2S_OK ? 0ull : 1ull;
            if (hipMemcpyAsync(d_samples, h_samples, S2 * 8, hipMemcpyHostToDevice, st) != hipSuccess) note(M2S_ERR_HIP, "sample upload failed");
        }
        {
            std::string terr;
            const m2s_status ts = t_all_gather(d, d_samples, d_all, S2, st, terr);
            if (ts != M2S_OK) { note(ts, terr); return mine; }                       // the transport itself failed: nothing left to agree through
        }
        if (!hipok(hipMemcpyAsync(h_all, d_all, (size_t)W * S2 * 8, hipMemcpyDeviceToHost, st), "samples download") ||
            !hipok(hipStreamSynchronize(st), "samples download")) {
            for (int r = 0; r < W; ++r) { h_all[(size_t)r * S2 + S] = 0; h_all[(size_t)r * S2 + S + 1] = 0; }   // unknown: my own status travels with the next gather
        }
        if (!agreed(h_all + S + 1, S2, "after the local sorts")) return mine;
        // 2. splitters: the same on every rank
        std::vector<unsigned long long> valid;
        for (int r = 0; r < W; ++r) {
            const size_t take = (size_t)std::min<unsigned long long>(h_all[(size_t)r * S2 + S], S);
            valid.insert(valid.end(), h_all + (size_t)r * S2, h_all + (size_t)r * S2 + take);
        }
        std::sort(valid.begin(), valid.end());
        const size_t m = valid.size();
        for (int j = 1; j < W; ++j) h_split[j - 1] = m ? valid[std::min((size_t)j * m / (size_t)W, m - 1)] : ~0ull;
        // 3. where my sorted block is cut: keys < splitter[0] stay with rank 0, [splitter[j-1], splitter[j]) go to rank j
        for (int j = 0; j < W; ++j) h_bounds[j] = 0;
        if (mine == M2S_OK && n) {
            if (hipok(hipMemcpyAsync(d_split, h_split, (size_t)(W - 1) * 8, hipMemcpyHostToDevice, st), "splitters upload")) {
                m2s::launch_lower_bounds(keys, n, d_split, (uint32_t)(W - 1), d_bounds, st);
                hipok(hipMemcpyAsync(h_bounds, d_bounds, (size_t)(W - 1) * 8, hipMemcpyDeviceToHost, st), "bounds download");
                hipok(hipStreamSynchronize(st), "bounds download");
            }
        }
        std::vector<uint64_t> edges((size_t)W + 1, 0);
        for (int j = 1; j < W; ++j) edges[(size_t)j] = std::max<uint64_t>(edges[(size_t)j - 1], std::min<uint64_t>(h_bounds[j - 1], n));
        edges[(size_t)W] = n;
        for (int j = 0; j < W; ++j) h_sendc[j] = mine == M2S_OK ? edges[(size_t)j + 1] - edges[(size_t)j] : 0;
        h_sendc[W] = mine == M2S_OK ? 0ull : 1ull;
        // 4. everybody's send counts (+ status): row r of the matrix = what rank r sends to each rank
        if (hipMemcpyAsync(d_sendc, h_sendc, W1 * 8, hipMemcpyHostToDevice, st) != hipSuccess) note(M2S_ERR_HIP, "send counts upload failed");
        {
            std::string terr;
            const m2s_status ts = t_all_gather(d, d_sendc, d_matrix, W1, st, terr);
            if (ts != M2S_OK) { note(ts, terr); return mine; }
        }
        if (!hipok(hipMemcpyAsync(h_matrix, d_matrix, (size_t)W * W1 * 8, hipMemcpyDeviceToHost, st), "matrix download") ||
            !hipok(hipStreamSynchronize(st), "matrix download"))
            for (size_t i = 0; i < (size_t)W * W1; ++i) h_matrix[i] = 0;
        const bool ok4 = agreed(h_matrix + W, W1, "after the split");
        std::vector<uint64_t> roff((size_t)W + 1, 0);
        uint64_t offset = 0;
        for (int r = 0; r < W; ++r) roff[(size_t)r + 1] = roff[(size_t)r] + h_matrix[(size_t)r * W1 + me];
        for (int q = 0; q < me; ++q)
            for (int r = 0; r < W; ++r) offset += h_matrix[(size_t)r * W1 + q];
        const uint64_t total_recv = roff[(size_t)W];
        if (!ok4) return mine;
        // 5. the receive buffer (the context's record pool; my sorted block stays where it is: a separate buffer) — the one
        //    step that can fail after the counts are known: agree once more, on one word, before anybody sends
        void* pool = nullptr;
        if (mine == M2S_OK) {
            const m2s_status rs = m2s_reserve_records(ctx, total_recv, &pool);
            if (rs != M2S_OK) note(rs, std::string("receive buffer: ") + m2s_last_error(ctx));
        }
        h_flag[0] = mine == M2S_OK ? 0ull : 1ull;
        if (hipMemcpyAsync(d_flag, h_flag, 8, hipMemcpyHostToDevice, st) != hipSuccess) note(M2S_ERR_HIP, "status upload failed");
        {
            std::string terr;
            const m2s_status ts = t_all_gather(d, d_flag, d_flags, 1, st, terr);
            if (ts != M2S_OK) { note(ts, terr); return mine; }
        }
        if (!hipok(hipMemcpyAsync(h_flags, d_flags, (size_t)W * 8, hipMemcpyDeviceToHost, st), "status download") ||
            !hipok(hipStreamSynchronize(st), "status download"))
            for (int r = 0; r < W; ++r) h_flags[r] = 1;   // cannot read the verdict: do not send (peers that can read it see my next status... there is none: report)
        if (!agreed(h_flags, 1, "before the record exchange")) return mine;
        char* dst = static_cast<char*>(pool);
        if (h_sendc[me]) hipok(hipMemcpyAsync(dst + roff[(size_t)me] * rec, sorted + edges[(size_t)me] * rec, h_sendc[me] * rec, hipMemcpyDeviceToDevice, st), "own block");
        std::vector<Xfer> sends, recvs;
        for (int step = 1; step < W; ++step) {
            const int to = (me + step) % W, from = (me - step + W) % W;
            sends.push_back(Xfer{ to, const_cast<char*>(sorted) + edges[(size_t)to] * rec, (size_t)(h_sendc[to] * rec) });
            recvs.push_back(Xfer{ from, dst + roff[(size_t)from] * rec, (size_t)(h_matrix[(size_t)from * W1 + me] * rec) });
        }
        {
            std::string terr;
            const m2s_status ts = t_exchange(d, sends, recvs, st, terr);
            if (ts != M2S_OK) { note(ts, terr); return mine; }
        }
        if (!hipok(hipStreamSynchronize(st), "record exchange")) return mine;
        if (mine != M2S_OK) return mine;
        // 6. what arrived, sorted (stable): my slice of the global order.  Local from here on: no collective follows.
        m2s_status fs;
        if ((fs = m2s_set_records(ctx, pool, total_recv, R)) != M2S_OK) { d->err = m2s_last_error(ctx); return fs; }
        uint64_t n2 = 0;
        if ((fs = m2s_sort_by_depth(ctx, world_to_view, &n2)) != M2S_OK) { d->err = std::string("final sort: ") + m2s_last_error(ctx); return fs; }
        if (out_n) *out_n = total_recv;
        if (out_offset) *out_offset = offset;
        return M2S_OK;
    } catch (...) { d->err = "host allocation failed"; return M2S_ERR_OOM; }
}

// Blocks until everything enqueued on hip_stream (e.g. the record exchange) has completed.
m2s_status m2s_dist_wait(m2s_dist* d, void* hip_stream) {
    if (!d) return M2S_ERR_INVALID;
    DCHK(d, hipSetDevice(d->device));
    DCHK(d, hipStreamSynchronize((hipStream_t)hip_stream));
    return M2S_OK;
}

}  // extern "C"
