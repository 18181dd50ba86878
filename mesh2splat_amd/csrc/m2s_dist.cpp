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
// librccl is opened at run time (dlopen), not linked: a Python process that already carries PyTorch's bundled RCCL must
// not end up with a second copy, and single-GPU users of libm2s_hip.so need no RCCL at all.
#include "../../include/m2s.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <thread>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

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
};
Rccl g_rccl;
std::once_flag g_rccl_once;
thread_local std::string g_dist_error;

void load_rccl() {
    Rccl& r = g_rccl;
    // an RCCL that is already in the process (PyTorch's) first, then the system one
    const char* names[] = { "librccl.so.1", "librccl.so" };
    for (const char* n : names) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
    if (const char* e = std::getenv("M2S_RCCL_PATH")) if (!r.handle) r.handle = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    const char* paths[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so" };
    for (const char* n : paths) if (!r.handle) r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.handle) { r.error = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
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

}  // namespace

struct m2s_dist {
    int device = 0, rank = 0, world = 1;
    ncclComm_p comm = nullptr;
    hipStream_t stream = nullptr;              // counter exchanges run here, off the conversion stream
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
            int r = kNcclSuccess;
            if (e == hipSuccess) r = g_rccl.AllGather(d->d_mine + k, d->d_all + (size_t)k * d->world, 1, kNcclUint64, d->comm, d->stream);
            if (e == hipSuccess && r == kNcclSuccess)
                e = hipMemcpyAsync(d->h_all + (size_t)k * d->world, d->d_all + (size_t)k * d->world, (size_t)d->world * 8, hipMemcpyDeviceToHost, d->stream);
            if (e == hipSuccess && r == kNcclSuccess) e = hipEventRecord(d->done[k], d->stream);
            if (r != kNcclSuccess) err = std::string("ncclAllGather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
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

m2s_status m2s_dist_create(int device, const uint8_t id[M2S_DIST_ID_BYTES], int rank, int world, m2s_dist** out) {
    if (!out || !id || world < 1 || rank < 0 || rank >= world) { g_dist_error = "bad argument"; return M2S_ERR_INVALID; }
    *out = nullptr;
    if (!rccl_ready()) return M2S_ERR_STATE;
    m2s_dist* d = new (std::nothrow) m2s_dist();
    if (!d) { g_dist_error = "host allocation failed"; return M2S_ERR_OOM; }
    d->device = device; d->rank = rank; d->world = world;
    auto bail = [&](m2s_status s) { g_dist_error = d->err; m2s_dist_destroy(d); return s; };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) { d->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return bail(M2S_ERR_NO_DEVICE); }
    ncclUniqueId_t uid;
    memcpy(&uid, id, sizeof uid);
    const int r = g_rccl.CommInitRank(&d->comm, world, uid, rank);
    if (r != kNcclSuccess) { d->err = std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(r); d->comm = nullptr; return bail(M2S_ERR_HIP); }
    if ((e = hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_mine, kRing * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipMalloc((void**)&d->d_all, (size_t)kRing * world * sizeof(unsigned long long))) != hipSuccess ||
        (e = hipHostMalloc((void**)&d->h_mine, kRing * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess ||
        (e = hipHostMalloc((void**)&d->h_all, (size_t)kRing * world * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess) {
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
    if (d->stream) (void)hipStreamDestroy(d->stream);
    delete d;
}

int m2s_dist_rank(const m2s_dist* d) { return d ? d->rank : -1; }
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
    NCHK(d, g_rccl.GroupStart());
    for (int step = 1; step < W; ++step) {
        const int dst = (me + step) % W, src = (me - step + W) % W;
        if (counts[me] && (root < 0 || root == dst)) NCHK(d, g_rccl.Send(d_mine, counts[me] * rec, kNcclUint8, dst, d->comm, st));
        if (counts[src] && receives) NCHK(d, g_rccl.Recv(merged + off[src] * rec, counts[src] * rec, kNcclUint8, src, d->comm, st));
    }
    NCHK(d, g_rccl.GroupEnd());
    return M2S_OK;
}

// Blocks until everything enqueued on hip_stream (e.g. the record exchange) has completed.
m2s_status m2s_dist_wait(m2s_dist* d, void* hip_stream) {
    if (!d) return M2S_ERR_INVALID;
    DCHK(d, hipSetDevice(d->device));
    DCHK(d, hipStreamSynchronize((hipStream_t)hip_stream));
    return M2S_OK;
}

}  // extern "C"
