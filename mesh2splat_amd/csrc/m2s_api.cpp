// m2s_api.cpp — host side of the C ABI (include/m2s.h): context, scene upload, the conversion
// pass driver (== ConversionPass::execute, src/renderer/renderPasses/ConversionPass.cpp:9-68) and
// read-back.  Compiled with hipcc; no CPU compute path exists here.
#include "../../include/m2s.h"
#include <cstdlib>
#include "m2s_device.h"
#include "m2s_ply.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

using namespace m2s;

namespace {
thread_local std::string g_create_error = "";

constexpr uint32_t kMaxGaussiansToSort = 7000000u;  // RenderPass.hpp:9
constexpr uint64_t kMaxTriangles = (1ull << 28) - 1;  // 32-bit byte offsets into the 16 B/triangle planes

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
const char* const kStaleMsg = "the records of the conversion last waited for have been overwritten by a later submission at another R "
                              "(wait for it, or submit into your own buffers)";
}  // namespace

constexpr int kBandSlotsMax = 64;   // (scene, R) entries remembered per context: band tables, decisions
struct m2s_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    void* tri_mem = nullptr;
    SceneDev scene{};
    MeshParams* d_meshes = nullptr;
    uint32_t* d_mesh_first = nullptr;
    uint32_t n_meshes_total = 0;
    bool has_scene = false;
    uint64_t range_first = 0, range_count = UINT64_MAX;

    // work buffers (sized by the scene)
    uint32_t* d_cnt = nullptr;
    uint32_t* d_off = nullptr;
    uint32_t* d_partials = nullptr;
    uint32_t* d_start = nullptr;
    size_t start_cap = 0;
    unsigned long long* d_total = nullptr;
    unsigned long long* h_total = nullptr;  // pinned: [0] = fragment counter, [1] = status words of the fused kernel
    unsigned long long* d_chain = nullptr;  // look-back chain of the fused kernel, one word per wave
    BigItem* d_biglist = nullptr;           // triangles deferred by the fused kernel (capacity: triangles in range)
    uint32_t* d_bigmeta = nullptr;          // [0] entries in d_biglist, [1] largest, [2] total fragment count; zero between conversions
    // What the context remembers about the uploaded scene at a given resolution R.  The reference converts on load and
    // whenever the density slider moves (guiRendererConcreteMediator.cpp:51-57), i.e. mostly at an R it has not seen
    // before, so nothing here may be REQUIRED for a fast conversion: a new R costs no counting pass and no extra host
    // round trip (the AUTO decision is taken from frag_per_R2, the band bases are a by-product of the first launch).
    struct RInfo {
        bool decided = false;      // AUTO: single-pass / multi-pass decision taken
        bool multipass = false;    // ... and it was "multi-pass"
        bool sparse = false;       // AUTO: fewer fragments than triangles, the sparse form of the single-pass kernel (k_sparse)
        bool sparse_off = false;   // k_sparse reported a workgroup that did not fit its LDS stream: use k_fused2
        bool team_off = false;     // k_fused2 reported a workgroup that did not fit its LDS stream: use k_fused
        bool async_ok = false;     // a completed conversion needed no host decision between kernels
        bool mp_ready = false;     // a multi-pass conversion has completed (its work buffers are sized)
        bool bands_ready = false;  // d_bands[band_slot] holds the XCD band table (cuts + bases) for this R ...
        uint32_t bands_unit = 0;   // ... in workgroups of this many triangles (256: cut from a k_fused2 launch, 512: k_sparse)
        uint32_t band_width = 0;   // ... whose widest band has this many workgroups (the grid of a banded launch is 8 x this)
        int band_slot = 0;
        uint32_t gen = 0;          // generation of the table this entry belongs to (the table starts over when it is full)
    };
    uint32_t rinfo_gen = 0;
    std::map<uint32_t, RInfo> rinfo;
    double frag_per_R2 = -1.0;              // fragments / R^2 of this scene, learned from its first conversion (any R)
    unsigned long long* d_bands = nullptr;  // kBandSlots x kBandTableWords: XCD band tables (device)
    unsigned long long* h_bands = nullptr;  // kBandSlots x 9 (pinned): the cuts of each table, written by k_pick_bands itself
    unsigned long long* d_wg_base = nullptr;   // where every workgroup's output started in the newest launch without bands
    uint32_t* d_batch_first = nullptr;      // work-balanced batches of k_fused2 (small scenes; built from the first exact count)
    uint32_t n_batch_tab = 0;               // batches in it (0: uniform batches)
    size_t chain_words = 0;                 // words of d_chain (and of the second lane's chain)
    void* d_setup = nullptr;                // multi-pass pipeline: per-triangle TriSetup records (allocated at its first use)
    int last_pipeline = 0;                  // what the last conversion ran (m2s_last_pipeline)
    // second lane for context-owned asynchronous submissions: odd slots run on their own stream with their own chain
    // and record buffer, so that consecutive single-kernel conversions overlap (the tail of one, where the GPU drains,
    // with the head of the next) instead of paying ~8 us between dependent kernels on one stream
    int lanes = 1;                          // m2s_set_async_lanes
    hipStream_t stream_b = nullptr;
    unsigned long long* d_chain_b = nullptr;
    void* d_records_b = nullptr;
    uint64_t records_b_cap = 0;
    int pipeline = M2S_PIPELINE_AUTO;
    uint32_t epoch = 0;                     // launch counter of the fused kernel (tags the chain words)

    // asynchronous submissions (m2s_convert_submit / m2s_convert_wait): a ring of result slots.  Slot k uses
    // h_total[2 + 2k] (counter) and h_total[3 + 2k] (status words), written by the kernel itself.
    struct Slot { hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr; uint64_t limit = 0; void* d_out = nullptr; uint32_t R = 0;
                  bool sync_result = false; uint64_t sync_total = 0; bool prof = false; float ms[M2S_K_N] = {};
                  int own_lane = -1; uint32_t gen = 0;   // context-owned buffer (0 / 1) and its generation at submission; -1: caller's buffer
                  bool wrote_bands = false; uint32_t bands_unit = 0; uint32_t ri_gen = 0;   // the launch leaves band bases behind, for the RInfo entry of that table generation
                  hipStream_t st = nullptr; bool shared_work = false; };   // stream it ran on; did it use the context's shared work buffers (chain, deferred list)?
    Slot slot[M2S_MAX_IN_FLIGHT];
    uint32_t slot_head = 0, slot_count = 0; // oldest in-flight slot, number in flight
    hipStream_t last_submit_stream = nullptr;   // stream of the newest in-flight submission (work buffers are shared: see submit)
    uint32_t buf_R[2] = { 0, 0 };           // context-owned record buffers (lane a / b): R of the newest conversion enqueued into it ...
    uint32_t buf_gen[2] = { 0, 0 };         // ... and a generation that advances whenever that R changes
    bool records_stale = false;             // the conversion last waited for has been overwritten by a later submission

    // output
    void* d_records = nullptr;
    uint64_t records_cap = 0;  // records
    const void* last_records = nullptr;
    int64_t cap_policy = -1;
    uint64_t last_total = 0, last_stored = 0;
    uint32_t last_R = 0;

    // depth sort (f-2)
    void* d_sorted = nullptr;
    uint64_t sorted_cap = 0, sorted_n = 0;
    uint32_t* d_sort_u32 = nullptr;   // keys_in | vals_in | keys_out | vals_out
    void* d_sort_temp = nullptr;
    size_t sort_temp_cap = 0;
    uint64_t sort_u32_cap = 0;
    float last_sort_ms = 0.0f;
    // viewer prepass (m2s_prepass): survivors, their depths, the look-back chain of its kernel, a copy of the depth image
    void* d_quads = nullptr;
    float* d_pp_depths = nullptr;
    uint64_t pp_cap = 0, pp_visible = 0;
    unsigned long long* d_pp_chain = nullptr;
    uint64_t pp_chain_words = 0;
    uint32_t pp_epoch = 0;
    float* d_pp_depthtex = nullptr;
    uint64_t pp_depthtex_cap = 0;
    float last_prepass_ms = 0.0f;
    m2s_gaussian* h_export[2] = { nullptr, nullptr };   // pinned chunk buffers of m2s_export_ply
    // upload staging: two pinned host chunks (filled by a few host threads while the previous chunk is on the bus) and
    // two device chunks for the AoS -> SoA repack; allocated at the first upload, kept
    void* h_stage[2] = { nullptr, nullptr };
    void* d_stage[2] = { nullptr, nullptr };
    hipEvent_t stage_ev[2] = { nullptr, nullptr };
    void* scene_arena = nullptr;             // one allocation for mesh table, textures, combo textures and work buffers
    float last_upload_ms[4] = { 0, 0, 0, 0 }; // [0] total, [1] geometry, [2] textures + mips + combo, [3] allocations
    void* d_rows = nullptr;                  // m2s_export_ply: .ply rows encoded on the device (formats 1 and 2)
    uint64_t rows_cap = 0;                   // bytes
    void* d_loaded = nullptr;                // m2s_upload_records (a loaded .ply)
    uint64_t loaded_cap = 0;
    void* d_sorted_quads = nullptr;          // m2s_sort_prepass
    uint64_t sq_cap = 0, sq_n = 0;
    float last_sort_prepass_ms = 0.0f;

    // measurement
    bool profiling = false;
    hipEvent_t ev[8] = {};
    float last_ms[M2S_K_N] = {};
};

#define HIPCHK(ctx, call)                                                                         \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                       \
            return e_ == hipErrorOutOfMemory ? M2S_ERR_OOM : M2S_ERR_HIP;                         \
        }                                                                                         \
    } while (0)

static m2s_status fail(m2s_ctx* c, m2s_status s, const std::string& msg) {
    if (c) c->err = msg;
    return s;
}

static void free_scene(m2s_ctx* c) {
    if (c->tri_mem) (void)hipFree(c->tri_mem);
    if (c->scene_arena) (void)hipFree(c->scene_arena);
    if (c->d_chain_b) (void)hipFree(c->d_chain_b);
    if (c->d_setup) (void)hipFree(c->d_setup);
    c->d_chain_b = nullptr; c->d_setup = nullptr;
    c->tri_mem = nullptr; c->scene_arena = nullptr;
    // everything below lived inside the arena
    c->d_meshes = nullptr; c->d_mesh_first = nullptr;
    c->d_cnt = c->d_off = c->d_partials = nullptr;
    c->d_chain = nullptr; c->d_biglist = nullptr; c->d_bigmeta = nullptr; c->d_bands = nullptr; c->d_wg_base = nullptr;
    c->d_batch_first = nullptr; c->n_batch_tab = 0; c->chain_words = 0;
    c->rinfo.clear();
    ++c->rinfo_gen;
    c->frag_per_R2 = -1.0;
    c->scene = SceneDev{};
    c->has_scene = false;
}

// What is remembered about this scene at resolution R (created on first use; the table is bounded: a slider dragged
// through hundreds of densities simply starts over).
constexpr int kBandSlots = kBandSlotsMax;
// AUTO: below this many fragments per triangle the sparse kernel runs.  Measured against k_fused2 on cube-spheres
// (tools/sparse_crossover.py, profiles/r03/v3_sparse_crossover_*): with 3 M and 6.2 M triangles k_sparse is ahead up to 1.75
// fragments per triangle (a workgroup's stream overflows from ~2.5); with 1 M triangles — 2.5 generations of its 512-triangle
// workgroups — k_fused2 is ahead down to 0.68 at least.
static double sparse_frags_per_triangle(uint32_t n_tri) { return n_tri >= 2000000u ? 1.75 : 0.5; }
static m2s_ctx::RInfo& rinfo_for(m2s_ctx* c, uint32_t R) {
    auto it = c->rinfo.find(R);
    if (it != c->rinfo.end()) return it->second;
    // (a full table starts over; submissions still in flight remember the generation they were made under, so that a band
    //  slot which now belongs to another density is never marked ready on their behalf: m2s_convert_wait)
    if (c->rinfo.size() >= (size_t)kBandSlots) { c->rinfo.clear(); ++c->rinfo_gen; }
    m2s_ctx::RInfo ri;
    ri.gen = c->rinfo_gen;
    ri.band_slot = (int)c->rinfo.size();
    return c->rinfo.emplace(R, ri).first->second;
}

extern "C" {

uint32_t m2s_abi_version(void) { return M2S_ABI_VERSION; }

const char* m2s_last_error(const m2s_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

m2s_status m2s_create(int device, m2s_ctx** out_ctx) {
    if (!out_ctx) { g_create_error = "out_ctx is NULL"; return M2S_ERR_INVALID; }
    *out_ctx = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        g_create_error = std::string("no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count=0") +
                         "); this library has no CPU path";
        return M2S_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) { g_create_error = "device index out of range"; return M2S_ERR_NO_DEVICE; }
    m2s_ctx* c = new (std::nothrow) m2s_ctx();
    if (!c) { g_create_error = "host allocation failed"; return M2S_ERR_OOM; }
    c->device = device;
    auto bail = [&](const char* what, hipError_t he) {
        g_create_error = std::string(what) + ": " + hipGetErrorString(he);
        delete c;
        return M2S_ERR_HIP;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipMalloc(&c->d_total, sizeof(unsigned long long))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc((void**)&c->h_total, (4 + 2 * M2S_MAX_IN_FLIGHT) * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    if ((e = hipHostMalloc((void**)&c->h_bands, (size_t)kBandSlotsMax * 9 * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    memset(c->h_bands, 0, (size_t)kBandSlotsMax * 9 * sizeof(unsigned long long));
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    for (auto& ev : c->stage_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    unsigned done_flags = hipEventDisableTiming;   // (completion only: its time is never asked for)
    if (const char* v = std::getenv("M2S_DONE_EVENT_FLAGS")) done_flags = (unsigned)strtoul(v, nullptr, 0);   // debug: A/B of event kinds
    for (auto& sl : c->slot)
        if ((e = hipEventCreateWithFlags(&sl.done, done_flags)) != hipSuccess || (e = hipEventCreate(&sl.t0)) != hipSuccess ||
            (e = hipEventCreate(&sl.t1)) != hipSuccess)
            return bail("hipEventCreate", e);
    *out_ctx = c;
    return M2S_OK;
}

// every conversion still in flight has finished when this returns (their slots stay queued for m2s_convert_wait)
static void drain_in_flight(m2s_ctx* c) {
    for (uint32_t k = 0; k < c->slot_count; ++k) {
        auto& sl = c->slot[(c->slot_head + k) % M2S_MAX_IN_FLIGHT];
        if (!sl.sync_result) (void)hipEventSynchronize(sl.done);
    }
}

void m2s_destroy(m2s_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drain_in_flight(c);   // conversions still in flight on a caller's stream
    free_scene(c);
    if (c->d_start) (void)hipFree(c->d_start);
    if (c->d_records) (void)hipFree(c->d_records);
    if (c->d_records_b) (void)hipFree(c->d_records_b);
    if (c->stream_b) { (void)hipStreamSynchronize(c->stream_b); (void)hipStreamDestroy(c->stream_b); }
    if (c->d_sorted) (void)hipFree(c->d_sorted);
    if (c->d_quads) (void)hipFree(c->d_quads);
    if (c->d_sorted_quads) (void)hipFree(c->d_sorted_quads);
    if (c->d_loaded) (void)hipFree(c->d_loaded);
    if (c->d_rows) (void)hipFree(c->d_rows);
    for (int k = 0; k < 2; ++k) if (c->h_export[k]) (void)hipHostFree(c->h_export[k]);
    for (int k = 0; k < 2; ++k) {
        if (c->h_stage[k]) (void)hipHostFree(c->h_stage[k]);
        if (c->d_stage[k]) (void)hipFree(c->d_stage[k]);
        if (c->stage_ev[k]) (void)hipEventDestroy(c->stage_ev[k]);
    }
    if (c->d_pp_depths) (void)hipFree(c->d_pp_depths);
    if (c->d_pp_chain) (void)hipFree(c->d_pp_chain);
    if (c->d_pp_depthtex) (void)hipFree(c->d_pp_depthtex);
    if (c->d_sort_u32) (void)hipFree(c->d_sort_u32);
    if (c->d_sort_temp) (void)hipFree(c->d_sort_temp);
    if (c->d_total) (void)hipFree(c->d_total);
    if (c->h_total) (void)hipHostFree(c->h_total);
    if (c->h_bands) (void)hipHostFree(c->h_bands);
    for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
    for (auto& sl : c->slot) {
        if (sl.done) (void)hipEventDestroy(sl.done);
        if (sl.t0) (void)hipEventDestroy(sl.t0);
        if (sl.t1) (void)hipEventDestroy(sl.t1);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

m2s_status m2s_set_triangle_range(m2s_ctx* c, uint64_t first, uint64_t count) {
    if (!c) return M2S_ERR_INVALID;
    c->range_first = first;
    c->range_count = count;
    return M2S_OK;
}

m2s_status m2s_set_max_gaussians(m2s_ctx* c, int64_t cap) {
    if (!c) return M2S_ERR_INVALID;
    if (cap < -1) return fail(c, M2S_ERR_INVALID, "cap must be -1 (reference formula), 0 (unlimited) or > 0");
    c->cap_policy = cap;
    return M2S_OK;
}

}  // extern "C"

// ---- host -> device through pinned staging ---------------------------------------------------------------------------
// The caller's buffers are ordinary pageable memory (std::vector in the reference, SceneManager.cpp:483-512): a plain
// hipMemcpy from them runs at 1-2 GB/s on this platform (round 1: 86-232 ms for the 254 MB of the C3 scene).  Here a few
// host threads copy a 16 MiB chunk into one of two PINNED buffers while the DMA engine moves the other one, so the bus,
// not the page-by-page staging inside the runtime, sets the pace.
namespace {
constexpr size_t kStageChunk = 16ull << 20;

void par_memcpy(void* dst, const void* src, size_t n) {
    const size_t kMin = 2ull << 20;
    unsigned nt = (unsigned)std::min<size_t>(4, n / kMin);
    if (const char* e = std::getenv("M2S_HOST_THREADS")) { if (std::atol(e) == 1) nt = 1; }
    if (nt <= 1) { memcpy(dst, src, n); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < nt; ++i) {
        const size_t b = n * i / nt, e = n * (i + 1) / nt;
        pool.emplace_back([=] { memcpy((char*)dst + b, (const char*)src + b, e - b); });
    }
    memcpy(dst, src, n / nt);
    for (auto& t : pool) t.join();
}

m2s_status ensure_stage(m2s_ctx* c) {
    for (int k = 0; k < 2; ++k) {
        if (!c->h_stage[k]) HIPCHK(c, hipHostMalloc(&c->h_stage[k], kStageChunk, hipHostMallocDefault));
        if (!c->d_stage[k]) HIPCHK(c, hipMalloc(&c->d_stage[k], kStageChunk));
    }
    return M2S_OK;
}

// Moves `bytes` from pageable `src` to the device in chunks of at most `chunk` bytes (<= kStageChunk).  Chunk i lands in
// dst + offset (dst != nullptr) or in the device staging buffer d_stage[i & 1] (dst == nullptr); then on_chunk(device
// pointer of the chunk, offset, bytes of the chunk) may enqueue work that consumes it on c->stream.
template <class F>
m2s_status staged_h2d(m2s_ctx* c, const char* src, size_t bytes, size_t chunk, char* dst, uint32_t& turn, F on_chunk) {
    for (size_t off = 0; off < bytes; off += chunk) {
        const size_t n = std::min(chunk, bytes - off);
        const int k = (int)(turn++ & 1u);
        HIPCHK(c, hipEventSynchronize(c->stage_ev[k]));          // the DMA that last read h_stage[k] has finished
        par_memcpy(c->h_stage[k], src + off, n);
        char* d = dst ? dst + off : (char*)c->d_stage[k];        // d_stage[k]: its previous consumer precedes us on the stream
        HIPCHK(c, hipMemcpyAsync(d, c->h_stage[k], n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipEventRecord(c->stage_ev[k], c->stream));
        on_chunk(d, off, n);
    }
    return M2S_OK;
}
}  // namespace

extern "C" {

m2s_status m2s_upload_scene(m2s_ctx* c, const m2s_mesh* meshes, uint32_t n_meshes) {
    if (!c) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (n_meshes > 0xFFFFFFu) return fail(c, M2S_ERR_INVALID, "more than 2^24-1 meshes");   // TriShade keeps the index in 24 bits
    if (n_meshes && !meshes) return fail(c, M2S_ERR_INVALID, "meshes is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    // ---- validate + global triangle index space -------------------------------------------------
    std::vector<uint32_t> mesh_first(n_meshes + 1, 0);
    uint64_t T = 0;
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const m2s_mesh& m = meshes[i];
        if (m.stride_floats < 12) return fail(c, M2S_ERR_INVALID, "stride_floats must be >= 12");
        if (m.stride_floats > 4096) return fail(c, M2S_ERR_INVALID, "stride_floats must be <= 4096");
        if (m.n_vertices % 3) return fail(c, M2S_ERR_INVALID, "n_vertices must be a multiple of 3");
        if (m.n_vertices && !m.vertices) return fail(c, M2S_ERR_INVALID, "vertices is NULL");
        for (int k = 0; k < 3; ++k)
            if (m.tex[k].rgba8 && (!m.tex[k].width || !m.tex[k].height || m.tex[k].width > 32768 || m.tex[k].height > 32768))
                return fail(c, M2S_ERR_INVALID, "texture dimensions must be in [1, 32768]");
        mesh_first[i] = (uint32_t)T;
        T += m.n_vertices / 3;
        if (T > kMaxTriangles) return fail(c, M2S_ERR_INVALID, "more than 2^28-1 triangles in one scene (shard it with m2s_set_triangle_range per context)");
    }
    mesh_first[n_meshes] = (uint32_t)T;
    const uint64_t first = std::min<uint64_t>(c->range_first, T);
    const uint64_t last = (c->range_count == UINT64_MAX || c->range_count > T - first) ? T : first + c->range_count;
    const uint32_t n_tri = (uint32_t)(last - first);

    free_scene(c);
    c->n_meshes_total = n_meshes;
    c->scene.n_meshes = n_meshes;
    c->scene.n_tri = n_tri;
    c->scene.tri_first = (uint32_t)first;
    c->last_total = c->last_stored = 0;
    c->last_records = nullptr;
    c->records_stale = false;

    // ---- layout: geometry planes (144 B / triangle) in one allocation, everything else in a second one -------------
    const auto t_alloc = std::chrono::steady_clock::now();
    const size_t np = std::max<size_t>(n_tri, 1);
    size_t offs[11], cur = 0;
    const size_t widths[11] = { 16, 16, 4, 16, 8, 16, 16, 4, 16, 16, 16 };
    for (int k = 0; k < 11; ++k) { offs[k] = cur; cur = align_up(cur + np * widths[k], 256); }
    HIPCHK(c, hipMalloc(&c->tri_mem, cur));

    // textures (deduplicated by host pointer), their mip chains, combo chains: sizes first
    struct TexPlan { const uint8_t* src; TexDesc d; size_t arena_off; };
    struct ComboPlan { ComboDesc d; size_t arena_off; uint32_t ia, in, im; };
    std::vector<TexPlan> tex_plan;
    std::vector<ComboPlan> combo_plan;
    std::map<std::tuple<const uint8_t*, uint32_t, uint32_t>, uint32_t> dedup;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t>, uint32_t> cdedup;
    std::vector<std::array<int, 3>> mesh_tex(std::max<uint32_t>(n_meshes, 1), std::array<int, 3>{ -1, -1, -1 });
    std::vector<int> mesh_combo(std::max<uint32_t>(n_meshes, 1), -1);
    size_t arena = 0;
    auto take = [&](size_t bytes) { const size_t o = arena; arena = align_up(arena + std::max<size_t>(bytes, 4), 256); return o; };
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const m2s_mesh& m = meshes[i];
        for (int k = 0; k < 3; ++k) {
            const m2s_texture& t = m.tex[k];
            if (!t.rgba8) continue;
            const auto key = std::make_tuple(t.rgba8, t.width, t.height);
            auto it = dedup.find(key);
            if (it != dedup.end()) { mesh_tex[i][k] = (int)it->second; continue; }
            TexPlan p{};
            p.src = t.rgba8;
            p.d.w = t.width; p.d.h = t.height;
            uint32_t mx = std::max(t.width, t.height), nl = 1;
            while (mx > 1 && nl < 5) { mx >>= 1; nl++; }
            p.d.n_levels = nl;
            size_t tot = 0;
            for (uint32_t l = 0; l < nl; ++l) {
                p.d.off[l] = (uint32_t)tot;
                tot += (size_t)std::max(1u, t.width >> l) * std::max(1u, t.height >> l);
            }
            p.arena_off = take(tot * 4);
            dedup[key] = (uint32_t)tex_plan.size();
            mesh_tex[i][k] = (int)tex_plan.size();
            tex_plan.push_back(p);
        }
        // combo texture (interleaved albedo / normal / MR, see ComboDesc): all three maps present, same size
        const int ia = mesh_tex[i][0], in = mesh_tex[i][1], im = mesh_tex[i][2];
        if (ia < 0 || in < 0 || im < 0) continue;
        const TexDesc &ta = tex_plan[ia].d, &tn = tex_plan[in].d, &tm = tex_plan[im].d;
        if (ta.w != tn.w || ta.w != tm.w || ta.h != tn.h || ta.h != tm.h) continue;
        const auto ckey = std::make_tuple((uint32_t)ia, (uint32_t)in, (uint32_t)im);
        auto cit = cdedup.find(ckey);
        if (cit != cdedup.end()) { mesh_combo[i] = (int)cit->second; continue; }
        ComboPlan cp{};
        size_t tot = 0;
        for (uint32_t l = 0; l < ta.n_levels; ++l) {
            cp.d.coff[l] = (uint32_t)tot;
            tot += (size_t)(std::max(1u, ta.w >> l) + 1) * std::max(1u, ta.h >> l) * 3;
        }
        if (tot > 0x3FFFFFF0ull) continue;  // the sampler addresses the combo texels with 32-bit BYTE offsets
        cp.arena_off = take(tot * 4);
        cp.ia = (uint32_t)ia; cp.in = (uint32_t)in; cp.im = (uint32_t)im;
        cdedup[ckey] = (uint32_t)combo_plan.size();
        mesh_combo[i] = (int)combo_plan.size();
        combo_plan.push_back(cp);
    }
    const size_t n_mp = std::max<uint32_t>(n_meshes, 1);
    const size_t chain_words = std::max<size_t>(std::max<size_t>(n_fused_waves(n_tri), batch_table_capacity(n_tri)), 1);
    const size_t o_meshes = take(n_mp * sizeof(MeshParams));
    const size_t o_mesh_first = take(mesh_first.size() * sizeof(uint32_t));
    const size_t o_mesh_of8 = take(((np + 7) / 8 + 1) * sizeof(uint2));
    const size_t o_cnt = take(np * sizeof(uint32_t));
    const size_t o_off = take((np + 1) * sizeof(uint32_t));
    const size_t o_partials = take(std::max<size_t>(n_count_blocks(n_tri), 1) * sizeof(uint32_t));
    const size_t o_chain = take(chain_words * sizeof(unsigned long long));
    const size_t o_biglist = take(np * sizeof(BigItem));
    const size_t o_bigmeta = take(4 * sizeof(uint32_t));
    const size_t o_bands = take((size_t)kBandSlots * kBandTableWords * sizeof(unsigned long long));
    const size_t o_wg_base = take(((size_t)n_fused_waves(n_tri) / 4 + 2) * sizeof(unsigned long long));
    const size_t o_batch = take(std::max<size_t>(batch_table_capacity(n_tri), 1) * sizeof(uint32_t));
    HIPCHK(c, hipMalloc(&c->scene_arena, arena));
    { const m2s_status s = ensure_stage(c); if (s != M2S_OK) return s; }
    c->last_upload_ms[3] = ms_since(t_alloc);
    char* A = (char*)c->scene_arena;
    char* b = (char*)c->tri_mem;
    TriPlanes& tp = c->scene.tri;
    tp.A0 = (const float4*)(b + offs[0]); tp.A1 = (const float4*)(b + offs[1]); tp.A2 = (const float*)(b + offs[2]);
    tp.B0 = (const float4*)(b + offs[3]); tp.B1 = (const float2*)(b + offs[4]);
    tp.C0 = (const float4*)(b + offs[5]); tp.C1 = (const float4*)(b + offs[6]); tp.C2 = (const float*)(b + offs[7]);
    tp.D0 = (const float4*)(b + offs[8]); tp.D1 = (const float4*)(b + offs[9]); tp.D2 = (const float4*)(b + offs[10]);

    // ---- geometry: AoS chunks -> pinned -> device staging -> k_repack into the SoA planes ---------------------------
    const auto t_geo = std::chrono::steady_clock::now();
    uint32_t turn = 0;
    for (uint32_t i = 0; i < n_meshes && n_tri; ++i) {
        const uint64_t s = std::max<uint64_t>(first, mesh_first[i]), e = std::min<uint64_t>(last, mesh_first[i + 1]);
        if (e <= s) continue;
        const uint32_t stride = meshes[i].stride_floats;
        const size_t tri_bytes = (size_t)3 * stride * sizeof(float);
        const size_t per_chunk = std::max<size_t>(1, kStageChunk / tri_bytes) * tri_bytes;    // whole triangles per chunk
        const char* src = (const char*)(meshes[i].vertices + (size_t)(s - mesh_first[i]) * 3 * stride);
        const uint32_t dst0 = (uint32_t)(s - first);
        const m2s_status st = staged_h2d(c, src, (size_t)(e - s) * tri_bytes, per_chunk, nullptr, turn,
            [&](char* d, size_t off, size_t n) {
                launch_repack((const float*)d, stride, (uint32_t)(n / tri_bytes), 0, (uint32_t)(n / tri_bytes),
                              dst0 + (uint32_t)(off / tri_bytes), tp, c->stream);
            });
        if (st != M2S_OK) return st;
    }
    c->last_upload_ms[1] = ms_since(t_geo);

    // ---- textures: level 0 through the same staging, levels 1..4 on the device (glUtils.cpp:292-313) -----------------
    const auto t_tex = std::chrono::steady_clock::now();
    for (TexPlan& p : tex_plan) {
        uint32_t* mem = (uint32_t*)(A + p.arena_off);
        p.d.texels = mem;
        const m2s_status st = staged_h2d(c, (const char*)p.src, (size_t)p.d.w * p.d.h * 4, kStageChunk, (char*)mem, turn,
                                         [](char*, size_t, size_t) {});
        if (st != M2S_OK) return st;
        for (uint32_t l = 1; l < p.d.n_levels; ++l)
            launch_mip_level(mem + p.d.off[l - 1], std::max(1u, p.d.w >> (l - 1)), std::max(1u, p.d.h >> (l - 1)),
                             mem + p.d.off[l], std::max(1u, p.d.w >> l), std::max(1u, p.d.h >> l), c->stream);
    }
    for (ComboPlan& cp : combo_plan) {
        uint32_t* mem = (uint32_t*)(A + cp.arena_off);
        cp.d.texels = mem;
        const TexDesc &ta = tex_plan[cp.ia].d, &tn = tex_plan[cp.in].d, &tm = tex_plan[cp.im].d;
        for (uint32_t l = 0; l < ta.n_levels; ++l)
            launch_combo_level(ta.texels + ta.off[l], tn.texels + tn.off[l], tm.texels + tm.off[l], std::max(1u, ta.w >> l),
                               std::max(1u, ta.h >> l), mem + cp.d.coff[l], c->stream);
    }
    std::vector<MeshParams> mp(n_mp);
    memset(mp.data(), 0, mp.size() * sizeof(MeshParams));
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const m2s_mesh& m = meshes[i];
        MeshParams& p = mp[i];
        memcpy(p.bmin, m.bbox_min, 12);
        memcpy(p.bmax, m.bbox_max, 12);
        memcpy(p.color, m.base_color, 16);
        for (int k = 0; k < 3; ++k) if (mesh_tex[i][k] >= 0) p.tex[k] = tex_plan[mesh_tex[i][k]].d;
        if (mesh_combo[i] >= 0) p.combo = combo_plan[mesh_combo[i]].d;
    }
    c->d_meshes = (MeshParams*)(A + o_meshes);
    c->d_mesh_first = (uint32_t*)(A + o_mesh_first);
    HIPCHK(c, hipMemcpyAsync(c->d_meshes, mp.data(), mp.size() * sizeof(MeshParams), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_mesh_first, mesh_first.data(), mesh_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    c->scene.meshes = c->d_meshes;
    c->scene.mesh_first = c->d_mesh_first;
    c->scene.mesh_of8 = (const uint2*)(A + o_mesh_of8);
    launch_mesh_table(c->scene, (uint2*)(A + o_mesh_of8), c->stream);   // (after mesh_first: same stream)

    // ---- work buffers -----------------------------------------------------------------------------
    c->d_cnt = (uint32_t*)(A + o_cnt);
    c->d_off = (uint32_t*)(A + o_off);
    c->d_partials = (uint32_t*)(A + o_partials);
    c->d_chain = (unsigned long long*)(A + o_chain);
    c->d_biglist = (BigItem*)(A + o_biglist);
    c->d_bigmeta = (uint32_t*)(A + o_bigmeta);
    c->d_bands = (unsigned long long*)(A + o_bands);
    c->d_wg_base = (unsigned long long*)(A + o_wg_base);
    c->d_batch_first = (uint32_t*)(A + o_batch);
    c->n_batch_tab = 0;
    c->chain_words = chain_words;
    HIPCHK(c, hipMemsetAsync(c->d_chain, 0, chain_words * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // mp / mesh_first are host temporaries; the caller's buffers are released
    HIPCHK(c, hipGetLastError());
    c->last_upload_ms[2] = ms_since(t_tex);
    c->last_upload_ms[0] = ms_since(t_begin);
    c->has_scene = true;
    return M2S_OK;
}

// Allocates what the first upload / the first export would otherwise allocate inside their own timed paths (pinned and
// device staging chunks, pinned export chunks): a caller that brings the context up on a second thread while it parses the
// input file (the command line does) takes ~100 MB of pinned allocations off its critical path.
m2s_status m2s_prepare(m2s_ctx* c, uint32_t flags) {
    if (!c) return M2S_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (flags & M2S_PREPARE_UPLOAD) { const m2s_status s = ensure_stage(c); if (s != M2S_OK) return s; }
    if (flags & M2S_PREPARE_EXPORT)
        for (int k = 0; k < 2; ++k)
            if (!c->h_export[k]) HIPCHK(c, hipHostMalloc((void**)&c->h_export[k], m2s_ply::kChunkRows * sizeof(m2s_gaussian), hipHostMallocDefault));
    return M2S_OK;
}

m2s_status m2s_last_upload_ms(const m2s_ctx* c, float out_ms[4]) {
    if (!c || !out_ms) return M2S_ERR_INVALID;
    memcpy(out_ms, c->last_upload_ms, sizeof c->last_upload_ms);
    return M2S_OK;
}

}  // extern "C"

// Chain words carry a 16-bit launch tag instead of being cleared per launch.  The two single-pass kernels use different
// numbers of words, so a word one of them left behind could read as freshly published 65 536 launches later: when the
// tag wraps, everything in flight is drained and both chains are cleared (once per ~10 s of back-to-back conversions).
static hipError_t next_epoch(m2s_ctx* c, uint32_t* out) {
    const uint32_t e = ++c->epoch;
    *out = e;
    if ((e & 0xFFFFu) != 0) return hipSuccess;
    const size_t bytes = std::max<size_t>(c->chain_words, 1) * sizeof(unsigned long long);
    hipError_t r = hipDeviceSynchronize();
    if (r == hipSuccess && c->d_chain) r = hipMemset(c->d_chain, 0, bytes);
    if (r == hipSuccess && c->d_chain_b) r = hipMemset(c->d_chain_b, 0, bytes);
    return r;
}

// Which form of the single-pass kernel (see m2s_fused2.hip)?  The workgroup-cooperative one unless a workgroup's
// fragments did not fit its LDS stream at this R before.
static bool use_team(const m2s_ctx* c, const m2s_ctx::RInfo& ri) {
    return !(c->pipeline == M2S_PIPELINE_WAVE || ri.team_off);
}
// ... or its sparse form (m2s_sparse.hip): meshes with fewer fragments than triangles, large enough for 64-triangle batches
static bool use_sparse(const m2s_ctx* c, const m2s_ctx::RInfo& ri) {
    return (c->pipeline == M2S_PIPELINE_SPARSE || (c->pipeline == M2S_PIPELINE_AUTO && ri.sparse)) && !ri.sparse_off &&
           sparse_supported(c->scene.n_tri);
}

static bool env_on(const char* name) { const char* v = std::getenv(name); return v && *v && *v != '0'; }   // (debug switches)
// XCD bands for this launch of k_fused2 (unit 256) or k_sparse (unit 512 triangles per workgroup): the table an earlier launch
// of the same kernel at this R left behind — or, if there is none, ask this launch to record where every workgroup's output
// starts, from which the table is cut right behind it (second lane: never asked to — two lanes would race on d_wg_base)
static uint32_t band_workgroups(const m2s_ctx* c, uint32_t unit) {
    const uint32_t team = fused2_band_workgroups(c->scene.n_tri);
    return !team ? 0u : unit == 256u ? team : sparse_workgroups(c->scene.n_tri);
}
static BandInfo bands_for(const m2s_ctx* c, const m2s_ctx::RInfo& ri, uint32_t unit, bool may_write, bool* writes) {
    BandInfo b{};
    if (writes) *writes = false;
    const uint32_t n_wg = band_workgroups(c, unit);
    if (!n_wg || !c->d_bands || (unit == 256u && c->n_batch_tab) || env_on("M2S_NO_BANDS")) return b;
    if (ri.bands_ready && ri.bands_unit == unit && ri.band_width) {
        b.table = c->d_bands + (size_t)ri.band_slot * kBandTableWords;
        b.max_width = ri.band_width;
    }
    else if (may_write) { b.out = c->d_wg_base; if (writes) *writes = true; }
    return b;
}
// behind a launch that recorded its workgroups' bases: cut the bands of the next launches at this R
static void pick_bands(const m2s_ctx* c, const m2s_ctx::RInfo& ri, uint32_t unit, const unsigned long long* total, hipStream_t st) {
    const uint32_t n_wg = band_workgroups(c, unit);
    // estimated work per triangle / per fragment.  k_fused2: 214 / 140 (cycles of its triangle phase per 64 triangles and of a strip per 64
    // fragments, tools/team_timing.py; config 3 is insensitive between 100 and 300 per triangle).  k_sparse: most triangles only pay
    // tier 1: 115 / 140, measured on config 5 at full size (profiles/r03/ab_band_cost_weights_c5.log: 100 / 140 2.86 ms, 115 2.86,
    // 130 2.90, 145 2.93, 160 2.99, 85 2.96).  (Cutting by MEASURED workgroup lifetimes instead was no better there and much worse
    // on config 3: a lifetime in the unbanded launch includes waits that depend on where the workgroup was dispatched.)
    uint32_t cost_tri = unit == 256u ? 214u : 115u, cost_frag = 140;
    if (const char* v = std::getenv("M2S_BAND_COST")) { unsigned a = 0, b = 0; if (sscanf(v, "%u,%u", &a, &b) == 2 && (a || b)) { cost_tri = a; cost_frag = b; } }   // debug
    launch_pick_bands(c->d_wg_base, n_wg, unit, c->scene.n_tri, total, band_max_width(n_wg), cost_tri, cost_frag,
                      c->d_bands + (size_t)ri.band_slot * kBandTableWords, c->h_bands + (size_t)ri.band_slot * 9, st);
}

// the conversion that cut the bands has completed: how wide is the widest one?  (0: the cuts do not describe this scene — never used)
static uint32_t band_width_of(const m2s_ctx* c, const m2s_ctx::RInfo& ri, uint32_t unit) {
    const unsigned long long* cut = c->h_bands + (size_t)ri.band_slot * 9;
    const uint32_t n_wg = band_workgroups(c, unit);
    if (!n_wg || cut[0] != 0 || cut[8] != n_wg) return 0;
    uint32_t w = 0;
    for (int x = 0; x < 8; ++x) {
        if (cut[x + 1] < cut[x]) return 0;
        w = std::max<uint32_t>(w, (uint32_t)(cut[x + 1] - cut[x]));
    }
    return w;
}

static BatchTable batches_for(const m2s_ctx* c) {
    return c->n_batch_tab ? BatchTable{ c->d_batch_first, c->n_batch_tab } : BatchTable{ nullptr, 0u };
}

static uint64_t resolve_cap(const m2s_ctx* c, uint32_t R) {
    if (c->cap_policy == 0) return 0;
    if (c->cap_policy > 0) return (uint64_t)c->cap_policy;
    // ConversionPass.cpp:21-24 (unsigned int arithmetic wraps)
    const uint32_t mc = std::max<uint32_t>(1u, c->n_meshes_total);
    const uint32_t mx = R * R * 6u * mc;
    return std::min(mx, kMaxGaussiansToSort);
}

// exact fragment count of the scene at R (k_count + scan + read-back): the one host round trip a NEW SCENE pays
static m2s_status count_now(m2s_ctx* c, uint32_t R, hipStream_t st) {
    const SceneDev& sc = c->scene;
    const bool prof = c->profiling;
    if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
    launch_count(sc, R, c->d_cnt, c->d_partials, st);
    if (prof) HIPCHK(c, hipEventRecord(c->ev[1], st));
    launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
    if (prof) HIPCHK(c, hipEventRecord(c->ev[2], st));
    HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    if (prof)
        for (int k = 0; k < 2; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_ms[k], c->ev[k], c->ev[k + 1]));
    c->frag_per_R2 = (double)c->h_total[0] / ((double)R * (double)R);
    // A scene small enough for ONE generation of workgroups (fused_tpw < 64) lasts as long as its slowest workgroup: cut it into
    // batches of equal estimated work instead of equal triangle counts (C2 stand-in: fragments per workgroup vary 1 : 3 over a
    // cube-sphere face).  Work = 214 per triangle + 140 per fragment (cycles of the triangle phase per 64 triangles and of a strip
    // per 64 fragments, tools/team_timing.py); fragments scale with R^2 everywhere alike, so the table serves every density.
    if (batch_table_capacity(sc.n_tri) && !c->n_batch_tab && !std::getenv("M2S_NO_BATCH_TABLE")) {
        try {
            std::vector<uint32_t> cnt(sc.n_tri), first;
            HIPCHK(c, hipMemcpy(cnt.data(), c->d_cnt, (size_t)sc.n_tri * sizeof(uint32_t), hipMemcpyDeviceToHost));
            const uint32_t n_target = n_fused_waves(sc.n_tri);
            const double ct = 214.0, cf = 140.0;
            double total = 0.0;
            for (uint32_t t = 0; t < sc.n_tri; ++t) total += ct + cf * (double)cnt[t];
            const double quota = total / (double)n_target;
            first.reserve(batch_table_capacity(sc.n_tri));
            first.push_back(0);
            double acc = 0.0;
            uint32_t in_batch = 0;
            for (uint32_t g = 0; g < sc.n_tri; g += 8) {          // batches start at multiples of 8 (mesh_of8)
                const uint32_t ge = std::min(g + 8u, sc.n_tri);
                double gc = 0.0;
                for (uint32_t t = g; t < ge; ++t) gc += ct + cf * (double)cnt[t];
                // close the batch before this group if it is full, or if the work so far has reached the batch's share
                if (in_batch && (in_batch + (ge - g) > 64u || acc + 0.5 * gc >= quota * (double)first.size())) { first.push_back(g); in_batch = 0; }
                acc += gc;
                in_batch += ge - g;
            }
            first.push_back(sc.n_tri);
            if (first.size() <= batch_table_capacity(sc.n_tri)) {
                HIPCHK(c, hipMemcpy(c->d_batch_first, first.data(), first.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                c->n_batch_tab = (uint32_t)first.size() - 1u;
            }
        } catch (...) { /* no table: uniform batches */ }
    }
    return M2S_OK;
}

// The context-owned record buffer is a grow-only pool.  The reference re-creates its SSBO whenever the cap changes
// (ConversionPass.cpp:25-33), i.e. on every move of the density slider; here a change of R costs no allocation: the pool
// doubles until it reaches the largest size the cap policy can ask for (7 M records = 672 MB for the reference formula).
static m2s_status ensure_records(m2s_ctx* c, uint64_t want) {
    if (c->records_cap >= want && c->d_records) return M2S_OK;
    uint64_t grow = std::max<uint64_t>(want, 1);
    if (c->records_cap) grow = std::max(grow, 2 * c->records_cap);
    if (c->cap_policy < 0) grow = std::max(want, std::min<uint64_t>(grow, kMaxGaussiansToSort));
    drain_in_flight(c);   // nothing may still be writing the buffer that is about to be released
    if (c->d_records && c->last_records == c->d_records) { c->last_records = nullptr; c->last_stored = 0; }   // they go with the old pool
    if (c->d_records) { (void)hipFree(c->d_records); c->d_records = nullptr; c->records_cap = 0; }
    hipError_t e = hipMalloc(&c->d_records, grow * sizeof(m2s_gaussian));
    if (e != hipSuccess && grow > want) { grow = want; e = hipMalloc(&c->d_records, grow * sizeof(m2s_gaussian)); }
    HIPCHK(c, e);
    c->records_cap = grow;
    // in-flight conversions into the old buffer are complete but their records are gone
    for (uint32_t k = 0; k < c->slot_count; ++k) {
        auto& sl = c->slot[(c->slot_head + k) % M2S_MAX_IN_FLIGHT];
        if (sl.own_lane == 0) { sl.d_out = c->d_records; sl.gen = c->buf_gen[0] - 1u; }
    }
    return M2S_OK;
}

// Multi-pass pipeline: handles every triangle size, output-balanced.  Second generation (m2s_emit2.hip): k_count_scan
// (count + offsets + per-triangle setup records, one kernel) -> k_emit2 (wave-granular).  M2S_MULTIPASS_V1=1 selects the
// first generation (count -> scan -> offsets -> emit, m2s_kernels.hip) for A/B measurements.
static bool multipass_v1() { static const bool v1 = std::getenv("M2S_MULTIPASS_V1") != nullptr; return v1; }

static m2s_status ensure_multipass_buffers(m2s_ctx* c, uint64_t limit) {
    const uint32_t n_start = multipass_v1() ? (uint32_t)((limit + kEmitF - 1) / kEmitF) : emit2_slices(limit);
    if (c->start_cap < n_start) {
        drain_in_flight(c);
        if (c->d_start) { (void)hipFree(c->d_start); c->d_start = nullptr; c->start_cap = 0; }
        const size_t want = std::max<size_t>(std::max<size_t>(n_start, 2 * c->start_cap), 16384);
        HIPCHK(c, hipMalloc((void**)&c->d_start, want * sizeof(uint32_t)));
        c->start_cap = want;
    }
    if (!multipass_v1() && !c->d_setup) HIPCHK(c, hipMalloc(&c->d_setup, setup_bytes(c->scene.n_tri)));
    return M2S_OK;
}

// enqueues the pipeline's kernels and the read-back of the counter into *h_res (pinned); no synchronisation
static m2s_status enqueue_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, bool counted, bool prof,
                                    unsigned long long* h_res, hipStream_t st) {
    const SceneDev& sc = c->scene;
    if (multipass_v1()) {
        if (!counted) {
            if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
            launch_count(sc, R, c->d_cnt, c->d_partials, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[1], st));
            launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[2], st));
        }
        const uint32_t n_blocks = (uint32_t)((limit + kEmitF - 1) / kEmitF);
        launch_offsets(c->d_cnt, c->d_partials, sc.n_tri, c->d_off, c->d_start, n_blocks, st);
        if (prof) HIPCHK(c, hipEventRecord(c->ev[3], st));
        launch_emit(sc, R, c->d_off, c->d_start, c->d_total, limit, d_out, n_blocks, st);
        if (prof) HIPCHK(c, hipEventRecord(c->ev[4], st));
    } else {
        uint32_t epoch;
        HIPCHK(c, next_epoch(c, &epoch));
        if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
        launch_count_scan(sc, R, c->d_off, c->d_start, emit2_slices(limit), c->d_chain, epoch, c->d_total, c->d_setup,
                          reinterpret_cast<uint32_t*>(&h_res[1]), st);
        if (prof) { HIPCHK(c, hipEventRecord(c->ev[1], st)); HIPCHK(c, hipEventRecord(c->ev[3], st)); }
        launch_emit2(sc, R, c->d_off, c->d_start, c->d_total, limit, c->d_setup, d_out, st);
        if (prof) HIPCHK(c, hipEventRecord(c->ev[4], st));
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(&h_res[0], c->d_total, 8, hipMemcpyDeviceToHost, st));
    return M2S_OK;
}

static m2s_status run_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, bool counted, hipStream_t st) {
    const bool prof = c->profiling;
    { const m2s_status s = ensure_multipass_buffers(c, limit); if (s != M2S_OK) return s; }
    c->h_total[0] = 0; c->h_total[1] = 0;
    { const m2s_status s = enqueue_multipass(c, R, d_out, limit, counted, prof, c->h_total, st); if (s != M2S_OK) return s; }
    HIPCHK(c, hipStreamSynchronize(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
    if (c->h_total[1] >> 32) return fail(c, M2S_ERR_HIP, "multi-pass pipeline: look-back chain timed out");
    if (prof) {
        if (multipass_v1()) { for (int k = counted ? 2 : 0; k < 4; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_ms[k], c->ev[k], c->ev[k + 1])); }
        else {
            HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_COUNT], c->ev[0], c->ev[1]));
            HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_EMIT], c->ev[3], c->ev[4]));
        }
    }
    return M2S_OK;
}

static m2s_status run_pass(m2s_ctx* c, uint32_t R, void* d_user, uint64_t user_cap, hipStream_t st, uint64_t* out_total,
                           bool from_submit = false) {
    if (!c->has_scene) return fail(c, M2S_ERR_STATE, "m2s_upload_scene has not been called");
    if (c->slot_count && !from_submit)
        return fail(c, M2S_ERR_STATE, "conversions submitted with m2s_convert_submit are still in flight: m2s_convert_wait first");
    if (R == 0 || R > 4096) return fail(c, M2S_ERR_INVALID, "R must be in [1, 4096]");
    HIPCHK(c, hipSetDevice(c->device));
    // A synchronous conversion shares the work buffers (chain, counts, offsets, deferred-triangle list) with whatever
    // was submitted before it, possibly on other streams: let that finish first.
    if (from_submit) drain_in_flight(c);
    const SceneDev& sc = c->scene;
    const uint64_t cap = resolve_cap(c, R);
    const bool prof = c->profiling;
    c->last_R = R;
    c->records_stale = false;
    memset(c->last_ms, 0, sizeof c->last_ms);

    if (sc.n_tri == 0) {
        c->last_total = c->last_stored = 0;
        if (!d_user && !c->d_records) {   // (an empty shard is a conversion like any other: its consumers see zero records, not "no conversion")
            const m2s_status s = ensure_records(c, 1);
            if (s != M2S_OK) return s;
        }
        c->last_records = d_user ? d_user : c->d_records;
        if (out_total) *out_total = 0;
        return M2S_OK;
    }
    m2s_ctx::RInfo& ri = rinfo_for(c, R);

    // ---- AUTO: which pipeline for this scene at this R? ---------------------------------------------
    // The single-pass kernel wins while triangles are small (it does the per-triangle work once and needs no second
    // sweep); with more than ~11 fragments per triangle on average the output-partitioned multi-pass pipeline is
    // faster and soon much faster (2.74 M fragments at R = 1024 from 1 M / 250 k / 125 k / 62 k triangles: fused 0.167 /
    // 0.138 / 0.323 / 0.626 ms, multi-pass 0.214 / 0.137 / 0.136 / 0.154 ms; tools/auto_probe.py).
    // The fragment count of a scene is proportional to R^2 (window coordinates scale with R), so ONE exact count — taken
    // at the scene's first conversion, 0.02-0.06 ms plus a host round trip — decides for every later R without touching
    // the device: the threshold is not sharp, and both pipelines produce the same bytes anyway.
    bool counted = false;  // k_count + k_scan already ran in this call, at this R
    const bool need_estimate = (c->pipeline == M2S_PIPELINE_AUTO && !ri.decided) || (!d_user && !cap);
    if (need_estimate && c->frag_per_R2 < 0.0) {
        const m2s_status s = count_now(c, R, st);
        if (s != M2S_OK) return s;
        counted = true;
    }
    const double predicted = c->frag_per_R2 >= 0.0 ? c->frag_per_R2 * (double)R * (double)R : 0.0;
    if (c->pipeline == M2S_PIPELINE_AUTO && !ri.decided) {
        ri.decided = true;
        const double frags = counted ? (double)c->h_total[0] : predicted;
        ri.multipass = frags >= 11.0 * (double)sc.n_tri;
        // about as many fragments as triangles, or fewer: many triangles cover no pixel centre, the sparse form drops them cheaply
        // (crossover measured with tools/sparse_probe.py: see DESIGN.md)
        ri.sparse = !ri.multipass && frags < sparse_frags_per_triangle(sc.n_tri) * (double)sc.n_tri && !std::getenv("M2S_NO_SPARSE");
    }

    // ---- where do the records go, and how many may be stored? ------------------------------------
    uint64_t limit;
    float4* d_out;
    if (d_user) {
        limit = cap ? std::min(cap, user_cap) : user_cap;
        d_out = (float4*)d_user;
    } else {
        // unlimited policy: room for the predicted count plus slack; a conversion that still overflows is repeated below
        const uint64_t want = cap ? cap : (counted ? std::max<uint64_t>(c->h_total[0], 1) : (uint64_t)(predicted * 1.02) + 4096);
        const m2s_status s = ensure_records(c, want);
        if (s != M2S_OK) return s;
        limit = cap ? cap : c->records_cap;
        d_out = (float4*)c->d_records;
    }
    if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;

    for (int round = 0; round < 2; ++round) {
    // ---- run ---------------------------------------------------------------------------------------
    bool done = false;
    if (c->pipeline != M2S_PIPELINE_MULTIPASS && !ri.multipass) {
        counted = false;   // the fused kernel does its own counting; a count taken above only sized / decided
        // single-pass kernel; triangles too large for its in-workgroup budget are only counted.
        // No memset, no memcpy: the look-back chain is epoch-tagged and the kernel writes the fragment
        // counter and its two status words straight into pinned host memory.
        uint32_t any_big = 0, err = 0;
        bool wrote_bands = false;
        for (int attempt = 0; attempt < 3; ++attempt) {
            const bool sparse = use_sparse(c, ri);
            const bool team = !sparse && use_team(c, ri);
            c->h_total[0] = 0;
            c->h_total[1] = 0;
            uint32_t epoch;
            HIPCHK(c, next_epoch(c, &epoch));
            if (prof) HIPCHK(c, hipEventRecord(c->ev[5], st));
            const uint32_t unit = sparse ? kSparseTrianglesPerWorkgroup : 256u;
            if (sparse) launch_sparse(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                                      c->d_biglist, c->d_bigmeta, bands_for(c, ri, unit, true, &wrote_bands), st);
            else if (team) launch_fused2(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                                    c->d_biglist, c->d_bigmeta, bands_for(c, ri, unit, true, &wrote_bands), batches_for(c), st);
            else launch_fused(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                              c->d_biglist, c->d_bigmeta, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[6], st));
            if ((team || sparse) && wrote_bands) pick_bands(c, ri, unit, &c->h_total[0], st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
            if (prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_FUSED], c->ev[5], c->ev[6]));
            any_big = (uint32_t)(c->h_total[1] & 0xFFFFFFFFull);
            err = (uint32_t)(c->h_total[1] >> 32);
            c->last_pipeline = sparse ? M2S_PIPELINE_SPARSE : team ? M2S_PIPELINE_TEAM : M2S_PIPELINE_WAVE;
            if ((team || sparse) && !err && wrote_bands) { ri.bands_ready = true; ri.bands_unit = unit; ri.band_width = band_width_of(c, ri, unit); }
            if (err && std::getenv("M2S_DEBUG"))
                fprintf(stderr, "[m2s] single-pass kernel (%s) reported 0x%x at R = %u: trying the next form\n", sparse ? "sparse" : team ? "team" : "wave", err, R);
            if (!(err && (team || sparse))) break;
            // a workgroup's fragments did not fit the kernel's LDS stream (or a wait timed out): sparse -> team -> the
            // one-wave-per-batch form, which has no such limit.  Remember it for this scene and R, forget what the aborted
            // launch listed, try again.
            if (sparse) ri.sparse_off = true; else ri.team_off = true;
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
        }
        done = true;
        // a clean single-kernel conversion: the same scene at the same R can be submitted asynchronously from now on
        ri.async_ok = !err && !any_big;
        if (err) {
            // The bounded look-back spin gave up (never observed; would need a dispatcher that starves earlier
            // workgroups).  Degrade to the multi-pass pipeline, which has no inter-workgroup dependency.
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
            ri.multipass = true;
            done = false;
        } else
        if (any_big) {
            uint32_t meta[4] = { 0, 0, 0, 0 };
            HIPCHK(c, hipMemcpyAsync(meta, c->d_bigmeta, sizeof meta, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, sizeof meta, st));   // restore the "zero between conversions" invariant
            const uint64_t total_now = c->h_total[0];
            if (meta[0] > 256 && (uint64_t)meta[2] * 8 > total_now) {
                // Scene dominated by mid-size / big triangles (e.g. a coarse mesh at high density): one workgroup per
                // triangle chunk would be mostly empty.  The output-partitioned multi-pass pipeline packs them densely;
                // remember the decision so that later conversions of this scene at this R go straight to it.
                ri.multipass = true;
                done = false;
            } else {
                // second stage: emit exactly the deferred triangles, one workgroup per 1024-fragment chunk
                if (prof) HIPCHK(c, hipEventRecord(c->ev[3], st));
                launch_emit_big(sc, R, c->d_biglist, meta[0], meta[1], limit, d_out, st);
                if (prof) HIPCHK(c, hipEventRecord(c->ev[4], st));
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(st));
                if (prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_EMIT], c->ev[3], c->ev[4]));
            }
        }
    }
    if (!done) {
        m2s_status s = run_multipass(c, R, d_out, limit, counted, st);
        if (s != M2S_OK) return s;
        ri.mp_ready = true;
        c->last_pipeline = M2S_PIPELINE_MULTIPASS;
    }
    // unlimited policy, context-owned buffer: the prediction was too low — make room for the exact count and repeat
    // (never seen with the 2 % slack; the fragment count scales with R^2 up to clipping at the viewport edge)
    if (!d_user && !cap && c->h_total[0] > limit && limit < 0xFFFFFFFFull && round == 0) {
        const m2s_status s = ensure_records(c, c->h_total[0]);
        if (s != M2S_OK) return s;
        limit = std::min<uint64_t>(c->records_cap, 0xFFFFFFFFull);
        d_out = (float4*)c->d_records;
        counted = false;
        continue;
    }
    break;
    }
    const uint64_t total = c->h_total[0];
    if (total > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 fragments: offsets are 32-bit");
    c->frag_per_R2 = (double)total / ((double)R * (double)R);
    c->last_total = total;
    c->last_stored = std::min(total, limit);
    c->last_records = d_out;
    if (!d_user) { if (c->buf_R[0] != R) { c->buf_R[0] = R; ++c->buf_gen[0]; } }
    if (out_total) *out_total = total;
    return M2S_OK;
}

extern "C" {

m2s_status m2s_convert(m2s_ctx* c, uint32_t R, uint64_t* out_total) {
    if (!c) return M2S_ERR_INVALID;
    return run_pass(c, R, nullptr, 0, c->stream, out_total);
}

m2s_status m2s_convert_into(m2s_ctx* c, uint32_t R, void* d_records, uint64_t capacity_records, void* hip_stream,
                            uint64_t* out_total) {
    if (!c) return M2S_ERR_INVALID;
    if (!d_records && capacity_records) return fail(c, M2S_ERR_INVALID, "d_records is NULL");
    if (!d_records) return fail(c, M2S_ERR_INVALID, "d_records is NULL (use m2s_convert for the context-owned buffer)");
    return run_pass(c, R, d_records, capacity_records, (hipStream_t)hip_stream, out_total);
}

// ---- asynchronous submissions --------------------------------------------------------------------------
m2s_status m2s_convert_submit(m2s_ctx* c, uint32_t R, void* d_records, uint64_t capacity_records, void* hip_stream) {
    if (!c) return M2S_ERR_INVALID;
    if (!d_records && capacity_records) return fail(c, M2S_ERR_INVALID, "d_records is NULL");
    if (c->slot_count == M2S_MAX_IN_FLIGHT) return fail(c, M2S_ERR_STATE, "M2S_MAX_IN_FLIGHT conversions already in flight");
    if (!c->has_scene) return fail(c, M2S_ERR_STATE, "m2s_upload_scene has not been called");
    if (R == 0 || R > 4096) return fail(c, M2S_ERR_INVALID, "R must be in [1, 4096]");
    hipStream_t st = d_records ? (hipStream_t)hip_stream : c->stream;
    const uint32_t k = (c->slot_head + c->slot_count) % M2S_MAX_IN_FLIGHT;
    m2s_ctx::Slot& sl = c->slot[k];
    const uint64_t cap = resolve_cap(c, R);
    m2s_ctx::RInfo& ri = rinfo_for(c, R);
    // Fast path: this scene at this R already converted cleanly with the single kernel (no deferred triangles, so no
    // host decision between kernels) and the output buffer needs no (re)allocation.
    const bool own_ready = d_records || (c->d_records && c->buf_R[0] == R && (cap ? c->records_cap >= cap : true));
    const bool fast = c->scene.n_tri > 0 && ri.async_ok && c->pipeline != M2S_PIPELINE_MULTIPASS && !ri.multipass && own_ready;
    // Multi-pass conversions have no host decision between their four kernels either; once this (scene, R) has been
    // converted that way (work buffers sized, AUTO decision taken) they are enqueued without waiting as well.
    // (With kernel timing on they run synchronously: the per-kernel events are shared.)
    const bool fast_mp = !fast && c->scene.n_tri > 0 && own_ready && !c->profiling && ri.mp_ready &&
                         (c->pipeline == M2S_PIPELINE_MULTIPASS || (ri.decided && ri.multipass));
    sl.R = R;
    sl.wrote_bands = false;
    sl.own_lane = d_records ? -1 : 0;
    // All conversions of a context share its work buffers (look-back chain, counts, offsets, the deferred-triangle
    // list): they must execute in submission order.  On one stream that is automatic; a submission on ANOTHER stream
    // than the newest one in flight is ordered behind it with an event.  (The second lane is exempt: it has its own
    // chain and is only taken by single-kernel conversions that touch nothing else.)
    auto chain_behind_newest = [&](hipStream_t on) -> hipError_t {
        // the newest in-flight submission that used the SHARED work buffers (everything but second-lane submissions, which
        // have a chain of their own): if it runs on another stream, this one is ordered behind it
        for (uint32_t q = c->slot_count; q-- > 0;) {
            const m2s_ctx::Slot& prev = c->slot[(c->slot_head + q) % M2S_MAX_IN_FLIGHT];
            if (prev.sync_result || !prev.shared_work) continue;
            return prev.st == on ? hipSuccess : hipStreamWaitEvent(on, prev.done, 0);
        }
        return hipSuccess;
    };
    if (fast_mp) {
        HIPCHK(c, hipSetDevice(c->device));
        uint64_t limit;
        void* d_out;
        if (d_records) { limit = cap ? std::min(cap, capacity_records) : capacity_records; d_out = d_records; }
        else { limit = cap ? cap : c->records_cap; d_out = c->d_records; }
        if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;
        const uint32_t n_start = multipass_v1() ? (uint32_t)((limit + kEmitF - 1) / kEmitF) : emit2_slices(limit);
        if (c->start_cap >= n_start && (multipass_v1() || c->d_setup)) {
            HIPCHK(c, chain_behind_newest(st));
            unsigned long long* res = &c->h_total[2 + 2 * k];
            res[0] = 0; res[1] = 0;
            { const m2s_status ms_ = enqueue_multipass(c, R, (float4*)d_out, limit, false, false, res, st); if (ms_ != M2S_OK) return ms_; }
            HIPCHK(c, hipEventRecord(sl.done, st));
            c->last_pipeline = M2S_PIPELINE_MULTIPASS;
            c->last_submit_stream = st;
            sl.prof = false;
            sl.sync_result = false;
            sl.limit = limit;
            sl.d_out = d_out;
            sl.gen = c->buf_gen[0];
            sl.st = st; sl.shared_work = true;
            ++c->slot_count;
            return M2S_OK;
        }
    }
    if (!fast) {
        // first conversion of a (scene, R), or one that needs the second stage / the multi-pass pipeline: run it now
        // (run_pass first lets everything in flight finish: it may re-allocate the record pool and reuses the work buffers)
        uint64_t total = 0;
        const m2s_status s = run_pass(c, R, d_records, capacity_records, st, &total, true);
        if (s != M2S_OK) return s;
        sl.sync_result = true;
        sl.shared_work = false;
        sl.sync_total = total;
        memcpy(sl.ms, c->last_ms, sizeof sl.ms);   // a later submit overwrites last_ms before this slot is waited for
        sl.limit = c->last_stored;   // already clamped
        sl.d_out = const_cast<void*>(c->last_records);
        sl.gen = c->buf_gen[0];
        ++c->slot_count;
        return M2S_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t limit;
    void* d_out;
    unsigned long long* chain = c->d_chain;
    bool second_lane = false;
    if (d_records) { limit = cap ? std::min(cap, capacity_records) : capacity_records; d_out = d_records; }
    else {
        limit = cap ? cap : c->records_cap;
        d_out = c->d_records;
        if ((k & 1u) && c->lanes == 2) {
            // odd slots: the second lane (allocated on first use).  Records of consecutive conversions then alternate
            // between two context-owned buffers; m2s_device_records / m2s_download follow the conversion last waited for.
            if (!c->stream_b) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
            if (!c->d_chain_b) {
                const size_t words = std::max<size_t>(c->chain_words, 1);
                HIPCHK(c, hipMalloc((void**)&c->d_chain_b, words * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->d_chain_b, 0, words * sizeof(unsigned long long), c->stream_b));
            }
            if (c->records_b_cap < c->records_cap) {
                for (uint32_t q = 0; q < c->slot_count; ++q) {   // nothing may still be writing the old second buffer
                    auto& o = c->slot[(c->slot_head + q) % M2S_MAX_IN_FLIGHT];
                    if (o.own_lane == 1 && !o.sync_result) (void)hipEventSynchronize(o.done);
                }
                if (c->d_records_b) { (void)hipFree(c->d_records_b); c->d_records_b = nullptr; c->records_b_cap = 0; }
                HIPCHK(c, hipMalloc(&c->d_records_b, c->records_cap * sizeof(m2s_gaussian)));
                c->records_b_cap = c->records_cap;
                ++c->buf_gen[1];
            }
            st = c->stream_b;
            chain = c->d_chain_b;
            d_out = c->d_records_b;
            second_lane = true;
            sl.own_lane = 1;
        }
    }
    if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;
    // (every submission that uses the shared chain is ordered behind the newest one that did, whatever stream that ran on — a
    //  first-lane submission after a conversion into a caller's buffer on the caller's stream included: ADVICE r2)
    if (!second_lane) HIPCHK(c, chain_behind_newest(st));
    if (sl.own_lane >= 0) {
        if (c->buf_R[sl.own_lane] != R) { c->buf_R[sl.own_lane] = R; ++c->buf_gen[sl.own_lane]; }
        sl.gen = c->buf_gen[sl.own_lane];
    }
    unsigned long long* res = &c->h_total[2 + 2 * k];
    res[0] = 0; res[1] = 0;
    sl.prof = c->profiling;
    uint32_t epoch;
    HIPCHK(c, next_epoch(c, &epoch));
    if (sl.prof) HIPCHK(c, hipEventRecord(sl.t0, st));
    const bool sparse = use_sparse(c, ri);
    const bool team = !sparse && use_team(c, ri);
    c->last_pipeline = sparse ? M2S_PIPELINE_SPARSE : team ? M2S_PIPELINE_TEAM : M2S_PIPELINE_WAVE;
    sl.bands_unit = sparse ? kSparseTrianglesPerWorkgroup : 256u;
    if (sparse) launch_sparse(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                              c->d_biglist, c->d_bigmeta, bands_for(c, ri, sl.bands_unit, !second_lane && c->lanes == 1, &sl.wrote_bands), st);
    else if (team) launch_fused2(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                            c->d_biglist, c->d_bigmeta, bands_for(c, ri, sl.bands_unit, !second_lane && c->lanes == 1, &sl.wrote_bands), batches_for(c), st);
    else launch_fused(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                      c->d_biglist, c->d_bigmeta, st);
    if (sl.prof) HIPCHK(c, hipEventRecord(sl.t1, st));
    if ((team || sparse) && sl.wrote_bands) pick_bands(c, ri, sl.bands_unit, &res[0], st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(sl.done, st));
    if (!second_lane) c->last_submit_stream = st;
    sl.st = st; sl.shared_work = !second_lane; sl.ri_gen = ri.gen;
    sl.sync_result = false;
    sl.limit = limit;
    sl.d_out = d_out;
    ++c->slot_count;
    return M2S_OK;
}

m2s_status m2s_convert_wait(m2s_ctx* c, uint64_t* out_total) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->slot_count) return fail(c, M2S_ERR_STATE, "no conversion in flight");
    const uint32_t k = c->slot_head;
    m2s_ctx::Slot& sl = c->slot[k];
    c->slot_head = (c->slot_head + 1) % M2S_MAX_IN_FLIGHT;
    --c->slot_count;
    // Conversions into a context-owned buffer overwrite each other in order, like repeated draws into one SSBO.  If a
    // LATER submission at another R has been enqueued into the buffer this conversion wrote, its records are not what
    // that buffer holds (any more): consumers (m2s_download, m2s_export_ply, m2s_prepass, sorts) then refuse instead of
    // returning the other conversion's records.
    const bool stale = sl.own_lane >= 0 && sl.gen != c->buf_gen[sl.own_lane];
    if (sl.sync_result) {   // run_pass already filled last_*
        if (out_total) *out_total = sl.sync_total;
        c->last_total = sl.sync_total; c->last_stored = sl.limit; c->last_records = sl.d_out; c->last_R = sl.R;
        c->records_stale = stale;
        memcpy(c->last_ms, sl.ms, sizeof sl.ms);
        return M2S_OK;
    }
    HIPCHK(c, hipEventSynchronize(sl.done));
    const uint64_t total = c->h_total[2 + 2 * k];
    const uint32_t any_big = (uint32_t)(c->h_total[3 + 2 * k] & 0xFFFFFFFFull), err = (uint32_t)(c->h_total[3 + 2 * k] >> 32);
    memset(c->last_ms, 0, sizeof c->last_ms);
    if (sl.prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_FUSED], sl.t0, sl.t1));
    // what is remembered about (scene, R) may have been dropped since the submission (full table, m2s_set_pipeline): only an
    // entry of the generation the submission was made under is updated, and none is created here
    auto rit = c->rinfo.find(sl.R);
    m2s_ctx::RInfo* rip = (rit != c->rinfo.end() && rit->second.gen == sl.ri_gen) ? &rit->second : nullptr;
    if (err || any_big) {   // cannot happen for a scene/R that converted cleanly before; never return partial output silently
        if (rip) rip->async_ok = false;
        (void)hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), c->stream);
        (void)hipStreamSynchronize(c->stream);
        return fail(c, M2S_ERR_STATE, "asynchronous conversion needed a host decision; convert synchronously");
    }
    if (sl.wrote_bands && rip) { rip->bands_ready = true; rip->bands_unit = sl.bands_unit; rip->band_width = band_width_of(c, *rip, sl.bands_unit); }   // that launch has completed: its band table is in place
    if (total > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 fragments: offsets are 32-bit");
    c->last_total = total;
    c->last_stored = std::min(total, sl.limit);
    c->last_records = sl.d_out;
    c->last_R = sl.R;
    c->records_stale = stale;
    if (out_total) *out_total = total;
    return M2S_OK;
}

uint64_t m2s_num_stored(const m2s_ctx* c) { return c ? c->last_stored : 0; }
const void* m2s_device_records(const m2s_ctx* c) { return c ? c->last_records : nullptr; }
uint64_t m2s_num_triangles(const m2s_ctx* c) { return c ? c->scene.n_tri : 0; }

m2s_status m2s_download(m2s_ctx* c, m2s_gaussian* dst, uint64_t capacity_records) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->last_stored) return M2S_OK;
    if (c->records_stale) return fail(c, M2S_ERR_STATE, kStaleMsg);
    if (!dst) return fail(c, M2S_ERR_INVALID, "dst is NULL");
    if (capacity_records < c->last_stored) return fail(c, M2S_ERR_CAPACITY, "dst holds fewer records than were stored");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dst, c->last_records, c->last_stored * sizeof(m2s_gaussian), hipMemcpyDeviceToHost));
    return M2S_OK;
}

m2s_status m2s_download_triangle_counts(m2s_ctx* c, uint32_t* dst, uint64_t n) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->has_scene || !c->last_R) return fail(c, M2S_ERR_STATE, "no conversion has run");
    if (n < c->scene.n_tri) return fail(c, M2S_ERR_CAPACITY, "dst holds fewer entries than triangles in range");
    if (!c->scene.n_tri) return M2S_OK;
    HIPCHK(c, hipSetDevice(c->device));
    // the fused pipeline keeps counts in registers only: (re)run the counting kernel for the last R
    launch_count(c->scene, c->last_R, c->d_cnt, c->d_partials, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(dst, c->d_cnt, (size_t)c->scene.n_tri * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return M2S_OK;
}

// Downloads rows [0, n_rows) of the last conversion's records and writes them as rows [first_row, first_row + n_rows) of
// a .ply that holds total_rows rows (whole-file export: first_row = 0, n_rows = total_rows = stored records).
//   format 0 (248 B / row from a 96 B record): the records cross PCIe, host threads encode (a device-side encoder would
//            inflate the transfer 2.6x);
//   formats 1, 2 (76 / 48 B per row): the rows are encoded ON THE DEVICE (k_encode_rows, m2s_export.hip) — log scale,
//            SH-DC colour, logit opacity; for format 2 also the octahedral normal and the u8 packing (parsers.cpp:232-428)
//            — so what crosses PCIe is the file's own bytes, which go from the pinned buffers straight into the file.
// Either way chunk k+1 is on the bus while chunk k is written.
static m2s_status export_rows(m2s_ctx* c, const char* path, uint32_t format, float gaussian_std, uint64_t first_row, uint64_t n_rows,
                              uint64_t total_rows, bool slice) {
    if (!c->last_R) return fail(c, M2S_ERR_STATE, "no conversion has run (uploaded records carry no resolutionTarget: use m2s_write_ply)");
    if (c->records_stale) return fail(c, M2S_ERR_STATE, kStaleMsg);
    if (n_rows > c->last_stored) return fail(c, M2S_ERR_INVALID, "more rows requested than the last conversion stored");
    HIPCHK(c, hipSetDevice(c->device));
    if (format > 2) format = 0;          // parsers.cpp:646-648
    // SceneManager.cpp:668
    const float scale_multiplier = gaussian_std / static_cast<float>(c->last_R);
    const size_t chunk = m2s_ply::kChunkRows;
    for (int k = 0; k < 2; ++k)
        if (!c->h_export[k]) HIPCHK(c, hipHostMalloc((void**)&c->h_export[k], chunk * sizeof(m2s_gaussian), hipHostMallocDefault));
    m2s_ply::Writer w;
    m2s_status s = slice ? w.open_slice(path, total_rows, format, scale_multiplier, first_row, n_rows)
                         : w.open(path, total_rows, format, scale_multiplier);
    if (s != M2S_OK) { c->err = std::string("could not write ") + path; return s; }
    const bool on_device = format != 0 && !std::getenv("M2S_HOST_ENCODE");
    const size_t unit = on_device ? w.row_bytes() : sizeof(m2s_gaussian);     // bytes per row on the bus
    const char* src = static_cast<const char*>(c->last_records);
    if (on_device && n_rows) {
        const uint64_t need = n_rows * unit;
        if (c->rows_cap < need) {
            if (c->d_rows) { (void)hipFree(c->d_rows); c->d_rows = nullptr; c->rows_cap = 0; }
            HIPCHK(c, hipMalloc(&c->d_rows, need));
            c->rows_cap = need;
        }
        launch_encode_rows((const float4*)c->last_records, n_rows, format, scale_multiplier, (uint8_t*)c->d_rows, c->stream);
        HIPCHK(c, hipGetLastError());
        src = static_cast<const char*>(c->d_rows);
    }
    auto rows_of = [&](uint64_t k) { return (size_t)std::min<uint64_t>(chunk, n_rows - k * chunk); };
    const uint64_t n_chunks = (n_rows + chunk - 1) / chunk;
    if (n_chunks) HIPCHK(c, hipMemcpyAsync(c->h_export[0], src, rows_of(0) * unit, hipMemcpyDeviceToHost, c->stream));
    for (uint64_t k = 0; k < n_chunks && s == M2S_OK; ++k) {
        HIPCHK(c, hipStreamSynchronize(c->stream));                      // chunk k has arrived
        if (k + 1 < n_chunks)
            HIPCHK(c, hipMemcpyAsync(c->h_export[(k + 1) & 1], src + (k + 1) * chunk * unit, rows_of(k + 1) * unit, hipMemcpyDeviceToHost, c->stream));
        // (both return once the pinned buffer has been read)
        s = on_device ? w.append_encoded(reinterpret_cast<const uint8_t*>(c->h_export[k & 1]), rows_of(k)) : w.append(c->h_export[k & 1], rows_of(k));
    }
    const m2s_status cs = w.close();
    if (s == M2S_OK) s = cs;
    if (s != M2S_OK) c->err = std::string("could not write ") + path;
    return s;
}

m2s_status m2s_export_ply(m2s_ctx* c, const char* path, uint32_t format, float gaussian_std) {
    if (!c || !path) return M2S_ERR_INVALID;
    return export_rows(c, path, format, gaussian_std, 0, c->last_stored, c->last_stored, false);
}

m2s_status m2s_export_ply_slice(m2s_ctx* c, const char* path, uint32_t format, float gaussian_std, uint64_t first_row, uint64_t n_rows,
                                uint64_t total_rows) {
    if (!c || !path) return M2S_ERR_INVALID;
    if (first_row > total_rows || n_rows > total_rows - first_row) return fail(c, M2S_ERR_INVALID, "slice exceeds the file");
    return export_rows(c, path, format, gaussian_std, first_row, n_rows, total_rows, true);
}

// RadixSortPass::execute (RadixSortPass.cpp:8-90) on the records of the last conversion.
m2s_status m2s_sort_by_depth(m2s_ctx* c, const float world_to_view[16], uint64_t* out_n) {
    if (!c || !world_to_view) return M2S_ERR_INVALID;
    if (!c->last_records) return fail(c, M2S_ERR_STATE, "no conversion has run and no records were uploaded");
    if (c->records_stale) return fail(c, M2S_ERR_STATE, kStaleMsg);
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = c->last_stored;
    c->sorted_n = 0;
    if (out_n) *out_n = n;
    if (!n) return M2S_OK;
    if (n > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 records");
    if (c->sorted_cap < n) {
        if (c->d_sorted) { (void)hipFree(c->d_sorted); c->d_sorted = nullptr; c->sorted_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_sorted, n * sizeof(m2s_gaussian)));
        c->sorted_cap = n;
    }
    if (c->sort_u32_cap < n) {
        if (c->d_sort_u32) { (void)hipFree(c->d_sort_u32); c->d_sort_u32 = nullptr; c->sort_u32_cap = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_sort_u32, n * 4 * sizeof(uint32_t)));
        c->sort_u32_cap = n;
    }
    const size_t tb = sort_temp_bytes((uint32_t)n);
    if (c->sort_temp_cap < tb) {
        if (c->d_sort_temp) { (void)hipFree(c->d_sort_temp); c->d_sort_temp = nullptr; c->sort_temp_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_sort_temp, std::max<size_t>(tb, 256)));
        c->sort_temp_cap = tb;
    }
    uint32_t* u = c->d_sort_u32;
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, sort_by_depth((const float4*)c->last_records, (uint32_t)n, world_to_view, u, u + n, u + 2 * n, u + 3 * n, c->d_sort_temp,
                            c->sort_temp_cap, (float4*)c->d_sorted, c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->profiling) HIPCHK(c, hipEventElapsedTime(&c->last_sort_ms, c->ev[0], c->ev[1]));
    c->sorted_n = n;
    return M2S_OK;
}

const void* m2s_device_sorted_records(const m2s_ctx* c) { return c && c->sorted_n ? c->d_sorted : nullptr; }
// the keys of those records (uint32, ascending): keys_out of the radix sort
const void* m2s_device_sorted_keys(const m2s_ctx* c) { return c && c->sorted_n ? c->d_sort_u32 + 2 * c->sorted_n : nullptr; }
uint64_t m2s_num_sorted(const m2s_ctx* c) { return c ? c->sorted_n : 0; }
uint32_t m2s_last_resolution(const m2s_ctx* c) { return c ? c->last_R : 0; }

m2s_status m2s_download_sorted(m2s_ctx* c, m2s_gaussian* dst, uint64_t capacity_records) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->sorted_n) return M2S_OK;
    if (!dst) return fail(c, M2S_ERR_INVALID, "dst is NULL");
    if (capacity_records < c->sorted_n) return fail(c, M2S_ERR_CAPACITY, "dst holds fewer records than were sorted");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dst, c->d_sorted, c->sorted_n * sizeof(m2s_gaussian), hipMemcpyDeviceToHost));
    return M2S_OK;
}

float m2s_last_sort_ms(const m2s_ctx* c) { return c ? c->last_sort_ms : 0.0f; }

// Renderer::updateGaussianBuffer after SceneManager::loadPly (guiRendererConcreteMediator.cpp:30-34; glUtils.cpp:676-684):
// host records (e.g. from m2s_read_ply) become the context's current records.
m2s_status m2s_upload_records(m2s_ctx* c, const m2s_gaussian* records, uint64_t n) {
    if (!c || (!records && n)) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t want = std::max<uint64_t>(n, 1);     // an empty upload still yields a valid (empty) record buffer
    if (c->loaded_cap < want) {
        if (c->d_loaded) { (void)hipFree(c->d_loaded); c->d_loaded = nullptr; c->loaded_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_loaded, want * sizeof(m2s_gaussian)));
        c->loaded_cap = want;
    }
    if (n) HIPCHK(c, hipMemcpy(c->d_loaded, records, n * sizeof(m2s_gaussian), hipMemcpyHostToDevice));
    c->last_records = c->d_loaded;
    c->last_total = c->last_stored = n;
    c->records_stale = false;
    c->last_R = 0;      // uploaded records carry no resolutionTarget: m2s_export_ply (scale multiplier = std / R) refuses them
    c->sorted_n = 0;
    c->pp_visible = 0;
    c->sq_n = 0;
    return M2S_OK;
}

// Records that live in DEVICE memory already (e.g. the merged buffer of a multi-GPU exchange) become the context's current
// records without a copy; R = the resolutionTarget they were converted at (m2s_export_ply's scale multiplier).
m2s_status m2s_set_records(m2s_ctx* c, const void* d_records, uint64_t n, uint32_t R) {
    if (!c || (!d_records && n)) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    c->last_records = d_records;
    c->last_total = c->last_stored = n;
    c->last_R = R;
    c->records_stale = false;
    c->sorted_n = 0; c->pp_visible = 0; c->sq_n = 0;
    return M2S_OK;
}

// Room for n records in the context-owned pool (grow-only); *out_ptr = its device address.  For consumers that fill the
// pool themselves (the root of m2s_dist_gather_records) and then call m2s_set_records.
m2s_status m2s_reserve_records(m2s_ctx* c, uint64_t n, void** out_ptr) {
    if (!c || !out_ptr) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    HIPCHK(c, hipSetDevice(c->device));
    const m2s_status s = ensure_records(c, std::max<uint64_t>(n, 1));
    if (s != M2S_OK) return s;
    *out_ptr = c->d_records;
    return M2S_OK;
}

// GaussiansPrepass::execute (GaussiansPrepass.cpp:8-56) + the counter read-back that follows it (RadixSortPass.cpp:18-22).
m2s_status m2s_prepass(m2s_ctx* c, const m2s_prepass_params* p, const void* d_records, uint64_t n, uint64_t* out_visible) {
    if (!c || !p) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (!d_records) {
        if (!c->last_records) return fail(c, M2S_ERR_STATE, "no conversion has run, no records were uploaded and none were passed");
        if (c->records_stale) return fail(c, M2S_ERR_STATE, kStaleMsg);
        d_records = c->last_records;
        n = c->last_stored;
    }
    if (p->resolution_target == 0) return fail(c, M2S_ERR_INVALID, "resolution_target is 0");
    if (p->depth_test_mesh == 1 && p->format == 0 && (!p->depth || !p->depth_w || !p->depth_h))
        return fail(c, M2S_ERR_INVALID, "depth_test_mesh is set but no depth image was passed");
    if (n > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 records");
    HIPCHK(c, hipSetDevice(c->device));
    c->pp_visible = 0;
    c->sq_n = 0;
    if (out_visible) *out_visible = 0;
    if (!n) return M2S_OK;
    if (c->pp_cap < n) {
        if (c->d_quads) { (void)hipFree(c->d_quads); c->d_quads = nullptr; }
        if (c->d_pp_depths) { (void)hipFree(c->d_pp_depths); c->d_pp_depths = nullptr; }
        c->pp_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_quads, n * sizeof(m2s_quad)));
        HIPCHK(c, hipMalloc((void**)&c->d_pp_depths, n * sizeof(float)));
        c->pp_cap = n;
    }
    const uint64_t words = (n + 63) / 64 + 1;          // [0] = the arrival-order counter, [1..] = the look-back chain
    // the chain is tagged with the low 16 bits of a launch counter instead of being cleared per launch; cleared when
    // it is (re)allocated and when the tag wraps (see next_epoch)
    bool clear_chain = false;
    if (c->pp_chain_words < words) {
        if (c->d_pp_chain) { (void)hipFree(c->d_pp_chain); c->d_pp_chain = nullptr; c->pp_chain_words = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_pp_chain, words * sizeof(unsigned long long)));
        c->pp_chain_words = words;
        clear_chain = true;
    }
    const uint32_t epoch = ++c->pp_epoch;
    if (clear_chain || (epoch & 0xFFFFu) == 0)
        HIPCHK(c, hipMemsetAsync(c->d_pp_chain, 0, c->pp_chain_words * sizeof(unsigned long long), c->stream));
    PrepassK k;
    prepass_prepare(*p, n, &k);
    if (p->depth_test_mesh == 1 && p->format == 0) {
        if (p->depth_on_device) k.depth = p->depth;
        else {
            const uint64_t texels = (uint64_t)p->depth_w * p->depth_h;
            if (c->pp_depthtex_cap < texels) {
                if (c->d_pp_depthtex) { (void)hipFree(c->d_pp_depthtex); c->d_pp_depthtex = nullptr; c->pp_depthtex_cap = 0; }
                HIPCHK(c, hipMalloc((void**)&c->d_pp_depthtex, texels * sizeof(float)));
                c->pp_depthtex_cap = texels;
            }
            HIPCHK(c, hipMemcpyAsync(c->d_pp_depthtex, p->depth, texels * sizeof(float), hipMemcpyHostToDevice, c->stream));
            k.depth = c->d_pp_depthtex;
        }
    } else k.depth_test = 0;
    unsigned long long* res = &c->h_total[2 + 2 * M2S_MAX_IN_FLIGHT];
    res[0] = 0; res[1] = 0;
    if (k.arrival_order) HIPCHK(c, hipMemsetAsync(c->d_pp_chain, 0, sizeof(unsigned long long), c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, launch_prepass(k, (const float4*)d_records, (uint32_t)n, (float4*)c->d_quads, c->d_pp_depths, c->d_pp_chain + 1, epoch,
                             c->d_pp_chain, &res[0], reinterpret_cast<uint32_t*>(&res[1]), c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    if (k.arrival_order) HIPCHK(c, hipMemcpyAsync(&res[0], c->d_pp_chain, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->profiling) HIPCHK(c, hipEventElapsedTime(&c->last_prepass_ms, c->ev[0], c->ev[1]));
    if (reinterpret_cast<uint32_t*>(&res[1])[1]) return fail(c, M2S_ERR_HIP, "prepass: look-back chain timed out");
    c->pp_visible = res[0];
    if (out_visible) *out_visible = res[0];
    return M2S_OK;
}

const void* m2s_device_quads(const m2s_ctx* c) { return c && c->pp_visible ? c->d_quads : nullptr; }
const void* m2s_device_prepass_depths(const m2s_ctx* c) { return c && c->pp_visible ? c->d_pp_depths : nullptr; }

m2s_status m2s_download_prepass(m2s_ctx* c, m2s_quad* dst_quads, float* dst_depths, uint64_t capacity) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->pp_visible) return M2S_OK;
    if (capacity < c->pp_visible) return fail(c, M2S_ERR_CAPACITY, "destination holds fewer entries than survived the prepass");
    HIPCHK(c, hipSetDevice(c->device));
    if (dst_quads) HIPCHK(c, hipMemcpy(dst_quads, c->d_quads, c->pp_visible * sizeof(m2s_quad), hipMemcpyDeviceToHost));
    if (dst_depths) HIPCHK(c, hipMemcpy(dst_depths, c->d_pp_depths, c->pp_visible * sizeof(float), hipMemcpyDeviceToHost));
    return M2S_OK;
}

float m2s_last_prepass_ms(const m2s_ctx* c) { return c ? c->last_prepass_ms : 0.0f; }

// RadixSortPass::execute (RadixSortPass.cpp:8-90) on what the last m2s_prepass left behind.
m2s_status m2s_sort_prepass(m2s_ctx* c, uint64_t* out_n) {
    if (!c) return M2S_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t n = c->pp_visible;          // the atomic counter the reference reads back (RadixSortPass.cpp:18-22)
    c->sq_n = 0;
    if (out_n) *out_n = n;
    if (!n) return M2S_OK;
    if (c->sq_cap < n) {
        if (c->d_sorted_quads) { (void)hipFree(c->d_sorted_quads); c->d_sorted_quads = nullptr; c->sq_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_sorted_quads, n * sizeof(m2s_quad)));
        c->sq_cap = n;
    }
    if (c->sort_u32_cap < n) {
        if (c->d_sort_u32) { (void)hipFree(c->d_sort_u32); c->d_sort_u32 = nullptr; c->sort_u32_cap = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_sort_u32, n * 4 * sizeof(uint32_t)));
        c->sort_u32_cap = n;
    }
    const size_t tb = sort_prepass_temp_bytes((uint32_t)n);
    if (c->sort_temp_cap < tb) {
        if (c->d_sort_temp) { (void)hipFree(c->d_sort_temp); c->d_sort_temp = nullptr; c->sort_temp_cap = 0; }
        HIPCHK(c, hipMalloc(&c->d_sort_temp, std::max<size_t>(tb, 256)));
        c->sort_temp_cap = tb;
    }
    uint32_t* u = c->d_sort_u32;
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    HIPCHK(c, sort_prepass(c->d_pp_depths, (const float4*)c->d_quads, (uint32_t)n, u, u + n, c->d_sort_temp, c->sort_temp_cap,
                           (float4*)c->d_sorted_quads, c->stream));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->profiling) HIPCHK(c, hipEventElapsedTime(&c->last_sort_prepass_ms, c->ev[0], c->ev[1]));
    c->sq_n = n;
    return M2S_OK;
}

const void* m2s_device_sorted_quads(const m2s_ctx* c) { return c && c->sq_n ? c->d_sorted_quads : nullptr; }

m2s_status m2s_download_sorted_quads(m2s_ctx* c, m2s_quad* dst, uint64_t capacity) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->sq_n) return M2S_OK;
    if (!dst) return fail(c, M2S_ERR_INVALID, "dst is NULL");
    if (capacity < c->sq_n) return fail(c, M2S_ERR_CAPACITY, "dst holds fewer quads than were sorted");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(dst, c->d_sorted_quads, c->sq_n * sizeof(m2s_quad), hipMemcpyDeviceToHost));
    return M2S_OK;
}

float m2s_last_sort_prepass_ms(const m2s_ctx* c) { return c ? c->last_sort_prepass_ms : 0.0f; }

m2s_status m2s_set_profiling(m2s_ctx* c, int enabled) {
    if (!c) return M2S_ERR_INVALID;
    c->profiling = enabled != 0;
    return M2S_OK;
}

m2s_status m2s_set_pipeline(m2s_ctx* c, int pipeline) {
    if (!c) return M2S_ERR_INVALID;
    if (pipeline < M2S_PIPELINE_AUTO || pipeline > M2S_PIPELINE_SPARSE) return fail(c, M2S_ERR_INVALID, "unknown pipeline");
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (c->pipeline != pipeline) {   // what was remembered about this scene under the old setting no longer applies
        c->rinfo.clear();
        ++c->rinfo_gen;
    }
    c->pipeline = pipeline;
    return M2S_OK;
}

int m2s_last_pipeline(const m2s_ctx* c) { return c ? c->last_pipeline : 0; }

m2s_status m2s_debug_set_launch_counter(m2s_ctx* c, uint32_t value) {
    if (!c) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    c->epoch = value;
    return M2S_OK;
}

m2s_status m2s_set_async_lanes(m2s_ctx* c, int lanes) {
    if (!c) return M2S_ERR_INVALID;
    if (lanes != 1 && lanes != 2) return fail(c, M2S_ERR_INVALID, "lanes must be 1 or 2");
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    c->lanes = lanes;
    return M2S_OK;
}

m2s_status m2s_last_kernel_ms(const m2s_ctx* c, float out_ms[M2S_K_N]) {
    if (!c || !out_ms) return M2S_ERR_INVALID;
    memcpy(out_ms, c->last_ms, sizeof c->last_ms);
    return M2S_OK;
}

}  // extern "C"
