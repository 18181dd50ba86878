// m2s_context.cpp — context lifetime, policy setters, the record pool and the launch tags of the look-back chains.
#include "m2s_ctx.h"
#include "m2s_ply.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <tuple>
#include <vector>

using namespace m2s;
using namespace m2s_host;

namespace {
thread_local std::string g_create_error = "";
}  // namespace

namespace m2s_host {
void free_scene(m2s_ctx* c) {
    if (c->tri_mem) (void)hipFree(c->tri_mem);
    if (c->scene_arena) (void)hipFree(c->scene_arena);
    if (c->d_chain_b) (void)hipFree(c->d_chain_b);
    if (c->d_setup) (void)hipFree(c->d_setup);
    if (c->d_off_b) (void)hipFree(c->d_off_b);
    if (c->d_start_b) (void)hipFree(c->d_start_b);
    if (c->d_setup_b) (void)hipFree(c->d_setup_b);
    if (c->d_total_b) (void)hipFree(c->d_total_b);
    c->d_chain_b = nullptr; c->d_setup = nullptr;
    c->d_off_b = nullptr; c->d_start_b = nullptr; c->start_b_cap = 0; c->d_setup_b = nullptr; c->d_total_b = nullptr;
    c->tri_mem = nullptr; c->scene_arena = nullptr;
    // everything below lived inside the arena
    c->d_meshes = nullptr; c->d_mesh_first = nullptr;
    c->d_cnt = c->d_off = c->d_partials = nullptr;
    c->d_chain = nullptr; c->d_biglist = nullptr; c->d_bigmeta = nullptr; c->d_bands = nullptr; c->run_table_words = 0; c->d_run_order = nullptr; c->run_order_unit = 0;
    c->d_batch_first = nullptr; c->n_batch_tab = 0; c->chain_words = 0;
    c->rinfo.clear();
    ++c->rinfo_gen;
    c->frag_per_R2 = -1.0;
    c->warm_R = 0; c->warm_total = 0; c->warm_mismatch_seen = false;
    c->sparse_off_R = c->team_off_R = c->lean_off_R = UINT32_MAX;
    c->lean_ok = false;
    c->scene = SceneDev{};
    c->has_scene = false;
}

// What is remembered about this scene at resolution R (created on first use; the table is bounded: a slider dragged
// through hundreds of densities simply starts over).
m2s_ctx::RInfo& rinfo_for(m2s_ctx* c, uint32_t R) {
    auto it = c->rinfo.find(R);
    if (it != c->rinfo.end()) return it->second;
    // (a full table starts over; submissions still in flight remember the generation they were made under, so that a band
    //  slot which now belongs to another density is never marked ready on their behalf: m2s_convert_wait)
    // (the slots' run tables are about to be re-used by other densities: nothing in flight may still be reading one)
    if (c->rinfo.size() >= (size_t)kBandSlots) { drain_in_flight(c); c->rinfo.clear(); ++c->rinfo_gen; }
    m2s_ctx::RInfo ri;
    ri.gen = c->rinfo_gen;
    ri.band_slot = (int)c->rinfo.size();
    ri.sparse_off = R >= c->sparse_off_R;
    ri.team_off = R >= c->team_off_R;
    if (ri.team_off) ri.multipass = true;       // (decide() keeps it; a forced single-pass setting hands such an R to the multi-pass pipeline)
    ri.lean_off = R >= c->lean_off_R;
    return c->rinfo.emplace(R, ri).first->second;
}

static bool spin_waits() { static const bool on = !debug_on("M2S_NO_SPIN"); return on; }
template <class Query, class Block>
static hipError_t spin_then_block(Query query, Block block) {
    if (spin_waits()) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t i = 0;; ++i) {
            const hipError_t e = query();
            if (e != hipErrorNotReady) return e;
            if ((i & 15u) == 15u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(400)) break;
        }
        (void)hipGetLastError();   // (hipErrorNotReady is sticky in hipGetLastError)
    }
    return block();
}
hipError_t wait_stream(hipStream_t st) { return spin_then_block([&] { return hipStreamQuery(st); }, [&] { return hipStreamSynchronize(st); }); }
hipError_t wait_event(hipEvent_t ev) { return spin_then_block([&] { return hipEventQuery(ev); }, [&] { return hipEventSynchronize(ev); }); }

// every conversion still in flight has finished when this returns (their slots stay queued for m2s_convert_wait)
void drain_in_flight(m2s_ctx* c) {
    for (uint32_t k = 0; k < c->slot_count; ++k) {
        auto& sl = c->slot[(c->slot_head + k) % M2S_MAX_IN_FLIGHT];
        if (!sl.sync_result) (void)hipEventSynchronize(sl.done);
    }
}

// Chain words carry a 16-bit launch tag instead of being cleared per launch.  The two single-pass kernels use different
// numbers of words, so a word one of them left behind could read as freshly published 65 536 launches later: when the
// tag wraps, everything in flight is drained and both chains are cleared (once per ~10 s of back-to-back conversions).
hipError_t next_epoch(m2s_ctx* c, uint32_t* out) {
    const uint32_t e = ++c->epoch;
    *out = e;
    if ((e & 0xFFFFu) != 0) return hipSuccess;
    const size_t bytes = std::max<size_t>(c->chain_words, 1) * sizeof(unsigned long long);
    hipError_t r = hipDeviceSynchronize();
    if (r == hipSuccess && c->d_chain) r = hipMemset(c->d_chain, 0, bytes);
    if (r == hipSuccess && c->d_chain_b) r = hipMemset(c->d_chain_b, 0, bytes);
    return r;
}

// The context-owned record buffer is a grow-only pool.  The reference re-creates its SSBO whenever the cap changes
// (ConversionPass.cpp:25-33), i.e. on every move of the density slider; here a change of R costs no allocation: the pool
// doubles until it reaches the largest size the cap policy can ask for (7 M records = 672 MB for the reference formula).
m2s_status ensure_records(m2s_ctx* c, uint64_t want) {
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
}  // namespace m2s_host

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
    if ((e = hipHostMalloc((void**)&c->h_total, m2s_ctx::kPinnedWords * sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", e);
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    for (auto& ev : c->stage_ev)
        if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    unsigned done_flags = hipEventDisableTiming;   // (completion only: its time is never asked for)
    if (const char* v = debug_env("M2S_DONE_EVENT_FLAGS")) {   // debug: A/B of event kinds; anything but a combination of the known bits is ignored
        const unsigned f = (unsigned)strtoul(v, nullptr, 0);
        if ((f & ~(hipEventBlockingSync | hipEventDisableTiming | hipEventReleaseToDevice | hipEventReleaseToSystem)) == 0u) done_flags = f;
    }
    for (auto& sl : c->slot)
        if ((e = hipEventCreateWithFlags(&sl.done, done_flags)) != hipSuccess || (e = hipEventCreate(&sl.t0)) != hipSuccess ||
            (e = hipEventCreate(&sl.t1)) != hipSuccess)
            return bail("hipEventCreate", e);
    *out_ctx = c;
    return M2S_OK;
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
    if (c->d_pos_plane) (void)hipFree(c->d_pos_plane);
    if (c->d_sort_temp) (void)hipFree(c->d_sort_temp);
    if (c->d_total) (void)hipFree(c->d_total);
    if (c->h_total) (void)hipHostFree(c->h_total);
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

m2s_status m2s_set_profiling(m2s_ctx* c, int enabled) {
    if (!c) return M2S_ERR_INVALID;
    c->profiling = enabled != 0;
    return M2S_OK;
}

m2s_status m2s_set_pipeline(m2s_ctx* c, int pipeline) {
    if (!c) return M2S_ERR_INVALID;
    if (pipeline < M2S_PIPELINE_AUTO || pipeline > M2S_PIPELINE_LEAN) return fail(c, M2S_ERR_INVALID, "unknown pipeline");
    if (pipeline == 2) return fail(c, M2S_ERR_INVALID, "pipeline 2 (the one-wave-per-batch kernel of rounds 1-5) no longer exists: use M2S_PIPELINE_TEAM");
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (c->pipeline != pipeline) {   // what was remembered about this scene under the old setting no longer applies
        c->rinfo.clear();
        ++c->rinfo_gen;
    }
    c->pipeline = pipeline;
    return M2S_OK;
}

int m2s_last_pipeline(const m2s_ctx* c) { return c ? c->last_pipeline : 0; }

int m2s_positions_ready(const m2s_ctx* c) {
    return c && c->d_pos_plane && c->last_records && c->pos_plane_of == c->last_records && c->pos_plane_n == c->last_stored &&
           c->pos_plane_epoch == c->records_epoch && c->last_stored ? 1 : 0;
}

m2s_status m2s_set_keep_positions(m2s_ctx* c, int enabled) {
    if (!c) return M2S_ERR_INVALID;
    c->keep_positions = enabled != 0;
    return M2S_OK;
}

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
