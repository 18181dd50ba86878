// m2s_ctx.h — the context behind the C ABI (include/m2s.h) and the helpers its translation units share.  Internal.
//   m2s_context.cpp  lifetime, setters, record pool, launch tags          m2s_upload.cpp   m2s_upload_scene / m2s_prepare
//   m2s_pass.cpp     the conversion pass driver (== ConversionPass::execute)   m2s_async.cpp    m2s_convert_submit / _wait
//   m2s_records.cpp  read-back, .ply export, record adoption               m2s_viewer.cpp   depth sort, viewer prepass
#pragma once
#include "../../include/m2s.h"
#include "m2s_device.h"

#include <cstdint>
#include <cstdlib>
#include <map>
#include <string>

namespace m2s_host {
constexpr uint32_t kMaxGaussiansToSort = 7000000u;  // RenderPass.hpp:9
constexpr uint64_t kMaxTriangles = (1ull << 28) - 1;  // 32-bit byte offsets into the 16 B/triangle planes
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline const char* const kStaleMsg = "the records of the conversion last waited for have been overwritten by a later submission at another R "
                                     "(wait for it, or submit into your own buffers)";
}  // namespace m2s_host

constexpr int kBandSlotsMax = 64;   // (scene, R) entries remembered per context: band tables, decisions
struct m2s_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    void* tri_mem = nullptr;
    m2s::SceneDev scene{};
    m2s::MeshParams* d_meshes = nullptr;
    uint32_t* d_mesh_first = nullptr;
    uint32_t n_meshes_total = 0;
    bool has_scene = false;
    bool lean_ok = false;                   // every mesh of the scene samples its combo texture or no map at all: k_fused3 may run
    uint64_t range_first = 0, range_count = UINT64_MAX;

    // work buffers (sized by the scene)
    uint32_t* d_cnt = nullptr;
    uint32_t* d_off = nullptr;
    uint32_t* d_partials = nullptr;
    uint32_t* d_start = nullptr;
    size_t start_cap = 0;
    unsigned long long* d_total = nullptr;
    unsigned long long* h_total = nullptr;  // pinned, kPinnedWords words: [0] = fragment counter, [1] = status words of the fused kernel,
                                            // [2 + 2k], [3 + 2k] = the same of in-flight slot k, then one word each for the prepass and the depth sort
    static constexpr int kPinnedPrepass = 2 + 2 * M2S_MAX_IN_FLIGHT;   // m2s_prepass: TWO words — survivors, then the status words of its look-back
    static constexpr int kPinnedSortMM = 4 + 2 * M2S_MAX_IN_FLIGHT;    // the depth sorts: {min, max} of the keys (two 32-bit words); NOT the prepass's second word —
                                                                       // m2s_prepass_sorted uses both inside one call
    static constexpr int kPinnedWords = 6 + 2 * M2S_MAX_IN_FLIGHT;                  // (the sorts' word and the one behind it: {min, max}, {survivors, clash})
    unsigned long long* d_chain = nullptr;  // look-back chain of the fused kernel, one word per wave
    m2s::BigItem* d_biglist = nullptr;           // triangles deferred by the fused kernel (capacity: triangles in range)
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
        bool team_off = false;     // k_fused2 reported a workgroup that did not fit its LDS stream: the multi-pass pipeline from this R on
        uint32_t tpw = 0;          // AUTO: k_fused2 in batches of this many triangles (0: fused_tpw) — the 11-18 fragments-per-triangle band
        bool lean_off = false;     // k_fused3 overflowed its LDS stream or deferred many triangles at this R: use k_fused2
        bool async_ok = false;     // a completed conversion needed no host decision between kernels
        bool mp_ready = false;     // a multi-pass conversion has completed (its work buffers are sized)
        bool bands_ready = false;  // the run table of this R (slot band_slot of d_bands: where every run's output starts) is in place ...
        uint32_t bands_unit = 0;   // ... in units of this many triangles (256: recorded by a k_fused2 launch, 512: k_sparse)
        int band_slot = 0;
        uint32_t gen = 0;          // generation of the table this entry belongs to (the table starts over when it is full)
    };
    // smallest R at which a workgroup of k_sparse / k_fused2 did not fit its LDS stream: a property of the SCENE (fragments grow
    // with R), so a new R above it starts with the next form at once instead of re-discovering the overflow
    uint32_t sparse_off_R = UINT32_MAX, team_off_R = UINT32_MAX, lean_off_R = UINT32_MAX;   // (lean: k_fused3 -> k_fused2)
    uint32_t rinfo_gen = 0;
    std::map<uint32_t, RInfo> rinfo;
    double frag_per_R2 = -1.0;              // fragments / R^2 of this scene: from the exact count m2s_upload_scene takes (warm_scene), refreshed by every conversion
    uint64_t warm_total = 0;                // fragments of the scene at warm_R (exact: warm_scene's count)
    uint64_t warm_big = 0;                  // ... of which in triangles of more than 96 fragments (what a single-pass kernel would defer)
    bool warm_mismatch_seen = false;        // a launch in runs disagreed with that count once (run_pass): not retried again
    uint32_t hint_R = 0;                    // m2s_set_resolution_hint: the R the next upload prepares for (0: the last R converted at, else 1024)
    uint32_t warm_R = 0;                    // the R the resident scene was prepared for
    uint32_t warm_spec_unit = 0, warm_spec_shift = 0;   // run table + dispatch order enqueued with the upload's count, for units of this many triangles (0: none)
    unsigned long long* d_bands = nullptr;  // kBandSlots run tables of run_table_words words each (RunInfo, m2s_device.h)
    size_t run_table_words = 0;
    uint32_t* d_run_order = nullptr;        // dispatch order of the runs (launch_run_order), built by warm_scene from the exact counts ...
    uint32_t run_order_unit = 0, run_order_shift = 0;   // ... for runs of (1 << shift) units of this many triangles (0: none)
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
    // ... and multi-pass conversions too: the second lane's own offsets / slice starts / TriSetup records / counter (allocated at
    // its first multi-pass submission), so that k_count_scan of conversion i + 1 runs beside k_emit2 of conversion i
    uint32_t* d_off_b = nullptr;
    uint32_t* d_start_b = nullptr;
    size_t start_b_cap = 0;
    void* d_setup_b = nullptr;
    unsigned long long* d_total_b = nullptr;
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
    uint32_t sorted_key_offset = 0;   // keys_out holds key - this (m2s_device_sorted_keys adds it back once)
    void* d_sort_temp = nullptr;
    size_t sort_temp_cap = 0;
    uint64_t sort_u32_cap = 0;
    float last_sort_ms = 0.0f;
    float last_sort_stage_ms[3] = { 0, 0, 0 };   // keys | radix sort | gather of the last m2s_sort_by_depth (profiling on)
    // the positions of the current records as a compact plane (16 B each), left behind by the first depth sort after the records
    // changed and used for the keys of every later one (m2s_sort.hip); valid for (pos_plane_of, pos_plane_n, pos_plane_epoch)
    void* d_pos_plane = nullptr;
    uint64_t pos_plane_cap = 0, pos_plane_n = 0, pos_plane_epoch = 0;
    const void* pos_plane_of = nullptr;
    bool keep_positions = false;      // m2s_set_keep_positions: conversions leave the plane behind themselves where their kernel can (k_sparse)
    uint64_t records_epoch = 1;       // advances whenever the records behind last_records may have changed
    // viewer prepass (m2s_prepass): survivors, their depths, the look-back chain of its kernel, a copy of the depth image
    void* d_quads = nullptr;
    float* d_pp_depths = nullptr;
    uint64_t pp_cap = 0, pp_depths_cap = 0, pp_visible = 0;
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
    float last_upload_ms[5] = { 0, 0, 0, 0, 0 }; // [0] total, [1] geometry, [2] textures + mips + combo, [3] allocations, [4] warm_scene
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

namespace m2s_host {
inline m2s_status fail(m2s_ctx* c, m2s_status s, const std::string& msg) {
    if (c) c->err = msg;
    return s;
}
constexpr int kBandSlots = kBandSlotsMax;

// m2s_context.cpp
// Waiting for a conversion.  The runtime's hipStreamSynchronize / hipEventSynchronize hand the caller back 10-18 us after the last
// kernel has finished — 8-13 % of a 0.12 ms conversion, the reference's glFinish (ConversionPass.cpp:54).  These poll the same
// completion signal (hipStreamQuery / hipEventQuery: the stream's own "all work done") for up to 400 us and only then block.
hipError_t wait_stream(hipStream_t st);
hipError_t wait_event(hipEvent_t ev);
void free_scene(m2s_ctx* c);
m2s_ctx::RInfo& rinfo_for(m2s_ctx* c, uint32_t R);
void drain_in_flight(m2s_ctx* c);          // every conversion still in flight has finished when this returns
hipError_t next_epoch(m2s_ctx* c, uint32_t* out);
m2s_status ensure_records(m2s_ctx* c, uint64_t want);
// m2s_upload.cpp
m2s_status ensure_stage(m2s_ctx* c);
// m2s_pass.cpp
bool use_team(const m2s_ctx* c, const m2s_ctx::RInfo& ri);
bool use_lean(const m2s_ctx* c, const m2s_ctx::RInfo& ri);   // the team kernel in its lean form (k_fused3)
bool use_sparse(const m2s_ctx* c, const m2s_ctx::RInfo& ri);
m2s::RunInfo bands_for(const m2s_ctx* c, const m2s_ctx::RInfo& ri, uint32_t unit, bool may_write, bool* writes);
m2s::BatchTable batches_for(const m2s_ctx* c, const m2s_ctx::RInfo& ri);
uint64_t resolve_cap(const m2s_ctx* c, uint32_t R);
m2s_status warm_scene(m2s_ctx* c, uint32_t R, bool counted = false);   // called by m2s_upload_scene once the scene is resident
m2s_status warm_count_enqueue(m2s_ctx* c, uint32_t R);                  // ... its exact count, enqueued earlier (in front of the texture copies)
m2s_status enqueue_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, bool prof,
                             unsigned long long* h_res, hipStream_t st, bool second_lane = false);
m2s_status ensure_second_lane(m2s_ctx* c);                       // stream, chain and record buffer of the second lane
m2s_status ensure_second_lane_multipass(m2s_ctx* c, uint32_t n_start);   // ... and its multi-pass work buffers
m2s_status run_pass(m2s_ctx* c, uint32_t R, void* d_user, uint64_t user_cap, hipStream_t st, uint64_t* out_total,
                    bool from_submit = false);
}  // namespace m2s_host
