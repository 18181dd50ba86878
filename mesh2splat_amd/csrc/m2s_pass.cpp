// m2s_pass.cpp — the conversion pass driver (== ConversionPass::execute, src/renderer/renderPasses/ConversionPass.cpp:9-68):
// pipeline choice, XCD band tables, the single-pass kernels with their fallbacks, the multi-pass pipeline.
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
// AUTO: below this many fragments per triangle the sparse kernel runs.  Measured against k_fused2 on cube-spheres
// (tools/sparse_crossover.py, profiles/r03/v3_sparse_crossover_*): with 3 M and 6.2 M triangles k_sparse is ahead up to 1.75
// fragments per triangle (a workgroup's stream overflows from ~2.5); with 1 M triangles — 2.5 generations of its 512-triangle
// workgroups — k_fused2 is ahead down to 0.68 at least.
static double sparse_frags_per_triangle(uint32_t n_tri) { return n_tri >= 2000000u ? 1.75 : 0.5; }
// XCD runs (RunInfo, m2s_device.h) for this launch of k_fused2 (unit 256) or k_sparse (unit 512 triangles): the table an earlier
// launch of the same kernel at this R — or the count at upload — left behind; or, if there is none, ask this launch to record
// where every run's output starts (second lane: never asked to — two lanes would race on the table)
static uint32_t band_workgroups(const m2s_ctx* c, uint32_t unit) {
    const uint32_t team = fused2_band_workgroups(c->scene.n_tri);
    return !team ? 0u : unit == 256u ? team : sparse_workgroups(c->scene.n_tri);
}
}  // namespace

namespace m2s_host {
// The single-pass kernel is the workgroup-cooperative one (m2s_fused2.hip / its lean and sparse forms).  A scene whose workgroups did
// not fit its LDS stream at this R (team_off) belongs to the multi-pass pipeline from then on (rinfo_for / decide set ri.multipass):
// the one-wave-per-batch form that used to answer such scenes (k_fused, rounds 1-5) was slower there than the multi-pass pipeline
// and is gone (round 6).
bool use_team(const m2s_ctx*, const m2s_ctx::RInfo& ri) {
    return !ri.team_off;
}
// the team kernel in its lean form (m2s_fused3.hip): LEAN uses it where the scene allows it (m2s_ctx::lean_ok) unless a launch at
// this R overflowed its LDS stream; AUTO only for scenes of more than one generation of workgroups (64 triangles per wave: a fourth
// workgroup per CU does nothing for a launch that fits the GPU once — C2 stand-in 0.0354 (k_fused2) vs 0.0362 ms, config 3 0.1170 vs
// 0.1138, profiles/r05/ab_lean_team_kernel.log) and while few triangles are deferred
bool use_lean(const m2s_ctx* c, const m2s_ctx::RInfo& ri) {
    if (!(use_team(c, ri) && c->lean_ok && !ri.lean_off) || ri.tpw || debug_on("M2S_NO_LEAN")) return false;
    return c->pipeline == M2S_PIPELINE_LEAN || (c->pipeline == M2S_PIPELINE_AUTO && fused_tpw(c->scene.n_tri) == 64u);
}
// ... or its sparse form (m2s_sparse.hip): meshes with fewer fragments than triangles, large enough for 64-triangle batches
bool use_sparse(const m2s_ctx* c, const m2s_ctx::RInfo& ri) {
    return (c->pipeline == M2S_PIPELINE_SPARSE || (c->pipeline == M2S_PIPELINE_AUTO && ri.sparse)) && !ri.sparse_off &&
           sparse_supported(c->scene.n_tri);
}
RunInfo bands_for(const m2s_ctx* c, const m2s_ctx::RInfo& ri, uint32_t unit, bool may_write, bool* writes) {
    RunInfo r{ nullptr, nullptr, 0u, nullptr };
    if (writes) *writes = false;
    const uint32_t n_wg = band_workgroups(c, unit);
    const uint32_t shift = run_shift_for(n_wg);
    if (!n_wg || !shift || !c->d_bands || !c->run_table_words || (unit == 256u && (c->n_batch_tab || ri.tpw)) || debug_on("M2S_NO_BANDS")) return r;
    if ((size_t)n_runs(n_wg, shift) > c->run_table_words) return r;
    unsigned long long* table = c->d_bands + (size_t)ri.band_slot * c->run_table_words;
    r.shift = shift;
    if (ri.bands_ready && ri.bands_unit == unit) {
        r.base = table;
        // heaviest runs first (built by warm_scene from the exact counts at ITS R: the ranking of the runs is a property of the mesh)
        if (c->run_order_unit == unit && c->run_order_shift == shift && !debug_on("M2S_NO_RUN_ORDER")) r.order = c->d_run_order;
    }
    else if (may_write) { r.out = table; if (writes) *writes = true; }
    return r;
}

BatchTable batches_for(const m2s_ctx* c, const m2s_ctx::RInfo& ri) {
    return c->n_batch_tab ? BatchTable{ c->d_batch_first, c->n_batch_tab, 0u } : BatchTable{ nullptr, 0u, ri.tpw };
}

uint64_t resolve_cap(const m2s_ctx* c, uint32_t R) {
    if (c->cap_policy == 0) return 0;
    if (c->cap_policy > 0) return (uint64_t)c->cap_policy;
    // ConversionPass.cpp:21-24 (unsigned int arithmetic wraps)
    const uint32_t mc = std::max<uint32_t>(1u, c->n_meshes_total);
    const uint32_t mx = R * R * 6u * mc;
    return std::min(mx, kMaxGaussiansToSort);
}
}  // namespace m2s_host

// AUTO's decision for this scene at R from its (exact or predicted) fragment count.
// The single-pass kernel wins while triangles are small (it does the per-triangle work once and needs no second
// sweep); with more than ~11 fragments per triangle on average the output-partitioned multi-pass pipeline is
// faster and soon much faster (2.74 M fragments at R = 1024 from 1 M / 250 k / 125 k / 62 k triangles: fused 0.167 /
// 0.138 / 0.323 / 0.626 ms, multi-pass 0.214 / 0.137 / 0.136 / 0.154 ms; tools/auto_probe.py).
// Round 5, the lower edge of the multi-pass range: from 11 to 13 fragments per triangle the team kernel in SMALLER batches (40 triangles
// per wave, so that a workgroup's 160 triangles still fit its 4096-entry LDS stream) beats the multi-pass pipeline — 235 200 triangles
// at 11.6 fragments each: kernels 0.113 against 0.118 ms, one blocking call 0.127 against 0.146 (one launch instead of two;
// profiles/r05/band_probe.jsonl); at 14 fragments per triangle a cube-sphere's densest workgroups overflow the stream even at 40
// (the launch is then repeated by the multi-pass pipeline, which wins from there on anyway: 0.110 ms).  Taken only at the R the
// scene was counted at (R == warm_R: the first conversion after an upload, the reference's case) and only if triangles of more than
// 96 fragments — which a single-pass kernel defers — hold less than an eighth of the fragments: a scene whose MEAN falls in the band
// because it mixes planes with foliage (synth.sponza_like: 16 per triangle) belongs to the multi-pass pipeline and its fine blocks.
static void decide(const m2s_ctx* c, m2s_ctx::RInfo& ri, double frags, uint32_t R) {
    ri.decided = true;
    ri.multipass = frags >= 11.0 * (double)c->scene.n_tri || ri.team_off;
    ri.tpw = 0;
    if (ri.multipass && frags < 13.0 * (double)c->scene.n_tri && R == c->warm_R && c->warm_total > 0 && c->warm_big * 8ull < c->warm_total &&
        fused_tpw(c->scene.n_tri) == 64u && !c->n_batch_tab && c->team_off_R > R && !debug_on("M2S_NO_BAND_TPW")) {
        ri.multipass = false;
        ri.tpw = 40;
    }
    // about as many fragments as triangles, or fewer: many triangles cover no pixel centre, the sparse form drops them cheaply
    // (crossover measured with tools/sparse_probe.py: see DESIGN.md)
    ri.sparse = !ri.multipass && frags < sparse_frags_per_triangle(c->scene.n_tri) * (double)c->scene.n_tri && !debug_on("M2S_NO_SPARSE");
}

static m2s_status ensure_multipass_buffers(m2s_ctx* c, uint64_t limit);

namespace m2s_host {
// Everything a FIRST conversion of a scene used to find out inside its own call — round 3: an exact count, a host round trip,
// the record pool's hipMalloc, a launch without bands: 0.45 ms on config 3 against 0.13 ms for a repeated conversion — is
// found out here, behind the upload's own kernels, at the resolution the caller is about to convert at (m2s_set_resolution_hint;
// default: the last R this context converted at, else 1024).  The reference converts right after SceneManager::loadModel at the
// RenderContext's current resolutionTarget (guiRendererConcreteMediator.cpp:11-29), never twice at one (scene, R): its first
// conversion is the one that counts.  Left behind: the scene's fragments / R^2 (AUTO's decision and the pool size at ANY R
// without touching the device), and for R itself the decision, the XCD run table from the exact counts (k_unit_bases), the batch table of one-generation scenes, the multi-pass work buffers, the record pool.
// The count itself, enqueued by m2s_upload_scene behind the geometry and in front of the texture copies: exact fragments per triangle
// (d_cnt), their scanned partial sums, the total on its way to pinned memory.  warm_scene(counted = true) picks it up after the
// upload's own synchronisation.
m2s_status warm_count_enqueue(m2s_ctx* c, uint32_t R) {
    const SceneDev& sc = c->scene;
    if (!sc.n_tri || R == 0 || R > 4096) return M2S_ERR_INVALID;
    hipStream_t st = c->stream;
    launch_count(sc, R, c->d_cnt, c->d_partials, st);
    launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, st));
    // ... and, on the expectation that AUTO will take the team kernel (units of 256 triangles) — the common case —, the run table and
    // the dispatch order of its first launch at R, which depend on the counts only: if the decision turns out otherwise, warm_scene
    // builds what it needs as before
    c->warm_spec_unit = 0;
    if (!debug_on("M2S_NO_WARM_BANDS")) {
        m2s_ctx::RInfo& ri = rinfo_for(c, R);
        bool writes = false;
        const RunInfo table = bands_for(c, ri, 256u, true, &writes);
        if (writes) {
            const uint32_t n_units = band_workgroups(c, 256u);
            launch_unit_bases(c->d_cnt, c->d_partials, sc.n_tri, 256u, table.shift, table.out, st);
            launch_run_order(table.out, n_runs(n_units, table.shift), c->d_total, c->d_run_order, run_order_slots(n_units, table.shift), st);
            HIPCHK(c, hipGetLastError());
            c->warm_spec_unit = 256u; c->warm_spec_shift = table.shift;
        }
    }
    return M2S_OK;
}

m2s_status warm_scene(m2s_ctx* c, uint32_t R, bool counted) {
    const SceneDev& sc = c->scene;
    if (!sc.n_tri || R == 0 || R > 4096) return M2S_OK;
    hipStream_t st = c->stream;
    if (!counted) {
        const m2s_status s = warm_count_enqueue(c, R);
        if (s != M2S_OK) return s;
        HIPCHK(c, hipStreamSynchronize(st));
    }
    const uint64_t total = c->h_total[0];
    c->frag_per_R2 = (double)total / ((double)R * (double)R);
    c->warm_R = R;
    c->warm_total = total;
    c->warm_big = 0;
    if (total >= 11ull * sc.n_tri && total < 13ull * sc.n_tri) {   // (the only case decide() asks for it)
        HIPCHK(c, hipMemsetAsync(c->d_total, 0, 8, st));
        launch_big_share(c->d_cnt, sc.n_tri, 96u, c->d_total, st);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(&c->h_total[1], c->d_total, 8, hipMemcpyDeviceToHost, st));
        HIPCHK(c, hipStreamSynchronize(st));
        c->warm_big = c->h_total[1];
        c->h_total[1] = 0;
        const unsigned long long tot = total;      // (d_total served as the accumulator: the total goes back, launch_run_order below reads it)
        HIPCHK(c, hipMemcpy(c->d_total, &tot, sizeof tot, hipMemcpyHostToDevice));
    }
    // A scene small enough for ONE generation of workgroups (fused_tpw < 64) lasts as long as its slowest workgroup: cut it into
    // batches of equal estimated work instead of equal triangle counts (C2 stand-in: fragments per workgroup vary 1 : 3 over a
    // cube-sphere face).  Work = 214 per triangle + 140 per fragment (cycles of the triangle phase per 64 triangles and of a strip
    // per 64 fragments, tools/team_timing.py); fragments scale with R^2 everywhere alike, so the table serves every density.
    if (batch_table_capacity(sc.n_tri) && !c->n_batch_tab && !debug_on("M2S_NO_BATCH_TABLE")) {
        try {
            std::vector<uint32_t> cnt(sc.n_tri), first;
            HIPCHK(c, hipMemcpy(cnt.data(), c->d_cnt, (size_t)sc.n_tri * sizeof(uint32_t), hipMemcpyDeviceToHost));
            uint32_t n_target = n_fused_waves(sc.n_tri);
            if (const char* v = debug_env("M2S_BATCH_TARGET")) { const unsigned long q = strtoul(v, nullptr, 10); if (q >= 1 && q <= 8192) n_target = (uint32_t)q; }   // debug: A/B
            double ct = 214.0;
            const double cf = 140.0;
            if (const char* v = debug_env("M2S_BATCH_CT")) ct = atof(v);   // debug: A/B of the work model
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
    m2s_ctx::RInfo& ri = rinfo_for(c, R);
    if (c->pipeline == M2S_PIPELINE_AUTO && !ri.decided) decide(c, ri, (double)total, R);
    const uint64_t cap = resolve_cap(c, R);
    const bool single = c->pipeline != M2S_PIPELINE_MULTIPASS && !ri.multipass;
    // the code object of the pipeline this scene is about to run: loaded here, not inside the first conversion of the process
    if (!debug_on("M2S_NO_PRELOAD")) {
        if (!single) { (void)preload_multipass(); if (!debug_on("M2S_NO_SCRATCH_WARM")) launch_scratch_warm(st); }
        else if (use_sparse(c, ri)) { (void)preload_sparse(); (void)preload_fused2(); }   // (the sparse form falls back to the team on a stream overflow)
        else if (use_lean(c, ri)) { (void)preload_fused3(); }
        else { (void)preload_fused2(); }
        (void)hipGetLastError();
    }
    if (single && (use_sparse(c, ri) || use_team(c, ri)) && !debug_on("M2S_NO_WARM_BANDS")) {
        // the run table of the first launch at R, from the exact counts (the same table a launch without runs leaves behind)
        const uint32_t unit = use_sparse(c, ri) ? kSparseTrianglesPerWorkgroup : 256u;
        bool writes = false;
        const RunInfo table = bands_for(c, ri, unit, true, &writes);
        if (writes) {
            // (already there if the count was enqueued by the upload and the expectation — units of 256 triangles — held)
            if (!(counted && c->warm_spec_unit == unit && c->warm_spec_shift == table.shift)) {
                launch_unit_bases(c->d_cnt, c->d_partials, sc.n_tri, unit, table.shift, table.out, st);
                const uint32_t n_units = band_workgroups(c, unit);
                launch_run_order(table.out, n_runs(n_units, table.shift), c->d_total, c->d_run_order, run_order_slots(n_units, table.shift), st);
                HIPCHK(c, hipGetLastError());
                HIPCHK(c, hipStreamSynchronize(st));
            }
            c->run_order_unit = unit; c->run_order_shift = table.shift;
            ri.bands_ready = true; ri.bands_unit = unit;
        }
    }
    // allocations a first conversion would otherwise make inside its own call; best effort (the conversion reports a failure).
    // Only for a caller that has said at which R it is about to convert (m2s_set_resolution_hint: the command line, the drop-in
    // ConversionPass, the reference's load-then-convert flow) or whose context already owns a record pool: a context that only ever
    // converts into its caller's buffers (m2s_convert_into: the ranks of a multi-GPU job) does not carry 600 MB it never uses.
    const uint64_t want = cap ? cap : std::max<uint64_t>(total, 1);
    if (c->hint_R != 0 || c->d_records != nullptr) {
        if (!single || total >= 8ull * sc.n_tri) (void)ensure_multipass_buffers(c, std::min<uint64_t>(cap ? cap : want, 0xFFFFFFFFull));
        // (touching the pool here does not pay: a hipMemset of the part the first conversion writes made that conversion 0.014 ms
        //  SLOWER on config 3, 0.185 vs 0.171 ms — profiles/r04/first_call_probe.jsonl)
        (void)ensure_records(c, want);
    }
    c->err.clear();
    return M2S_OK;
}
}  // namespace m2s_host

// Multi-pass pipeline: handles every triangle size, output-balanced (m2s_emit2.hip): k_count_scan (count + offsets + per-triangle
// setup records, one kernel) -> k_emit2 (wave-granular).

static m2s_status ensure_multipass_buffers(m2s_ctx* c, uint64_t limit) {
    const uint32_t n_start = emit2_slices(limit);
    if (c->start_cap < n_start) {
        drain_in_flight(c);
        if (c->d_start) { (void)hipFree(c->d_start); c->d_start = nullptr; c->start_cap = 0; }
        const size_t want = std::max<size_t>(std::max<size_t>(n_start, 2 * c->start_cap), 16384);
        HIPCHK(c, hipMalloc((void**)&c->d_start, want * sizeof(uint32_t)));
        c->start_cap = want;
    }
    if (!c->d_setup) {
        HIPCHK(c, hipMalloc(&c->d_setup, setup_bytes(c->scene.n_tri)));
        HIPCHK(c, hipMemsetAsync((char*)c->d_setup + setup_tall_offset(c->scene.n_tri), 0, 16, c->stream));   // the tall-triangle table's slot counter
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return M2S_OK;
}

namespace m2s_host {
// The second lane of asynchronous submissions (m2s_set_async_lanes(2)): its stream, its look-back chain and its record buffer,
// allocated at its first use.  A record buffer that has become too small is replaced after every conversion still writing it has finished.
m2s_status ensure_second_lane(m2s_ctx* c) {
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
    return M2S_OK;
}

// ... and what a MULTI-PASS conversion on that lane works in: offsets, slice starts, TriSetup records, counter (second generation
// of the pipeline only).  With these, k_count_scan of one conversion runs beside k_emit2 of the previous one.
m2s_status ensure_second_lane_multipass(m2s_ctx* c, uint32_t n_start) {
    const size_t np = std::max<size_t>(c->scene.n_tri, 1);
    if (!c->d_off_b) HIPCHK(c, hipMalloc((void**)&c->d_off_b, (np + 1) * sizeof(uint32_t)));
    if (!c->d_setup_b) {
        HIPCHK(c, hipMalloc(&c->d_setup_b, setup_bytes(c->scene.n_tri)));
        HIPCHK(c, hipMemsetAsync((char*)c->d_setup_b + setup_tall_offset(c->scene.n_tri), 0, 16, c->stream_b));
        HIPCHK(c, hipStreamSynchronize(c->stream_b));
    }
    if (!c->d_total_b) HIPCHK(c, hipMalloc((void**)&c->d_total_b, sizeof(unsigned long long)));
    if (c->start_b_cap < n_start) {
        for (uint32_t q = 0; q < c->slot_count; ++q) {   // (a second-lane conversion in flight reads the old table)
            auto& o = c->slot[(c->slot_head + q) % M2S_MAX_IN_FLIGHT];
            if (o.own_lane == 1 && !o.sync_result) (void)hipEventSynchronize(o.done);
        }
        if (c->d_start_b) { (void)hipFree(c->d_start_b); c->d_start_b = nullptr; c->start_b_cap = 0; }
        const size_t want = std::max<size_t>(std::max<size_t>(n_start, c->start_cap), 16384);
        HIPCHK(c, hipMalloc((void**)&c->d_start_b, want * sizeof(uint32_t)));
        c->start_b_cap = want;
    }
    return M2S_OK;
}

// enqueues the pipeline's kernels and the read-back of the counter into *h_res (pinned); no synchronisation
m2s_status enqueue_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, bool prof,
                                    unsigned long long* h_res, hipStream_t st, bool second_lane) {
    const SceneDev& sc = c->scene;
    if (second_lane) {   // the second lane's own work buffers (ensure_second_lane_multipass); second generation only
        uint32_t epoch;
        HIPCHK(c, next_epoch(c, &epoch));
        launch_count_scan(sc, R, c->d_off_b, c->d_start_b, emit2_slices(limit), c->d_chain_b, epoch, c->d_total_b, c->d_setup_b,
                          reinterpret_cast<uint32_t*>(&h_res[1]), &h_res[0], st);
        launch_emit2(sc, R, c->d_off_b, c->d_start_b, c->d_total_b, limit, c->d_setup_b, d_out, st);
        HIPCHK(c, hipGetLastError());
        return M2S_OK;
    }
    uint32_t epoch;
    HIPCHK(c, next_epoch(c, &epoch));
    if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
    launch_count_scan(sc, R, c->d_off, c->d_start, emit2_slices(limit), c->d_chain, epoch, c->d_total, c->d_setup,
                      reinterpret_cast<uint32_t*>(&h_res[1]), &h_res[0], st);
    if (prof) { HIPCHK(c, hipEventRecord(c->ev[1], st)); HIPCHK(c, hipEventRecord(c->ev[3], st)); }
    launch_emit2(sc, R, c->d_off, c->d_start, c->d_total, limit, c->d_setup, d_out, st);
    if (prof) HIPCHK(c, hipEventRecord(c->ev[4], st));
    HIPCHK(c, hipGetLastError());
    // (k_count_scan's last workgroup has written the counter to *h_res itself — no copy behind the pipeline)
    return M2S_OK;
}
}

static m2s_status run_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, hipStream_t st) {
    const bool prof = c->profiling;
    { const m2s_status s = ensure_multipass_buffers(c, limit); if (s != M2S_OK) return s; }
    c->h_total[0] = 0; c->h_total[1] = 0;
    { const m2s_status s = enqueue_multipass(c, R, d_out, limit, prof, c->h_total, st); if (s != M2S_OK) return s; }
    HIPCHK(c, wait_stream(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
    if (c->h_total[1] >> 32) return fail(c, M2S_ERR_HIP, "multi-pass pipeline: look-back chain timed out");
    if (prof) {
        HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_COUNT], c->ev[0], c->ev[1]));
        HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_EMIT], c->ev[3], c->ev[4]));
    }
    return M2S_OK;
}

namespace m2s_host {
m2s_status run_pass(m2s_ctx* c, uint32_t R, void* d_user, uint64_t user_cap, hipStream_t st, uint64_t* out_total,
                           bool from_submit) {
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
        ++c->records_epoch;
        if (out_total) *out_total = 0;
        return M2S_OK;
    }
    m2s_ctx::RInfo& ri = rinfo_for(c, R);

    // ---- AUTO: which pipeline for this scene at this R? ---------------------------------------------
    // The fragment count of a scene is proportional to R^2 (window coordinates scale with R), so the ONE exact count
    // m2s_upload_scene took (warm_scene) decides for every R without touching the device: the threshold is not sharp, and
    // all pipelines produce the same bytes anyway.
    if (c->frag_per_R2 < 0.0) {   // only with the debug switch M2S_NO_WARM (round 3's behaviour: the analysis inside the first conversion)
        const m2s_status s = warm_scene(c, R);
        if (s != M2S_OK) return s;
    }
    const double predicted = c->frag_per_R2 * (double)R * (double)R;
    if (c->pipeline == M2S_PIPELINE_AUTO && !ri.decided) decide(c, ri, predicted, R);

    // ---- where do the records go, and how many may be stored? ------------------------------------
    uint64_t limit;
    float4* d_out;
    if (d_user) {
        limit = cap ? std::min(cap, user_cap) : user_cap;
        d_out = (float4*)d_user;
    } else {
        // unlimited policy: room for the predicted count plus slack; a conversion that still overflows is repeated below
        const uint64_t want = cap ? cap : (uint64_t)(predicted * 1.02) + 4096;
        const m2s_status s = ensure_records(c, want);
        if (s != M2S_OK) return s;
        limit = cap ? cap : c->records_cap;
        d_out = (float4*)c->d_records;
    }
    if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;

    bool wrote_plane = false;      // the last single-pass launch also wrote the position plane (k_sparse, m2s_set_keep_positions)
    for (int round = 0; round < 2; ++round) {
    // ---- run ---------------------------------------------------------------------------------------
    bool done = false;
    if (c->pipeline != M2S_PIPELINE_MULTIPASS && !ri.multipass) {
        // single-pass kernel; triangles too large for its in-workgroup budget are only counted.
        // No memset, no memcpy: the look-back chain is epoch-tagged and the kernel writes the fragment
        // counter and its two status words straight into pinned host memory.
        uint32_t any_big = 0, err = 0;
        bool wrote_bands = false;
        wrote_plane = false;
        for (int attempt = 0; attempt < 4; ++attempt) {
            const bool sparse = use_sparse(c, ri);
            const bool team = !sparse;
            const bool lean = team && use_lean(c, ri);
            c->h_total[0] = 0;
            c->h_total[1] = 0;
            uint32_t epoch;
            HIPCHK(c, next_epoch(c, &epoch));
            if (prof) HIPCHK(c, hipEventRecord(c->ev[5], st));
            const uint32_t unit = sparse ? kSparseTrianglesPerWorkgroup : 256u;
            const RunInfo runs = bands_for(c, ri, unit, true, &wrote_bands);
            // m2s_set_keep_positions: the sparse kernel — the one that runs on scenes of tens of millions of records, where a depth sort
            // is worth preparing for — also writes the records' positions as a 16-byte plane (the context's, grown here if need be)
            float4* plane = nullptr;
            if (sparse && c->keep_positions && st == c->stream) {
                if (c->pos_plane_cap < limit) {
                    if (c->d_pos_plane) { (void)hipFree(c->d_pos_plane); c->d_pos_plane = nullptr; c->pos_plane_cap = 0; }
                    if (hipMalloc(&c->d_pos_plane, limit * 16) == hipSuccess) c->pos_plane_cap = limit; else (void)hipGetLastError();
                }
                c->pos_plane_n = 0;
                plane = (float4*)c->d_pos_plane;
            }
            wrote_plane = plane != nullptr;
            if (sparse) launch_sparse(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                                      c->d_biglist, c->d_bigmeta, runs, st, plane);
            else if (lean) launch_fused3(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                                    c->d_biglist, c->d_bigmeta, runs, batches_for(c, ri), st);
            else launch_fused2(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                               c->d_biglist, c->d_bigmeta, runs, batches_for(c, ri), st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[6], st));
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, wait_stream(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
            if (prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_FUSED], c->ev[5], c->ev[6]));
            any_big = (uint32_t)(c->h_total[1] & 0xFFFFFFFFull);
            err = (uint32_t)(c->h_total[1] >> 32);
            c->last_pipeline = sparse ? M2S_PIPELINE_SPARSE : lean ? M2S_PIPELINE_LEAN : M2S_PIPELINE_TEAM;
            if (!err && wrote_bands) { ri.bands_ready = true; ri.bands_unit = unit; }
            // A launch in runs trusts the run table: at the R the scene was counted at (warm_scene), that table comes from k_count's
            // counts, not from a launch of this kernel.  The two must agree on every triangle; if the totals ever differ, the table is
            // dropped and the conversion repeated in plain order (until now only the parity tests guarded this)
            if (runs.base && !err && R == c->warm_R && c->h_total[0] != c->warm_total && !c->warm_mismatch_seen) {
                c->warm_mismatch_seen = true;
                ri.bands_ready = false;
                if (debug_on("M2S_DEBUG")) fprintf(stderr, "[m2s] run table of R = %u disagrees with the launch (%llu vs %llu fragments): repeated without runs\n",
                                                   R, (unsigned long long)c->warm_total, (unsigned long long)c->h_total[0]);
                HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
                continue;
            }
            if (err && debug_on("M2S_DEBUG"))
                fprintf(stderr, "[m2s] single-pass kernel (%s) reported 0x%x at R = %u: trying the next form\n", sparse ? "sparse" : lean ? "lean" : "team", err, R);
            if (lean && !err && any_big && c->pipeline == M2S_PIPELINE_AUTO) {
                // k_fused3 shades only triangles of at most 8 x 8 pixels itself.  A few deferred ones are what k_emit_big is for;
                // MANY mean the scene at this R belongs to k_fused2, which expands triangles of up to 16 pixel rows in the
                // workgroup: remember that (for this R and every larger one) and convert again
                uint32_t meta[4] = { 0, 0, 0, 0 };
                HIPCHK(c, hipMemcpyAsync(meta, c->d_bigmeta, sizeof meta, hipMemcpyDeviceToHost, st));
                HIPCHK(c, hipStreamSynchronize(st));
                if (meta[0] > 64u && (uint64_t)meta[0] * 256u > sc.n_tri) {
                    ri.lean_off = true;
                    c->lean_off_R = std::min(c->lean_off_R, R);
                    HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, sizeof meta, st));
                    continue;
                }
            }
            if (!err) break;
            if (team && ri.tpw) { ri.tpw = 0; break; }   // a scene of the 11-18 band whose workgroups overflow even in small batches: multi-pass (below)
            // a workgroup's fragments did not fit the kernel's LDS stream (or a wait timed out): sparse -> team, lean -> team,
            // team -> the multi-pass pipeline, which has no such limit (below: err != 0).  Remember it for this scene and R, forget
            // what the aborted launch listed, try again.
            // (error value 2 = "entries do not fit": true of every larger R as well)
            if (sparse) { ri.sparse_off = true; if ((err & 0xFu) == 2u) c->sparse_off_R = std::min(c->sparse_off_R, R); }
            else if (lean) { ri.lean_off = true; if ((err & 0xFu) == 2u) c->lean_off_R = std::min(c->lean_off_R, R); }
            else { ri.team_off = true; if ((err & 0xFu) == 2u) c->team_off_R = std::min(c->team_off_R, R); break; }
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
        }
        done = true;
        // a clean single-kernel conversion: the same scene at the same R can be submitted asynchronously from now on
        ri.async_ok = !err && !any_big;
        if (err) {
            // The team kernel's workgroups do not fit their LDS stream at this R, or the bounded look-back spin gave up (never
            // observed; would need a dispatcher that starves earlier workgroups): the multi-pass pipeline, which has neither limit.
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
        m2s_status s = run_multipass(c, R, d_out, limit, st);
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
    ++c->records_epoch;
    // the plane is these records' if the sparse kernel wrote every one of them (no deferred triangle went to k_emit_big, no fallback ran)
    if (wrote_plane && c->last_pipeline == M2S_PIPELINE_SPARSE && !c->h_total[1] && c->d_pos_plane) {
        c->pos_plane_of = d_out; c->pos_plane_n = c->last_stored; c->pos_plane_epoch = c->records_epoch;
    }
    if (!d_user) { if (c->buf_R[0] != R) { c->buf_R[0] = R; ++c->buf_gen[0]; } }
    if (out_total) *out_total = total;
    return M2S_OK;
}
}  // namespace m2s_host

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

}  // extern "C"
