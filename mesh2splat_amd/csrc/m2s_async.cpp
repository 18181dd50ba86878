// m2s_async.cpp — m2s_convert_submit / m2s_convert_wait: a ring of result slots so that consecutive conversions run back to
// back (the reference blocks in glFinish once per conversion, ConversionPass.cpp:54).
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

extern "C" {

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
        const uint32_t n_start = emit2_slices(limit);
        if (c->start_cap >= n_start && c->d_setup) {
            // odd slots of a two-lane context: the second lane, with work buffers of its own — nothing is shared with the
            // conversion before it, so its k_count_scan runs beside that conversion's k_emit2
            bool second_lane = false;
            if (!d_records && (k & 1u) && c->lanes == 2) {
                { const m2s_status s2 = ensure_second_lane(c); if (s2 != M2S_OK) return s2; }
                { const m2s_status s2 = ensure_second_lane_multipass(c, n_start); if (s2 != M2S_OK) return s2; }
                st = c->stream_b;
                d_out = c->d_records_b;
                second_lane = true;
                sl.own_lane = 1;
            }
            if (!second_lane) HIPCHK(c, chain_behind_newest(st));
            if (sl.own_lane >= 0) {
                if (c->buf_R[sl.own_lane] != R) { c->buf_R[sl.own_lane] = R; ++c->buf_gen[sl.own_lane]; }
                sl.gen = c->buf_gen[sl.own_lane];
            }
            unsigned long long* res = &c->h_total[2 + 2 * k];
            res[0] = 0; res[1] = 0;
            { const m2s_status ms_ = enqueue_multipass(c, R, (float4*)d_out, limit, false, res, st, second_lane); if (ms_ != M2S_OK) return ms_; }
            HIPCHK(c, hipEventRecord(sl.done, st));
            c->last_pipeline = M2S_PIPELINE_MULTIPASS;
            if (!second_lane) c->last_submit_stream = st;
            sl.prof = false;
            sl.sync_result = false;
            sl.limit = limit;
            sl.d_out = d_out;
            sl.st = st; sl.shared_work = !second_lane; sl.ri_gen = ri.gen;
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
            { const m2s_status s2 = ensure_second_lane(c); if (s2 != M2S_OK) return s2; }
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
    const bool lean = !sparse && use_lean(c, ri);
    c->last_pipeline = sparse ? M2S_PIPELINE_SPARSE : lean ? M2S_PIPELINE_LEAN : M2S_PIPELINE_TEAM;
    sl.bands_unit = sparse ? kSparseTrianglesPerWorkgroup : 256u;
    if (sparse) launch_sparse(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                              c->d_biglist, c->d_bigmeta, bands_for(c, ri, sl.bands_unit, !second_lane && c->lanes == 1, &sl.wrote_bands), st);
    else if (lean) launch_fused3(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                            c->d_biglist, c->d_bigmeta, bands_for(c, ri, sl.bands_unit, !second_lane && c->lanes == 1, &sl.wrote_bands), batches_for(c, ri), st);
    else launch_fused2(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                       c->d_biglist, c->d_bigmeta, bands_for(c, ri, sl.bands_unit, !second_lane && c->lanes == 1, &sl.wrote_bands), batches_for(c, ri), st);
    if (sl.prof) HIPCHK(c, hipEventRecord(sl.t1, st));
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
        ++c->records_epoch;
        c->records_stale = stale;
        memcpy(c->last_ms, sl.ms, sizeof sl.ms);
        return M2S_OK;
    }
    HIPCHK(c, wait_event(sl.done));
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
    if (sl.wrote_bands && rip) { rip->bands_ready = true; rip->bands_unit = sl.bands_unit; }   // that launch has completed: its run table is in place
    if (total > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 fragments: offsets are 32-bit");
    c->last_total = total;
    c->last_stored = std::min(total, sl.limit);
    c->last_records = sl.d_out;
    ++c->records_epoch;
    c->last_R = sl.R;
    c->records_stale = stale;
    if (out_total) *out_total = total;
    return M2S_OK;
}

}  // extern "C"
