// m2s_viewer.cpp — the two viewer passes that consume the records: depth sort (RadixSortPass.cpp:8-90) and prepass
// (GaussiansPrepass.cpp:8-56).
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
    if (c->pos_plane_cap < n) {
        if (c->d_pos_plane) { (void)hipFree(c->d_pos_plane); c->d_pos_plane = nullptr; c->pos_plane_cap = 0; }
        c->pos_plane_n = 0;
        if (hipMalloc(&c->d_pos_plane, n * 16) == hipSuccess) c->pos_plane_cap = n;   // (without it every sort reads the records: slower, not wrong)
        else (void)hipGetLastError();
    }
    const bool plane_valid = c->d_pos_plane && c->pos_plane_of == c->last_records && c->pos_plane_n == n && c->pos_plane_epoch == c->records_epoch;
    uint32_t* u = c->d_sort_u32;     // keys_in | (unused) | keys_out | vals_out
    HIPCHK(c, sort_by_depth((const float4*)c->last_records, (uint32_t)n, world_to_view, u, u + 2 * n, u + 3 * n, c->d_sort_temp,
                            c->sort_temp_cap, (float4*)c->d_sorted, (float4*)c->d_pos_plane, plane_valid, c->profiling ? c->ev : nullptr, c->stream,
                            &c->sorted_key_offset, reinterpret_cast<uint32_t*>(&c->h_total[m2s_ctx::kPinnedSortMM])));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->d_pos_plane) { c->pos_plane_of = c->last_records; c->pos_plane_n = n; c->pos_plane_epoch = c->records_epoch; }
    if (c->profiling) {
        for (int k = 0; k < 3; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_sort_stage_ms[k], c->ev[k], c->ev[k + 1]));
        HIPCHK(c, hipEventElapsedTime(&c->last_sort_ms, c->ev[0], c->ev[3]));
    }
    c->sorted_n = n;
    return M2S_OK;
}

const void* m2s_device_sorted_records(const m2s_ctx* c) { return c && c->sorted_n ? c->d_sorted : nullptr; }
// the keys of those records (uint32, ascending): keys_out of the radix sort
// (the sort runs over the bits in which the keys differ, on `key - smallest key`: the smallest key is added back here, once, when
//  somebody asks for the keys — the distributed sort does; a frame's sort + gather does not)
const void* m2s_device_sorted_keys(const m2s_ctx* cc) {
    m2s_ctx* c = const_cast<m2s_ctx*>(cc);
    if (!c || !c->sorted_n) return nullptr;
    uint32_t* keys = c->d_sort_u32 + 2 * c->sorted_n;
    if (c->sorted_key_offset) {
        if (hipSetDevice(c->device) != hipSuccess) return nullptr;
        launch_add_to_keys(keys, (uint32_t)c->sorted_n, c->sorted_key_offset, c->stream);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return nullptr;
        c->sorted_key_offset = 0;
    }
    return keys;
}
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
m2s_status m2s_last_sort_stage_ms(const m2s_ctx* c, float out_ms[3]) {
    if (!c || !out_ms) return M2S_ERR_INVALID;
    memcpy(out_ms, c->last_sort_stage_ms, sizeof c->last_sort_stage_ms);
    return M2S_OK;
}

// GaussiansPrepass::execute (GaussiansPrepass.cpp:8-56) + the counter read-back that follows it (RadixSortPass.cpp:18-22).
// sorted = m2s_prepass_sorted: the depth sort of RadixSortPass::execute taken FIRST, as a permutation of the records by the depth bits this
// prepass stores; the prepass then reads the records through it and appends its survivors in that order, straight into the sorted-quads buffer.
static m2s_status prepass_impl(m2s_ctx* c, const m2s_prepass_params* p, const void* d_records, uint64_t n, uint64_t* out_visible, bool sorted) {
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
    if (!sorted && c->pp_cap < n) {
        if (c->d_quads) { (void)hipFree(c->d_quads); c->d_quads = nullptr; }
        c->pp_cap = 0;
        HIPCHK(c, hipMalloc(&c->d_quads, n * sizeof(m2s_quad)));
        c->pp_cap = n;
    }
    if (c->pp_depths_cap < n) {
        if (c->d_pp_depths) { (void)hipFree(c->d_pp_depths); c->d_pp_depths = nullptr; }
        c->pp_depths_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_pp_depths, n * sizeof(float)));
        c->pp_depths_cap = n;
    }
    const uint32_t* perm = nullptr;
    if (sorted) {   // room for the sorted survivors, the keys / permutation, the radix sort's work area and the position plane
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
        const size_t tb = sort_temp_bytes((uint32_t)n);
        if (c->sort_temp_cap < tb) {
            if (c->d_sort_temp) { (void)hipFree(c->d_sort_temp); c->d_sort_temp = nullptr; c->sort_temp_cap = 0; }
            HIPCHK(c, hipMalloc(&c->d_sort_temp, std::max<size_t>(tb, 256)));
            c->sort_temp_cap = tb;
        }
        if (c->pos_plane_cap < n) {
            if (c->d_pos_plane) { (void)hipFree(c->d_pos_plane); c->d_pos_plane = nullptr; c->pos_plane_cap = 0; }
            c->pos_plane_n = 0;
            if (hipMalloc(&c->d_pos_plane, n * 16) == hipSuccess) c->pos_plane_cap = n;   // (without it every frame's keys read the records: slower, not wrong)
            else (void)hipGetLastError();
        }
        c->sorted_n = 0;          // (the key / permutation words are shared with m2s_sort_by_depth: its sorted keys are gone)
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
    if (sorted) k.arrival_order = 0;            // the order of the survivors IS the result
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
    unsigned long long* res = &c->h_total[m2s_ctx::kPinnedPrepass];
    res[0] = 0; res[1] = 0;
    if (k.arrival_order) HIPCHK(c, hipMemsetAsync(c->d_pp_chain, 0, sizeof(unsigned long long), c->stream));
    // Without a depth image the prepass's only test is the frustum test, a function of the position: the sort applies it to the keys
    // (view_project, the prepass's own function), the culled records sort behind the survivors, and the prepass runs over the survivors
    // alone — dense: no compaction, no look-back, no read of a record that is not drawn.
    bool dense = false;
    uint32_t n_run = (uint32_t)n;
    if (sorted) {
        const bool plane_valid = c->d_pos_plane && c->pos_plane_of == c->last_records && c->pos_plane_n == n && c->pos_plane_epoch == c->records_epoch;
        uint32_t* u = c->d_sort_u32;     // keys_in | (unused) | keys_out | vals_out = the permutation
        uint32_t* pinned4 = reinterpret_cast<uint32_t*>(&c->h_total[m2s_ctx::kPinnedSortMM]);
        bool cull = k.depth_test == 0u, clash = false;
        HIPCHK(c, sort_prepass_permutation((const float4*)d_records, (uint32_t)n, k.M, k.V, k.P, cull, u, u + 2 * n, u + 3 * n, c->d_sort_temp, c->sort_temp_cap,
                                           (float4*)c->d_pos_plane, plane_valid, c->profiling ? c->ev : nullptr, c->stream, pinned4, &n_run, &clash));
        if (c->d_pos_plane) { c->pos_plane_of = c->last_records; c->pos_plane_n = n; c->pos_plane_epoch = c->records_epoch; }
        if (cull && clash) {   // a survivor whose depth bits ARE the marker of the culled ones (a NaN with that payload): sort everything, compact in the prepass
            cull = false;
            HIPCHK(c, sort_prepass_permutation((const float4*)d_records, (uint32_t)n, k.M, k.V, k.P, false, u, u + 2 * n, u + 3 * n, c->d_sort_temp, c->sort_temp_cap,
                                               (float4*)c->d_pos_plane, true, c->profiling ? c->ev : nullptr, c->stream, pinned4, &n_run, &clash));
        }
        dense = cull;
        perm = u + 3 * n;
    }
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    if (n_run) HIPCHK(c, launch_prepass(k, (const float4*)d_records, n_run, (float4*)(sorted ? c->d_sorted_quads : c->d_quads), c->d_pp_depths, c->d_pp_chain + 1,
                                        epoch, c->d_pp_chain, &res[0], reinterpret_cast<uint32_t*>(&res[1]), c->stream, perm, dense));
    if (c->profiling) HIPCHK(c, hipEventRecord(c->ev[4], c->stream));
    if (k.arrival_order) HIPCHK(c, hipMemcpyAsync(&res[0], c->d_pp_chain, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->profiling) {
        HIPCHK(c, hipEventElapsedTime(&c->last_prepass_ms, c->ev[3], c->ev[4]));
        if (sorted) {
            for (int j = 0; j < 2; ++j) HIPCHK(c, hipEventElapsedTime(&c->last_sort_stage_ms[j], c->ev[j], c->ev[j + 1]));
            c->last_sort_stage_ms[2] = c->last_prepass_ms;
            HIPCHK(c, hipEventElapsedTime(&c->last_sort_ms, c->ev[0], c->ev[4]));
        }
    }
    if (reinterpret_cast<uint32_t*>(&res[1])[1] == 2u) return fail(c, M2S_ERR_HIP, "prepass_sorted: the sort's frustum test and the prepass's disagree (internal error)");
    if (reinterpret_cast<uint32_t*>(&res[1])[1]) return fail(c, M2S_ERR_HIP, "prepass: look-back chain timed out");
    if (dense) res[0] = n_run;                  // (the survivors were counted by the sort; the dense prepass appends nothing)
    if (sorted) c->sq_n = res[0]; else c->pp_visible = res[0];
    if (out_visible) *out_visible = res[0];
    return M2S_OK;
}

m2s_status m2s_prepass(m2s_ctx* c, const m2s_prepass_params* p, const void* d_records, uint64_t n, uint64_t* out_visible) {
    return prepass_impl(c, p, d_records, n, out_visible, false);
}

// GaussiansPrepass::execute + RadixSortPass::execute (GaussiansPrepass.cpp:8-56, RadixSortPass.cpp:8-90) of one frame as ONE pass over the
// records: == m2s_prepass (input order) followed by m2s_sort_prepass, byte for byte, in m2s_device_sorted_quads.
m2s_status m2s_prepass_sorted(m2s_ctx* c, const m2s_prepass_params* p, uint64_t* out_visible) {
    return prepass_impl(c, p, nullptr, 0, out_visible, true);
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

}  // extern "C"
