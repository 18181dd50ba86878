// m2s_records.cpp — what happens to the records after the pass: read-back, .ply export (SceneManager::exportPly,
// SceneManager.cpp:651-678), adoption of loaded / merged records.
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
    const bool on_device = format != 0 && !debug_on("M2S_HOST_ENCODE");
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
    ++c->records_epoch;
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
    ++c->records_epoch;
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

}  // extern "C"
