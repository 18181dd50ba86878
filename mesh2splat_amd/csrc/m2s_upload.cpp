// m2s_upload.cpp — m2s_upload_scene (== SceneManager::setupMeshBuffers + glUtils::generateTextures, SceneManager.cpp:483-565,
// glUtils.cpp:292-313): pinned staging, AoS -> SoA re-layout, mip chains and combo textures on the device, work buffers.
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

}  // namespace

namespace m2s_host {
m2s_status ensure_stage(m2s_ctx* c) {
    for (int k = 0; k < 2; ++k) {
        if (!c->h_stage[k]) HIPCHK(c, hipHostMalloc(&c->h_stage[k], kStageChunk, hipHostMallocDefault));
        if (!c->d_stage[k]) HIPCHK(c, hipMalloc(&c->d_stage[k], kStageChunk));
    }
    return M2S_OK;
}
}  // namespace m2s_host

namespace {
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
    ++c->records_epoch;
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
    // (n_tri / 32: room for AUTO's smaller batches of the 11-18 fragments-per-triangle band, BatchTable::tpw >= 32)
    const size_t chain_words = std::max<size_t>(std::max<size_t>(std::max<size_t>(n_fused_waves(n_tri), batch_table_capacity(n_tri)), (size_t)n_tri / 32 + 1), 1);
    const size_t o_meshes = take(n_mp * sizeof(MeshParams));
    const size_t o_mesh_first = take(mesh_first.size() * sizeof(uint32_t));
    const size_t o_mesh_of8 = take(((np + 7) / 8 + 1) * sizeof(uint2));
    const size_t o_cnt = take(np * sizeof(uint32_t));
    const size_t o_off = take((np + 1) * sizeof(uint32_t));
    const size_t o_partials = take(std::max<size_t>(n_count_blocks(n_tri), 1) * sizeof(uint32_t));
    const size_t o_chain = take(chain_words * sizeof(unsigned long long));
    const size_t o_biglist = take(np * sizeof(BigItem));
    const size_t o_bigmeta = take(4 * sizeof(uint32_t));
    // run tables (RunInfo): one per remembered density, one word per run of the kernel with the most units (k_fused2)
    const uint32_t units = fused2_band_workgroups(n_tri);
    const size_t run_words = run_shift_for(units) ? (size_t)n_runs(units, run_shift_for(units)) + 1 : 0;
    const size_t o_bands = take(std::max<size_t>((size_t)kBandSlots * run_words, 1) * sizeof(unsigned long long));
    const size_t o_run_order = take(std::max<size_t>(run_shift_for(units) ? run_order_slots(units, run_shift_for(units)) : 0, 1) * sizeof(uint32_t));
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

    const auto t_tex = std::chrono::steady_clock::now();
    // ---- mesh table, work buffers, and the upload's exact count — BEFORE the textures ------------------------------------------------
    // The count (warm_scene) needs the geometry and the meshes' bounding boxes, not the texels: enqueued here, it runs as soon as the
    // last geometry chunk has been repacked, and its result is on the host when the texture copies behind it have finished — the
    // upload's one synchronisation serves both (until round 6 the count started after that synchronisation and cost
    // the upload a second round trip, 0.15 ms of the "warm" share).  The texel pointers are addresses inside the arena: known already.
    for (TexPlan& p : tex_plan) p.d.texels = (uint32_t*)(A + p.arena_off);
    for (ComboPlan& cp : combo_plan) cp.d.texels = (uint32_t*)(A + cp.arena_off);
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
    // k_fused3's fragment stage has no sampler for separately sized maps: it may run if every mesh has a combo texture or no map
    c->lean_ok = true;
    for (uint32_t i = 0; i < n_meshes; ++i)
        if (mesh_combo[i] < 0 && (mesh_tex[i][0] >= 0 || mesh_tex[i][1] >= 0 || mesh_tex[i][2] >= 0)) c->lean_ok = false;
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
    c->run_table_words = run_words;
    c->d_run_order = (uint32_t*)(A + o_run_order);
    c->run_order_unit = 0; c->run_order_shift = 0;
    c->d_batch_first = (uint32_t*)(A + o_batch);
    c->n_batch_tab = 0;
    c->chain_words = chain_words;
    HIPCHK(c, hipMemsetAsync(c->d_chain, 0, chain_words * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), c->stream));
    const uint32_t warm_R = c->hint_R ? c->hint_R : c->last_R ? c->last_R : 1024u;
    const bool warm = !debug_on("M2S_NO_WARM");
    bool counted = false;
    if (warm && !debug_on("M2S_NO_WARM_OVERLAP")) counted = warm_count_enqueue(c, warm_R) == M2S_OK;

    // ---- textures: level 0 through the same staging, levels 1..4 on the device (glUtils.cpp:292-313) -----------------
    for (TexPlan& p : tex_plan) {
        uint32_t* mem = (uint32_t*)(A + p.arena_off);
        const m2s_status st = staged_h2d(c, (const char*)p.src, (size_t)p.d.w * p.d.h * 4, kStageChunk, (char*)mem, turn,
                                         [](char*, size_t, size_t) {});
        if (st != M2S_OK) return st;
        for (uint32_t l = 1; l < p.d.n_levels; ++l)
            launch_mip_level(mem + p.d.off[l - 1], std::max(1u, p.d.w >> (l - 1)), std::max(1u, p.d.h >> (l - 1)),
                             mem + p.d.off[l], std::max(1u, p.d.w >> l), std::max(1u, p.d.h >> l), c->stream);
    }
    for (ComboPlan& cp : combo_plan) {
        uint32_t* mem = (uint32_t*)(A + cp.arena_off);
        const TexDesc &ta = tex_plan[cp.ia].d, &tn = tex_plan[cp.in].d, &tm = tex_plan[cp.im].d;
        for (uint32_t l = 0; l < ta.n_levels; ++l)
            launch_combo_level(ta.texels + ta.off[l], tn.texels + tn.off[l], tm.texels + tm.off[l], std::max(1u, ta.w >> l),
                               std::max(1u, ta.h >> l), mem + cp.d.coff[l], c->stream);
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // mp / mesh_first are host temporaries; the caller's buffers are released
    HIPCHK(c, hipGetLastError());
    c->last_upload_ms[2] = ms_since(t_tex);
    c->has_scene = true;
    // what the first conversion would otherwise have to find out inside its own call (see warm_scene)
    const auto t_warm = std::chrono::steady_clock::now();
    if (warm) {
        const m2s_status ws = warm_scene(c, warm_R, counted);
        if (ws != M2S_OK) { c->has_scene = false; return ws; }
    }
    c->last_upload_ms[4] = ms_since(t_warm);
    c->last_upload_ms[0] = ms_since(t_begin);
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
    if ((flags & M2S_PREPARE_KERNELS) && !debug_on("M2S_NO_PRELOAD")) {
        HIPCHK(c, preload_fused2()); HIPCHK(c, preload_fused3()); HIPCHK(c, preload_sparse()); HIPCHK(c, preload_multipass());
        HIPCHK(c, preload_export()); HIPCHK(c, preload_prepass()); HIPCHK(c, preload_sort());
    }
    return M2S_OK;
}

m2s_status m2s_last_upload_ms(const m2s_ctx* c, float out_ms[4]) {
    if (!c || !out_ms) return M2S_ERR_INVALID;
    memcpy(out_ms, c->last_upload_ms, 4 * sizeof(float));
    return M2S_OK;
}

float m2s_last_warm_ms(const m2s_ctx* c) { return c ? c->last_upload_ms[4] : 0.0f; }

m2s_status m2s_set_resolution_hint(m2s_ctx* c, uint32_t R) {
    if (!c) return M2S_ERR_INVALID;
    if (R > 4096) return fail(c, M2S_ERR_INVALID, "R must be in [0, 4096]");
    c->hint_R = R;
    return M2S_OK;
}

}  // extern "C"
