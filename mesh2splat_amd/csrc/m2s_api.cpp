// m2s_api.cpp — host side of the C ABI (include/m2s.h): context, scene upload, the conversion
// pass driver (== ConversionPass::execute, src/renderer/renderPasses/ConversionPass.cpp:9-68) and
// read-back.  Compiled with hipcc; no CPU compute path exists here.
#include "../../include/m2s.h"
#include <cstdlib>
#include "m2s_device.h"
#include "m2s_ply.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

using namespace m2s;

namespace {
thread_local std::string g_create_error = "";

constexpr uint32_t kMaxGaussiansToSort = 7000000u;  // RenderPass.hpp:9
constexpr uint64_t kMaxTriangles = (1ull << 28) - 1;  // 32-bit byte offsets into the 16 B/triangle planes
constexpr size_t kStagingBytes = 256ull << 20;      // H2D staging for the AoS -> SoA repack

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
}  // namespace

struct m2s_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    void* tri_mem = nullptr;
    SceneDev scene{};
    MeshParams* d_meshes = nullptr;
    uint32_t* d_mesh_first = nullptr;
    std::vector<void*> tex_mem;
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
    uint32_t multipass_R = 0;               // AUTO: R at which this scene is converted by the multi-pass pipeline
    uint32_t decided_R = 0;                 // AUTO: R for which the fused / multi-pass decision has been taken
    uint32_t mp_ready_R = 0;                // R of the last completed multi-pass conversion (its work buffers are sized)
    int last_pipeline = 0;                  // what the last conversion ran (m2s_last_pipeline)
    // second lane for context-owned asynchronous submissions: odd slots run on their own stream with their own chain
    // and record buffer, so that consecutive single-kernel conversions overlap (the tail of one, where the GPU drains,
    // with the head of the next) instead of paying ~8 us between dependent kernels on one stream
    int lanes = 1;                          // m2s_set_async_lanes
    hipStream_t stream_b = nullptr;
    unsigned long long* d_chain_b = nullptr;
    void* d_records_b = nullptr;
    uint64_t records_b_cap = 0;
    BandInfo bands{};                       // XCD bands of k_fused2 for the scene at R == band_R (from the exact count)
    uint32_t band_R = 0;
    uint32_t team_off_R = 0;                // R at which k_fused2 reported a workgroup that did not fit its LDS stream
    int pipeline = M2S_PIPELINE_AUTO;
    uint32_t sized_R = 0;                   // unlimited-cap policy: R the context buffer was sized for
    uint32_t epoch = 0;                     // launch counter of the fused kernel (tags the chain words)

    // asynchronous submissions (m2s_convert_submit / m2s_convert_wait): a ring of result slots.  Slot k uses
    // h_total[2 + 2k] (counter) and h_total[3 + 2k] (status words), written by the kernel itself.
    struct Slot { hipEvent_t done = nullptr, t0 = nullptr, t1 = nullptr; uint64_t limit = 0; void* d_out = nullptr; uint32_t R = 0;
                  bool sync_result = false; uint64_t sync_total = 0; bool prof = false; float ms[M2S_K_N] = {}; };
    Slot slot[M2S_MAX_IN_FLIGHT];
    uint32_t slot_head = 0, slot_count = 0; // oldest in-flight slot, number in flight
    uint32_t async_ok_R = 0;                // R at which a completed conversion of this scene needed no host decision

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
    if (c->d_meshes) (void)hipFree(c->d_meshes);
    if (c->d_mesh_first) (void)hipFree(c->d_mesh_first);
    for (void* p : c->tex_mem) (void)hipFree(p);
    if (c->d_cnt) (void)hipFree(c->d_cnt);
    if (c->d_off) (void)hipFree(c->d_off);
    if (c->d_partials) (void)hipFree(c->d_partials);
    if (c->d_chain) (void)hipFree(c->d_chain);
    if (c->d_chain_b) (void)hipFree(c->d_chain_b);
    c->d_chain_b = nullptr;
    if (c->d_biglist) (void)hipFree(c->d_biglist);
    if (c->d_bigmeta) (void)hipFree(c->d_bigmeta);
    c->d_chain = nullptr; c->d_biglist = nullptr; c->d_bigmeta = nullptr;
    c->sized_R = 0;
    c->multipass_R = 0;
    c->decided_R = 0;
    c->mp_ready_R = 0;
    c->team_off_R = 0;
    c->band_R = 0;
    c->async_ok_R = 0;
    c->tri_mem = nullptr; c->d_meshes = nullptr; c->d_mesh_first = nullptr;
    c->tex_mem.clear();
    c->d_cnt = c->d_off = c->d_partials = nullptr;
    c->scene = SceneDev{};
    c->has_scene = false;
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
    for (auto& ev : c->ev)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    for (auto& sl : c->slot)
        if ((e = hipEventCreate(&sl.done)) != hipSuccess || (e = hipEventCreate(&sl.t0)) != hipSuccess ||
            (e = hipEventCreate(&sl.t1)) != hipSuccess)
            return bail("hipEventCreate", e);
    *out_ctx = c;
    return M2S_OK;
}

void m2s_destroy(m2s_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (uint32_t k = 0; k < c->slot_count; ++k) {   // conversions still in flight on a caller's stream
        auto& sl = c->slot[(c->slot_head + k) % M2S_MAX_IN_FLIGHT];
        if (!sl.sync_result) (void)hipEventSynchronize(sl.done);
    }
    free_scene(c);
    if (c->d_start) (void)hipFree(c->d_start);
    if (c->d_records) (void)hipFree(c->d_records);
    if (c->d_records_b) (void)hipFree(c->d_records_b);
    if (c->stream_b) { (void)hipStreamSynchronize(c->stream_b); (void)hipStreamDestroy(c->stream_b); }
    if (c->d_sorted) (void)hipFree(c->d_sorted);
    if (c->d_quads) (void)hipFree(c->d_quads);
    if (c->d_sorted_quads) (void)hipFree(c->d_sorted_quads);
    if (c->d_loaded) (void)hipFree(c->d_loaded);
    for (int k = 0; k < 2; ++k) if (c->h_export[k]) (void)hipHostFree(c->h_export[k]);
    if (c->d_pp_depths) (void)hipFree(c->d_pp_depths);
    if (c->d_pp_chain) (void)hipFree(c->d_pp_chain);
    if (c->d_pp_depthtex) (void)hipFree(c->d_pp_depthtex);
    if (c->d_sort_u32) (void)hipFree(c->d_sort_u32);
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

m2s_status m2s_upload_scene(m2s_ctx* c, const m2s_mesh* meshes, uint32_t n_meshes) {
    if (!c) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (n_meshes > 0xFFFFFFu) return fail(c, M2S_ERR_INVALID, "more than 2^24-1 meshes");   // TriShade keeps the index in 24 bits
    if (n_meshes && !meshes) return fail(c, M2S_ERR_INVALID, "meshes is NULL");
    HIPCHK(c, hipSetDevice(c->device));
    // ---- validate + global triangle index space -------------------------------------------------
    std::vector<uint32_t> mesh_first(n_meshes + 1, 0);
    uint64_t T = 0;
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const m2s_mesh& m = meshes[i];
        if (m.stride_floats < 12) return fail(c, M2S_ERR_INVALID, "stride_floats must be >= 12");
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

    // ---- geometry planes: 144 B / triangle ------------------------------------------------------
    const size_t np = std::max<size_t>(n_tri, 1);
    size_t offs[11], cur = 0;
    const size_t widths[11] = { 16, 16, 4, 16, 8, 16, 16, 4, 16, 16, 16 };
    for (int k = 0; k < 11; ++k) { offs[k] = cur; cur = align_up(cur + np * widths[k], 256); }
    HIPCHK(c, hipMalloc(&c->tri_mem, cur));
    char* b = (char*)c->tri_mem;
    TriPlanes& tp = c->scene.tri;
    tp.A0 = (const float4*)(b + offs[0]); tp.A1 = (const float4*)(b + offs[1]); tp.A2 = (const float*)(b + offs[2]);
    tp.B0 = (const float4*)(b + offs[3]); tp.B1 = (const float2*)(b + offs[4]);
    tp.C0 = (const float4*)(b + offs[5]); tp.C1 = (const float4*)(b + offs[6]); tp.C2 = (const float*)(b + offs[7]);
    tp.D0 = (const float4*)(b + offs[8]); tp.D1 = (const float4*)(b + offs[9]); tp.D2 = (const float4*)(b + offs[10]);

    if (n_tri) {
        void* staging = nullptr;
        size_t need = 0;
        for (uint32_t i = 0; i < n_meshes; ++i) {
            const uint64_t s = std::max<uint64_t>(first, mesh_first[i]), e = std::min<uint64_t>(last, mesh_first[i + 1]);
            if (e > s) need = std::max<size_t>(need, (size_t)(e - s) * 3 * meshes[i].stride_floats * sizeof(float));
        }
        const size_t stage_bytes = std::min(need, kStagingBytes);
        HIPCHK(c, hipMalloc(&staging, std::max<size_t>(stage_bytes, 256)));
        for (uint32_t i = 0; i < n_meshes; ++i) {
            const uint64_t s = std::max<uint64_t>(first, mesh_first[i]), e = std::min<uint64_t>(last, mesh_first[i + 1]);
            if (e <= s) continue;
            const size_t tri_bytes = (size_t)3 * meshes[i].stride_floats * sizeof(float);
            const uint64_t per_chunk = std::max<uint64_t>(1, stage_bytes / tri_bytes);
            for (uint64_t t0 = s; t0 < e; t0 += per_chunk) {
                const uint64_t n = std::min<uint64_t>(per_chunk, e - t0);
                const float* src = meshes[i].vertices + (size_t)(t0 - mesh_first[i]) * 3 * meshes[i].stride_floats;
                hipError_t he = hipMemcpyAsync(staging, src, n * tri_bytes, hipMemcpyHostToDevice, c->stream);
                if (he == hipSuccess) {
                    launch_repack((const float*)staging, meshes[i].stride_floats, (uint32_t)n, 0, (uint32_t)n,
                                  (uint32_t)(t0 - first), tp, c->stream);
                    he = hipStreamSynchronize(c->stream);  // staging is reused by the next chunk
                }
                if (he != hipSuccess) { (void)hipFree(staging); HIPCHK(c, he); }
            }
        }
        (void)hipFree(staging);
    }

    // ---- textures: level 0 upload + mip levels 1..4 (glUtils.cpp:292-313) ------------------------
    std::vector<MeshParams> mp(std::max<uint32_t>(n_meshes, 1));
    std::map<std::tuple<const uint8_t*, uint32_t, uint32_t>, TexDesc> dedup;
    for (uint32_t i = 0; i < n_meshes; ++i) {
        const m2s_mesh& m = meshes[i];
        MeshParams& p = mp[i];
        memset(&p, 0, sizeof p);
        memcpy(p.bmin, m.bbox_min, 12);
        memcpy(p.bmax, m.bbox_max, 12);
        memcpy(p.color, m.base_color, 16);
        for (int k = 0; k < 3; ++k) {
            const m2s_texture& t = m.tex[k];
            if (!t.rgba8) continue;
            auto key = std::make_tuple(t.rgba8, t.width, t.height);
            auto it = dedup.find(key);
            if (it != dedup.end()) { p.tex[k] = it->second; continue; }
            TexDesc d{};
            d.w = t.width; d.h = t.height;
            uint32_t mx = std::max(t.width, t.height), nl = 1;
            while (mx > 1 && nl < 5) { mx >>= 1; nl++; }
            d.n_levels = nl;
            size_t tot = 0;
            for (uint32_t l = 0; l < nl; ++l) {
                d.off[l] = (uint32_t)tot;
                tot += (size_t)std::max(1u, t.width >> l) * std::max(1u, t.height >> l);
            }
            void* mem = nullptr;
            HIPCHK(c, hipMalloc(&mem, tot * 4));
            c->tex_mem.push_back(mem);
            d.texels = (const uint32_t*)mem;
            HIPCHK(c, hipMemcpyAsync(mem, t.rgba8, (size_t)t.width * t.height * 4, hipMemcpyHostToDevice, c->stream));
            for (uint32_t l = 1; l < nl; ++l)
                launch_mip_level(d.texels + d.off[l - 1], std::max(1u, t.width >> (l - 1)), std::max(1u, t.height >> (l - 1)),
                                 (uint32_t*)mem + d.off[l], std::max(1u, t.width >> l), std::max(1u, t.height >> l), c->stream);
            p.tex[k] = d;
            dedup[key] = d;
        }
    }
    // ---- combo textures (interleaved albedo/normal/MR, see ComboDesc) ----------------------------------
    {
        std::map<std::tuple<const uint32_t*, const uint32_t*, const uint32_t*>, ComboDesc> cdedup;
        for (uint32_t i = 0; i < n_meshes; ++i) {
            MeshParams& p = mp[i];
            const TexDesc &ta = p.tex[0], &tn = p.tex[1], &tm = p.tex[2];
            if (!ta.texels || !tn.texels || !tm.texels) continue;
            if (ta.w != tn.w || ta.w != tm.w || ta.h != tn.h || ta.h != tm.h) continue;
            auto key = std::make_tuple(ta.texels, tn.texels, tm.texels);
            auto it = cdedup.find(key);
            if (it != cdedup.end()) { p.combo = it->second; continue; }
            ComboDesc cd{};
            size_t tot = 0;
            for (uint32_t l = 0; l < ta.n_levels; ++l) {
                cd.coff[l] = (uint32_t)tot;
                tot += (size_t)(std::max(1u, ta.w >> l) + 1) * std::max(1u, ta.h >> l) * 3;
            }
            if (tot > 0x3FFFFFF0ull) continue;  // the sampler addresses the combo texels with 32-bit BYTE offsets
            void* mem = nullptr;
            HIPCHK(c, hipMalloc(&mem, tot * 4));
            c->tex_mem.push_back(mem);
            cd.texels = (const uint32_t*)mem;
            for (uint32_t l = 0; l < ta.n_levels; ++l)
                launch_combo_level(ta.texels + ta.off[l], tn.texels + tn.off[l], tm.texels + tm.off[l], std::max(1u, ta.w >> l),
                                   std::max(1u, ta.h >> l), (uint32_t*)mem + cd.coff[l], c->stream);
            p.combo = cd;
            cdedup[key] = cd;
        }
    }
    HIPCHK(c, hipMalloc((void**)&c->d_meshes, mp.size() * sizeof(MeshParams)));
    HIPCHK(c, hipMemcpyAsync(c->d_meshes, mp.data(), mp.size() * sizeof(MeshParams), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMalloc((void**)&c->d_mesh_first, mesh_first.size() * sizeof(uint32_t)));
    HIPCHK(c, hipMemcpyAsync(c->d_mesh_first, mesh_first.data(), mesh_first.size() * sizeof(uint32_t), hipMemcpyHostToDevice,
                             c->stream));
    c->scene.meshes = c->d_meshes;
    c->scene.mesh_first = c->d_mesh_first;

    // ---- work buffers -----------------------------------------------------------------------------
    HIPCHK(c, hipMalloc((void**)&c->d_cnt, np * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_off, (np + 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_partials, std::max<size_t>(n_count_blocks(n_tri), 1) * sizeof(uint32_t)));
    HIPCHK(c, hipMalloc((void**)&c->d_chain, std::max<size_t>(n_fused_waves(n_tri), 1) * sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(c->d_chain, 0, std::max<size_t>(n_fused_waves(n_tri), 1) * sizeof(unsigned long long), c->stream));
    HIPCHK(c, hipMalloc((void**)&c->d_biglist, np * sizeof(BigItem)));
    HIPCHK(c, hipMalloc((void**)&c->d_bigmeta, 4 * sizeof(uint32_t)));
    HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // mp / mesh_first are host temporaries
    c->has_scene = true;
    c->last_total = c->last_stored = 0;
    c->last_records = nullptr;
    return M2S_OK;
}

// Which form of the single-pass kernel (see m2s_fused2.hip)?  The workgroup-cooperative one unless the scene is too
// small to fill the GPU with 64-triangle batches (k_fused then runs 32 / 16 triangles per wave) or a workgroup's
// fragments did not fit its LDS stream at this R before.
// Chain words carry a 16-bit launch tag instead of being cleared per launch.  The two single-pass kernels use different
// numbers of words, so a word one of them left behind could read as freshly published 65 536 launches later: when the
// tag wraps, everything in flight is drained and both chains are cleared (once per ~10 s of back-to-back conversions).
static hipError_t next_epoch(m2s_ctx* c, uint32_t* out) {
    const uint32_t e = ++c->epoch;
    *out = e;
    if ((e & 0xFFFFu) != 0) return hipSuccess;
    const size_t bytes = std::max<size_t>(n_fused_waves(c->scene.n_tri), 1) * sizeof(unsigned long long);
    hipError_t r = hipDeviceSynchronize();
    if (r == hipSuccess && c->d_chain) r = hipMemset(c->d_chain, 0, bytes);
    if (r == hipSuccess && c->d_chain_b) r = hipMemset(c->d_chain_b, 0, bytes);
    return r;
}

static bool use_team(const m2s_ctx* c, uint32_t R) {
    if (c->pipeline == M2S_PIPELINE_WAVE || c->team_off_R == R) return false;
    return true;
}

static BandInfo bands_for(const m2s_ctx* c, uint32_t R) {
    if (c->band_R == R) return c->bands;
    BandInfo none{};
    return none;
}

static uint64_t resolve_cap(const m2s_ctx* c, uint32_t R) {
    if (c->cap_policy == 0) return 0;
    if (c->cap_policy > 0) return (uint64_t)c->cap_policy;
    // ConversionPass.cpp:21-24 (unsigned int arithmetic wraps)
    const uint32_t mc = std::max<uint32_t>(1u, c->n_meshes_total);
    const uint32_t mx = R * R * 6u * mc;
    return std::min(mx, kMaxGaussiansToSort);
}

// Multi-pass pipeline (count -> scan -> offsets -> emit): handles every triangle size, output-balanced.
static m2s_status run_multipass(m2s_ctx* c, uint32_t R, float4* d_out, uint64_t limit, bool counted, hipStream_t st) {
    const SceneDev& sc = c->scene;
    const bool prof = c->profiling;
    if (!counted) {
        if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
        launch_count(sc, R, c->d_cnt, c->d_partials, st);
        if (prof) HIPCHK(c, hipEventRecord(c->ev[1], st));
        launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
        if (prof) HIPCHK(c, hipEventRecord(c->ev[2], st));
    }
    const uint32_t n_blocks = (uint32_t)((limit + kEmitF - 1) / kEmitF);
    if (c->start_cap < n_blocks) {
        if (c->d_start) { (void)hipFree(c->d_start); c->d_start = nullptr; c->start_cap = 0; }
        HIPCHK(c, hipMalloc((void**)&c->d_start, std::max<size_t>(n_blocks, 1) * sizeof(uint32_t)));
        c->start_cap = n_blocks;
    }
    launch_offsets(c->d_cnt, c->d_partials, sc.n_tri, c->d_off, c->d_start, n_blocks, st);
    if (prof) HIPCHK(c, hipEventRecord(c->ev[3], st));
    launch_emit(sc, R, c->d_off, c->d_start, c->d_total, limit, d_out, n_blocks, st);
    if (prof) HIPCHK(c, hipEventRecord(c->ev[4], st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
    if (prof)
        for (int k = counted ? 2 : 0; k < 4; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_ms[k], c->ev[k], c->ev[k + 1]));
    return M2S_OK;
}

static m2s_status run_pass(m2s_ctx* c, uint32_t R, void* d_user, uint64_t user_cap, hipStream_t st, uint64_t* out_total,
                           bool from_submit = false) {
    if (!c->has_scene) return fail(c, M2S_ERR_STATE, "m2s_upload_scene has not been called");
    if (c->slot_count && !from_submit)
        return fail(c, M2S_ERR_STATE, "conversions submitted with m2s_convert_submit are still in flight: m2s_convert_wait first");
    if (R == 0 || R > 4096) return fail(c, M2S_ERR_INVALID, "R must be in [1, 4096]");
    HIPCHK(c, hipSetDevice(c->device));
    const SceneDev& sc = c->scene;
    const uint64_t cap = resolve_cap(c, R);
    const bool prof = c->profiling;
    c->last_R = R;
    memset(c->last_ms, 0, sizeof c->last_ms);

    if (sc.n_tri == 0) {
        c->last_total = c->last_stored = 0;
        c->last_records = d_user ? d_user : c->d_records;
        if (out_total) *out_total = 0;
        return M2S_OK;
    }

    // ---- where do the records go, and how many may be stored? ------------------------------------
    bool counted = false;  // k_count + k_scan already ran in this call
    uint64_t limit;
    float4* d_out;
    if (d_user) {
        limit = cap ? std::min(cap, user_cap) : user_cap;
        d_out = (float4*)d_user;
    } else {
        uint64_t want = cap;
        if (!cap && (c->sized_R != R || !c->d_records)) {
            // unlimited policy: size the SSBO from an exact count (once per (scene, R))
            if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
            launch_count(sc, R, c->d_cnt, c->d_partials, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[1], st));
            launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[2], st));
            HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (prof)
                for (int k = 0; k < 2; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_ms[k], c->ev[k], c->ev[k + 1]));
            counted = true;
            want = std::max<uint64_t>(c->h_total[0], 1);
            c->sized_R = R;
        } else if (!cap) {
            want = c->records_cap;
        }
        // ConversionPass.cpp:25-33: (re)allocate when the size differs (grow-only for the unlimited policy)
        if ((cap && c->records_cap != want) || (!cap && c->records_cap < want)) {
            if (c->d_records) { (void)hipFree(c->d_records); c->d_records = nullptr; c->records_cap = 0; }
            HIPCHK(c, hipMalloc(&c->d_records, want * sizeof(m2s_gaussian)));
            c->records_cap = want;
        }
        limit = cap ? cap : c->records_cap;
        d_out = (float4*)c->d_records;
    }
    if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;

    // ---- AUTO: which pipeline for this scene at this R? ---------------------------------------------
    // The single-pass kernel wins while triangles are small (it does the per-triangle work once and needs no second
    // sweep); with more than ~11 fragments per triangle on average the output-partitioned multi-pass pipeline is
    // faster and soon much faster (2.74 M fragments at R = 1024 from 1 M / 250 k / 125 k / 62 k triangles: fused 0.167 /
    // 0.138 / 0.323 / 0.626 ms, multi-pass 0.214 / 0.137 / 0.136 / 0.154 ms; tools/auto_probe.py).  The exact count
    // costs 0.02-0.06 ms and is taken once per (scene, R); the decision is remembered.
    if (c->pipeline == M2S_PIPELINE_AUTO && c->decided_R != R) {
        if (!counted) {
            if (prof) HIPCHK(c, hipEventRecord(c->ev[0], st));
            launch_count(sc, R, c->d_cnt, c->d_partials, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[1], st));
            launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[2], st));
            HIPCHK(c, hipMemcpyAsync(c->h_total, c->d_total, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipStreamSynchronize(st));
            if (prof)
                for (int k = 0; k < 2; ++k) HIPCHK(c, hipEventElapsedTime(&c->last_ms[k], c->ev[k], c->ev[k + 1]));
            counted = true;
        }
        c->decided_R = R;
        c->multipass_R = (c->h_total[0] >= 11ull * sc.n_tri) ? R : 0;
        // XCD bands for k_fused2: the same count tells where the output of each eighth of the triangle list starts
        // (d_partials now holds the exclusive prefix per 1024 triangles = per 4 workgroups of 256)
        c->band_R = 0;
        if (!c->multipass_R && fused_tpw(sc.n_tri) == 64u && !std::getenv("M2S_NO_BANDS")) {
            const uint32_t wgs = (n_fused_waves(sc.n_tri) + 3u) / 4u;
            uint32_t bpb = (wgs + 7u) / 8u;
            bpb = (bpb + 3u) & ~3u;
            const uint32_t n_part = n_count_blocks(sc.n_tri);
            uint32_t pre[8];
            for (int x = 0; x < 8; ++x) {
                const uint32_t pi = (uint32_t)x * (bpb / 4u);
                if (pi < n_part) HIPCHK(c, hipMemcpyAsync(&pre[x], c->d_partials + pi, 4, hipMemcpyDeviceToHost, st));
            }
            HIPCHK(c, hipStreamSynchronize(st));
            for (int x = 0; x < 8; ++x) {
                const uint32_t pi = (uint32_t)x * (bpb / 4u);
                c->bands.base[x] = pi < n_part ? (unsigned long long)pre[x] : c->h_total[0];
            }
            c->bands.workgroups_per_band = bpb;
            c->band_R = R;
        }
    }

    // ---- run ---------------------------------------------------------------------------------------
    bool done = false;
    if (c->pipeline != M2S_PIPELINE_MULTIPASS && c->multipass_R != R) {
        counted = false;   // the fused kernel does its own counting; a count taken above only sized / decided
        // single-pass kernel; triangles too large for its in-workgroup budget are only counted.
        // No memset, no memcpy: the look-back chain is epoch-tagged and the kernel writes the fragment
        // counter and its two status words straight into pinned host memory.
        uint32_t any_big = 0, err = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool team = use_team(c, R);
            c->h_total[0] = 0;
            c->h_total[1] = 0;
            uint32_t epoch;
            HIPCHK(c, next_epoch(c, &epoch));
            if (prof) HIPCHK(c, hipEventRecord(c->ev[5], st));
            if (team) launch_fused2(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                                    c->d_biglist, c->d_bigmeta, bands_for(c, R), st);
            else launch_fused(sc, R, c->d_chain, limit, d_out, &c->h_total[0], reinterpret_cast<uint32_t*>(&c->h_total[1]), epoch,
                              c->d_biglist, c->d_bigmeta, st);
            if (prof) HIPCHK(c, hipEventRecord(c->ev[6], st));
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipStreamSynchronize(st));  // glFinish + counter read-back (ConversionPass.cpp:54-59)
            if (prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_FUSED], c->ev[5], c->ev[6]));
            any_big = (uint32_t)(c->h_total[1] & 0xFFFFFFFFull);
            err = (uint32_t)(c->h_total[1] >> 32);
            c->last_pipeline = team ? M2S_PIPELINE_TEAM : M2S_PIPELINE_WAVE;
            if (!(err && team)) break;
            // a workgroup's fragments did not fit the team kernel's LDS stream (or a wait timed out): the one-wave-per-batch
            // form has no such limit.  Remember it for this scene and R, forget what the aborted launch listed, try again.
            c->team_off_R = R;
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
        }
        done = true;
        // a clean single-kernel conversion: the same scene at the same R can be submitted asynchronously from now on
        c->async_ok_R = (!err && !any_big) ? R : 0;
        if (err) {
            // The bounded look-back spin gave up (never observed; would need a dispatcher that starves earlier
            // workgroups).  Degrade to the multi-pass pipeline, which has no inter-workgroup dependency.
            HIPCHK(c, hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), st));
            c->multipass_R = R;
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
                c->multipass_R = R;
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
        c->mp_ready_R = R;
        c->last_pipeline = M2S_PIPELINE_MULTIPASS;
    }
    const uint64_t total = c->h_total[0];
    if (total > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 fragments: offsets are 32-bit");
    c->last_total = total;
    c->last_stored = std::min(total, limit);
    c->last_records = d_out;
    if (out_total) *out_total = total;
    return M2S_OK;
}

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
    // Fast path: this scene at this R already converted cleanly with the single kernel (no deferred triangles, so no
    // host decision between kernels) and the output buffer needs no (re)allocation.
    const bool own_ready = d_records || (cap ? (c->d_records && c->records_cap == cap) : (c->d_records && c->sized_R == R));
    const bool fast = c->scene.n_tri > 0 && c->async_ok_R == R && c->pipeline != M2S_PIPELINE_MULTIPASS && c->multipass_R != R && own_ready;
    // Multi-pass conversions have no host decision between their four kernels either; once this (scene, R) has been
    // converted that way (work buffers sized, AUTO decision taken) they are enqueued without waiting as well.
    // (With kernel timing on they run synchronously: the per-kernel events are shared.)
    const bool fast_mp = !fast && c->scene.n_tri > 0 && own_ready && !c->profiling && c->mp_ready_R == R &&
                         (c->pipeline == M2S_PIPELINE_MULTIPASS || (c->decided_R == R && c->multipass_R == R));
    sl.R = R;
    if (fast_mp) {
        HIPCHK(c, hipSetDevice(c->device));
        uint64_t limit;
        void* d_out;
        if (d_records) { limit = cap ? std::min(cap, capacity_records) : capacity_records; d_out = d_records; }
        else { limit = cap ? cap : c->records_cap; d_out = c->d_records; }
        if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;
        const uint32_t n_blocks = (uint32_t)((limit + kEmitF - 1) / kEmitF);
        if (c->start_cap >= n_blocks) {
            unsigned long long* res = &c->h_total[2 + 2 * k];
            res[0] = 0; res[1] = 0;
            const SceneDev& sc = c->scene;
            launch_count(sc, R, c->d_cnt, c->d_partials, st);
            launch_scan_partials(c->d_partials, n_count_blocks(sc.n_tri), c->d_total, st);
            launch_offsets(c->d_cnt, c->d_partials, sc.n_tri, c->d_off, c->d_start, n_blocks, st);
            launch_emit(sc, R, c->d_off, c->d_start, c->d_total, limit, (float4*)d_out, n_blocks, st);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(&res[0], c->d_total, 8, hipMemcpyDeviceToHost, st));
            HIPCHK(c, hipEventRecord(sl.done, st));
            c->last_pipeline = M2S_PIPELINE_MULTIPASS;
            sl.prof = false;
            sl.sync_result = false;
            sl.limit = limit;
            sl.d_out = d_out;
            ++c->slot_count;
            return M2S_OK;
        }
    }
    if (!fast) {
        // first conversion of a (scene, R), or one that needs the second stage / the multi-pass pipeline: run it now
        uint64_t total = 0;
        const m2s_status s = run_pass(c, R, d_records, capacity_records, st, &total, true);
        if (s != M2S_OK) return s;
        sl.sync_result = true;
        sl.sync_total = total;
        memcpy(sl.ms, c->last_ms, sizeof sl.ms);   // a later submit overwrites last_ms before this slot is waited for
        sl.limit = c->last_stored;   // already clamped
        sl.d_out = const_cast<void*>(c->last_records);
        ++c->slot_count;
        return M2S_OK;
    }
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t limit;
    void* d_out;
    unsigned long long* chain = c->d_chain;
    if (d_records) { limit = cap ? std::min(cap, capacity_records) : capacity_records; d_out = d_records; }
    else {
        limit = cap ? cap : c->records_cap;
        d_out = c->d_records;
        if ((k & 1u) && c->lanes == 2) {
            // odd slots: the second lane (allocated on first use).  Records of consecutive conversions then alternate
            // between two context-owned buffers; m2s_device_records / m2s_download follow the conversion last waited for.
            if (!c->stream_b) HIPCHK(c, hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
            if (!c->d_chain_b) {
                const size_t words = std::max<size_t>(n_fused_waves(c->scene.n_tri), 1);
                HIPCHK(c, hipMalloc((void**)&c->d_chain_b, words * sizeof(unsigned long long)));
                HIPCHK(c, hipMemsetAsync(c->d_chain_b, 0, words * sizeof(unsigned long long), c->stream_b));
            }
            if (c->records_b_cap != c->records_cap) {
                if (c->d_records_b) { (void)hipFree(c->d_records_b); c->d_records_b = nullptr; c->records_b_cap = 0; }
                HIPCHK(c, hipMalloc(&c->d_records_b, c->records_cap * sizeof(m2s_gaussian)));
                c->records_b_cap = c->records_cap;
            }
            st = c->stream_b;
            chain = c->d_chain_b;
            d_out = c->d_records_b;
        }
    }
    if (limit > 0xFFFFFFFFull) limit = 0xFFFFFFFFull;
    unsigned long long* res = &c->h_total[2 + 2 * k];
    res[0] = 0; res[1] = 0;
    sl.prof = c->profiling;
    uint32_t epoch;
    HIPCHK(c, next_epoch(c, &epoch));
    if (sl.prof) HIPCHK(c, hipEventRecord(sl.t0, st));
    c->last_pipeline = use_team(c, R) ? M2S_PIPELINE_TEAM : M2S_PIPELINE_WAVE;
    if (use_team(c, R)) launch_fused2(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                                      c->d_biglist, c->d_bigmeta, bands_for(c, R), st);
    else launch_fused(c->scene, R, chain, limit, (float4*)d_out, &res[0], reinterpret_cast<uint32_t*>(&res[1]), epoch,
                      c->d_biglist, c->d_bigmeta, st);
    if (sl.prof) HIPCHK(c, hipEventRecord(sl.t1, st));
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipEventRecord(sl.done, st));
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
    if (sl.sync_result) {   // run_pass already filled last_*
        if (out_total) *out_total = sl.sync_total;
        c->last_total = sl.sync_total; c->last_stored = sl.limit; c->last_records = sl.d_out; c->last_R = sl.R;
        memcpy(c->last_ms, sl.ms, sizeof sl.ms);
        return M2S_OK;
    }
    HIPCHK(c, hipEventSynchronize(sl.done));
    const uint64_t total = c->h_total[2 + 2 * k];
    const uint32_t any_big = (uint32_t)(c->h_total[3 + 2 * k] & 0xFFFFFFFFull), err = (uint32_t)(c->h_total[3 + 2 * k] >> 32);
    memset(c->last_ms, 0, sizeof c->last_ms);
    if (sl.prof) HIPCHK(c, hipEventElapsedTime(&c->last_ms[M2S_K_FUSED], sl.t0, sl.t1));
    if (err || any_big) {   // cannot happen for a scene/R that converted cleanly before; never return partial output silently
        c->async_ok_R = 0;
        (void)hipMemsetAsync(c->d_bigmeta, 0, 4 * sizeof(uint32_t), c->stream);
        (void)hipStreamSynchronize(c->stream);
        return fail(c, M2S_ERR_STATE, "asynchronous conversion needed a host decision; convert synchronously");
    }
    if (total > 0xFFFFFFFFull) return fail(c, M2S_ERR_CAPACITY, "more than 2^32-1 fragments: offsets are 32-bit");
    c->last_total = total;
    c->last_stored = std::min(total, sl.limit);
    c->last_records = sl.d_out;
    c->last_R = sl.R;
    if (out_total) *out_total = total;
    return M2S_OK;
}

uint64_t m2s_num_stored(const m2s_ctx* c) { return c ? c->last_stored : 0; }
const void* m2s_device_records(const m2s_ctx* c) { return c ? c->last_records : nullptr; }
uint64_t m2s_num_triangles(const m2s_ctx* c) { return c ? c->scene.n_tri : 0; }

m2s_status m2s_download(m2s_ctx* c, m2s_gaussian* dst, uint64_t capacity_records) {
    if (!c) return M2S_ERR_INVALID;
    if (!c->last_stored) return M2S_OK;
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

m2s_status m2s_export_ply(m2s_ctx* c, const char* path, uint32_t format, float gaussian_std) {
    if (!c || !path) return M2S_ERR_INVALID;
    if (!c->last_R) return fail(c, M2S_ERR_STATE, "no conversion has run");
    HIPCHK(c, hipSetDevice(c->device));
    // SceneManager.cpp:668
    const float scale_multiplier = gaussian_std / static_cast<float>(c->last_R);
    // The records come down in chunks through two pinned buffers: while chunk k is encoded and written, chunk k+1 is on the
    // PCIe bus (the reference maps the whole SSBO, then issues 62 stream writes per Gaussian from one thread).
    const uint64_t n = c->last_stored;
    const size_t chunk = m2s_ply::kChunkRows;
    for (int k = 0; k < 2; ++k)
        if (!c->h_export[k]) HIPCHK(c, hipHostMalloc((void**)&c->h_export[k], chunk * sizeof(m2s_gaussian), hipHostMallocDefault));
    m2s_ply::Writer w;
    m2s_status s = w.open(path, n, format, scale_multiplier);
    if (s != M2S_OK) { c->err = std::string("could not write ") + path; return s; }
    const char* src = static_cast<const char*>(c->last_records);
    auto rows_of = [&](uint64_t k) { return (size_t)std::min<uint64_t>(chunk, n - k * chunk); };
    const uint64_t n_chunks = (n + chunk - 1) / chunk;
    if (n_chunks) HIPCHK(c, hipMemcpyAsync(c->h_export[0], src, rows_of(0) * sizeof(m2s_gaussian), hipMemcpyDeviceToHost, c->stream));
    for (uint64_t k = 0; k < n_chunks && s == M2S_OK; ++k) {
        HIPCHK(c, hipStreamSynchronize(c->stream));                      // chunk k has arrived
        if (k + 1 < n_chunks)
            HIPCHK(c, hipMemcpyAsync(c->h_export[(k + 1) & 1], src + (k + 1) * chunk * sizeof(m2s_gaussian), rows_of(k + 1) * sizeof(m2s_gaussian),
                                     hipMemcpyDeviceToHost, c->stream));
        s = w.append(c->h_export[k & 1], rows_of(k));                    // returns once the pinned buffer has been read
    }
    const m2s_status cs = w.close();
    if (s == M2S_OK) s = cs;
    if (s != M2S_OK) c->err = std::string("could not write ") + path;
    return s;
}

// RadixSortPass::execute (RadixSortPass.cpp:8-90) on the records of the last conversion.
m2s_status m2s_sort_by_depth(m2s_ctx* c, const float world_to_view[16], uint64_t* out_n) {
    if (!c || !world_to_view) return M2S_ERR_INVALID;
    if (!c->last_records) return fail(c, M2S_ERR_STATE, "no conversion has run and no records were uploaded");
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
    c->sorted_n = 0;
    c->pp_visible = 0;
    c->sq_n = 0;
    return M2S_OK;
}

// GaussiansPrepass::execute (GaussiansPrepass.cpp:8-56) + the counter read-back that follows it (RadixSortPass.cpp:18-22).
m2s_status m2s_prepass(m2s_ctx* c, const m2s_prepass_params* p, const void* d_records, uint64_t n, uint64_t* out_visible) {
    if (!c || !p) return M2S_ERR_INVALID;
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (!d_records) {
        if (!c->last_records) return fail(c, M2S_ERR_STATE, "no conversion has run, no records were uploaded and none were passed");
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
    if (pipeline < M2S_PIPELINE_AUTO || pipeline > M2S_PIPELINE_TEAM) return fail(c, M2S_ERR_INVALID, "unknown pipeline");
    if (c->slot_count) return fail(c, M2S_ERR_STATE, "conversions are still in flight: m2s_convert_wait first");
    if (c->pipeline != pipeline) {   // what was remembered about this scene under the old setting no longer applies
        c->decided_R = 0; c->multipass_R = 0; c->team_off_R = 0; c->async_ok_R = 0; c->mp_ready_R = 0; c->band_R = 0;
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
