// m2s_device.h — device-side data layout and launcher prototypes (internal; the public
// boundary is include/m2s.h).  gfx950 only.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace m2s {

// Debug / A-B switches of the library (M2S_NO_BANDS, M2S_NO_LEAN, M2S_NO_WARM, ...) exist only in a DEBUG BUILD (`make EXTRA=-DM2S_DEBUG_BUILD
// OUT=../_build_debug`: what tools/first_call_probe.py, tools/mp_probe.py and the A/B scripts under tools/ab/ load through M2S_LIB_PATH), and
// even there they are only looked at when M2S_DEBUG is set in the environment.  In the library that ships every debug_on() is the constant
// false: the conversion path reads no environment and carries no switch (round 6).
#ifdef M2S_DEBUG_BUILD
inline const char* debug_env(const char* name) {
    static const bool enabled = std::getenv("M2S_DEBUG") != nullptr;
    return enabled ? std::getenv(name) : nullptr;
}
#else
inline const char* debug_env(const char*) { return nullptr; }
#endif
inline bool debug_on(const char* name) { const char* v = debug_env(name); return v && *v && *v != '0'; }

// ---- launch geometry ----------------------------------------------------------------------
constexpr int kBlock = 256;              // 4 wave64 per workgroup
constexpr int kTriPerBlock = 1024;       // triangles per workgroup in the upload's count kernel
constexpr int kEmitF = 1024;             // output records per workgroup of k_emit_big
constexpr int kRowsThread = 32;          // k_emit2: triangles with more pixel rows are expanded wave-cooperatively
constexpr int kRowsCount = 128;          // k_count: triangles with more pixel rows are counted wave-cooperatively
constexpr int kStageStride = 7;          // float4 per staged record in LDS (6 + 1 pad: conflict-free b128)

// ---- HBM layout of the geometry: 144 B / triangle in 11 coalescable planes ------------------
// (reference VBO: 17 floats/vertex AoS with 5 dead floats, SceneManager.cpp:483-512)
struct TriPlanes {
    const float4* A0;  // p0.x p0.y p0.z p1.x
    const float4* A1;  // p1.y p1.z p2.x p2.y
    const float*  A2;  // p2.z
    const float4* B0;  // uv0.x uv0.y uv1.x uv1.y
    const float2* B1;  // uv2.x uv2.y
    const float4* C0;  // n0.x n0.y n0.z n1.x
    const float4* C1;  // n1.y n1.z n2.x n2.y
    const float*  C2;  // n2.z
    const float4* D0;  // tangent 0 (xyz w)
    const float4* D1;  // tangent 1
    const float4* D2;  // tangent 2
};

struct TexDesc {
    const uint32_t* texels;  // RGBA8 mip chain, level 0 first; nullptr = map absent
    uint32_t w, h, n_levels;
    uint32_t off[5];         // level offsets in texels
};

// per-mesh "uniforms" (ConversionPass.cpp:77-112)
// "Combo" texture: when the three maps of a material exist and have identical dimensions their texels
// are interleaved — texel (i,j) of level l is the 12-byte triple {albedo, normal, metallic-roughness}
// at combo + coff[l] + (j*(W_l+1) + i)*3 dwords, and every row carries one extra wrapped texel
// (column W_l == column 0).  A bilinear footprint of ALL THREE maps is then two contiguous 24-byte
// row reads instead of twelve scattered 4-byte gathers.
struct ComboDesc {
    const uint32_t* texels;  // nullptr = not available (fallback: the separate maps)
    uint32_t coff[5];        // level offsets in dwords
};

struct MeshParams {
    float bmin[3];
    float bmax[3];
    float color[4];
    TexDesc tex[3];
    ComboDesc combo;
};

// One triangle deferred by the fused kernel (too large for its in-workgroup budget): emitted by k_emit_big.
struct BigItem {
    uint32_t t, cnt;           // local triangle index, fragment count
    unsigned long long off;    // index of its first record in the canonically ordered output
};

struct SceneDev {
    TriPlanes tri;
    const MeshParams* meshes;
    const uint32_t* mesh_first;  // n_meshes+1 prefix of GLOBAL triangle indices
    const uint2* mesh_of8;       // per 8 local triangles: {mesh of local triangle 8k, local index one past that mesh's last
                                 // triangle}: ONE scalar load tells a wave / workgroup its mesh and whether its range lies
                                 // inside it (round 3; before: a binary search over mesh_first = log2(meshes) + 1 dependent
                                 // scalar loads in front of every batch's position loads)
    uint32_t n_meshes;
    uint32_t n_tri;              // triangles resident on this device (the shard)
    uint32_t tri_first;          // global index of local triangle 0
};

// ---- launchers (all asynchronous on `st`) ----------------------------------------------------
void launch_repack(const float* d_aos, uint32_t stride_floats, uint32_t n_tri_src, uint32_t src_first,
                   uint32_t n, uint32_t dst_first, TriPlanes dst /*non-const view*/, hipStream_t st);
void launch_mesh_table(const SceneDev& sc, uint2* table, hipStream_t st);   // fills mesh_of8 (ceil(n_tri / 8) entries)
void launch_mip_level(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh,
                      hipStream_t st);
void launch_combo_level(const uint32_t* a, const uint32_t* n, const uint32_t* m, uint32_t w, uint32_t h, uint32_t* dst,
                        hipStream_t st);
void launch_count(const SceneDev& sc, uint32_t R, uint32_t* cnt, uint32_t* partials, hipStream_t st);
void launch_scan_partials(uint32_t* partials, uint32_t n_partials, unsigned long long* total, hipStream_t st);
// *out (zero before) += fragments of the triangles with more than `threshold` fragments (cnt: launch_count's counts)
void launch_big_share(const uint32_t* cnt, uint32_t n_tri, uint32_t threshold, unsigned long long* out, hipStream_t st);
// where the output of every RUN (1 << shift units of `unit` = 256 / 512 triangles) starts, from launch_count's counts and the
// scanned partial sums: the table a launch in runs reads (RunInfo::base)
void launch_unit_bases(const uint32_t* cnt, const uint32_t* partials, uint32_t n_tri, uint32_t unit, uint32_t shift, unsigned long long* run_base, hipStream_t st);
// Dispatch order of the runs of a launch in runs: order[slot] = run, runs sorted by fragment count, heaviest first (LPT).  A launch
// lasts until its last workgroup has finished; in mesh order the units dispatched last are as heavy as any (config 3: 9 ... 1592
// fragments per unit), so the GPU drains for a whole heavy workgroup's lifetime.  With the light runs last the drain is short.  Runs
// are independent (the look-back chain restarts at every run, whose base comes from the table), so any order is correct.
// n_slots = runs rounded up to whole groups of eight (slots past the last run map to themselves: their workgroups exit at once).
void launch_run_order(const unsigned long long* run_base, uint32_t n_runs, const unsigned long long* total, uint32_t* order, uint32_t n_slots, hipStream_t st);
inline uint32_t run_order_slots(uint32_t n_units, uint32_t shift) { return ((n_units + (8u << shift) - 1u) / (8u << shift)) * 8u; }
// multi-pass pipeline (m2s_emit2.hip): count + scan + offsets in one kernel, wave-granular emit
uint32_t emit2_slices(uint64_t limit);       // entries of start[] needed for `limit` output records
uint32_t count_scan_blocks(uint32_t n_tri);  // chain words k_count_scan uses
size_t setup_bytes(uint32_t n_tri);          // per-triangle TriSetup array + the tall-triangle table behind it
size_t setup_tall_offset(uint32_t n_tri);    // where that table's 16-byte header starts (zero when the buffer is allocated)
// total_host (pinned, may be nullptr): the last workgroup also stores the fragment counter there
void launch_count_scan(const SceneDev& sc, uint32_t R, uint32_t* off, uint32_t* start, uint32_t n_start, unsigned long long* chain,
                       uint32_t epoch, unsigned long long* total, void* setup, uint32_t* status, unsigned long long* total_host, hipStream_t st);
void launch_emit2(const SceneDev& sc, uint32_t R, const uint32_t* off, const uint32_t* start, const unsigned long long* total,
                  uint64_t limit, const void* setup, float4* out, hipStream_t st);

// device-side .ply row encoder, formats 1 and 2 (m2s_export.hip)
void launch_encode_rows(const float4* rec, uint64_t n, uint32_t format, float scale_multiplier, uint8_t* out, hipStream_t st);

// XCD runs of k_fused2 / k_sparse.  Hardware workgroup h runs on XCD h % 8 and every XCD has a private L2: consecutive units
// (workgroups' worth of triangles: neighbours on the mesh, neighbouring texels) should meet in ONE L2.  A launch "in runs" gives
// XCD x the runs x, x + 8, x + 16, ... of `1 << shift` consecutive units each; the look-back chain restarts at every run, whose
// first unit reads where the run's output starts from a table (`base`) — so a unit still only waits for units dispatched before
// it (its own run's), whatever the other XCDs are doing.  The table is a by-product: a launch WITHOUT runs (plain order, one
// chain) stores where every run's output starts (`out`), and so does the exact count m2s_upload_scene takes (k_unit_bases); the
// next launches of the scene at that R read it.  No counting pass, no host round trip, no picker kernel.
// (Rounds 1-3 cut the unit list into EIGHT contiguous bands of equal estimated work, one per XCD.  A launch then lasted as long as
// its slowest band — the linear work model left the XCDs' finishing times 4-9 % apart — and as wide as its widest; the cuts had to
// travel to the host.  With ~30 runs per XCD the work evens out statistically and every XCD gets work until the end.)
struct RunInfo {
    const unsigned long long* base;  // launch in runs: base[j] = record index at which the output of run j starts; else nullptr
    unsigned long long* out;         // launch without runs: out[j] = the same, recorded for the next launches (or nullptr)
    uint32_t shift;                  // log2 of the run length in units
    const uint32_t* order;           // launch in runs: dispatch slot -> run (heaviest runs first, see launch_run_order); nullptr: identity
};
// run length for a scene of n_units units (0: too few units for runs to make sense — plain order): 32 units (8192 triangles of
// k_fused2) where that still leaves every XCD four runs, else 16 or 8.  Measured on config 3 (3916 units), kernel time / read
// traffic: runs of 8 / 16 / 32 / 64 units 0.117-0.118 / 0.120 / 0.118-0.119 / 0.124-0.125 ms, eight equal-work bands 0.118-0.119
// (profiles/r04/ab_xcd_runs_vs_equal_work_bands.log); longer runs keep more neighbours in one L2 (fewer texel re-reads), shorter
// ones even the XCDs' work out.
inline uint32_t run_shift_for(uint32_t n_units) {
    if (const char* v = debug_env("M2S_RUN_SHIFT")) { const uint32_t s = (uint32_t)std::atoi(v) & 15u; return n_units >= (32u << s) ? s : 0u; }   // debug: A/B
    for (uint32_t shift = 5; shift >= 3; --shift)
        if (n_units >= (32u << shift)) return shift;      // at least four runs per XCD
    return 0u;
}
inline uint32_t n_runs(uint32_t n_units, uint32_t shift) { return (n_units + (1u << shift) - 1u) >> shift; }
// Work-balanced batches of k_fused2 for scenes that one generation of workgroups converts (fewer than ~172 k triangles): batch b
// = triangles [first[b], first[b + 1]), at most 64, starts at multiples of 8.  first == nullptr: uniform batches of fused_tpw.
struct BatchTable {
    const uint32_t* first;   // device, n + 1 entries
    uint32_t n;
    uint32_t tpw;            // k_fused2 only, first == nullptr: uniform batches of this many triangles instead of fused_tpw (0: fused_tpw);
                             // a multiple of 8, at most 64 — AUTO's choice for scenes of 11-18 fragments per triangle (run_pass: decide)
};
// units (workgroups of 256 triangles) of k_fused2 for a scene of n_tri triangles that can be converted in runs (0: too small for
// 64-triangle batches)
uint32_t fused2_band_workgroups(uint32_t n_tri);
void launch_fused2(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                   unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta,
                   const RunInfo& runs, const BatchTable& batches, hipStream_t st);
// lean form of the team kernel (m2s_fused3.hip): same units, same run tables, same output; only for scenes whose meshes all sample
// a combo texture or no map at all (m2s_ctx::lean_ok); triangles larger than an 8 x 8 pixel box are deferred to k_emit_big
void launch_fused3(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                   unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta,
                   const RunInfo& runs, const BatchTable& batches, hipStream_t st);
// sparse form of the single-pass kernel (m2s_sparse.hip); `runs` as for launch_fused2, in ITS units (512 triangles)
// plane (or nullptr): the positions of the records as a compact plane (16 B each, record order): what a depth sort builds its keys from
void launch_sparse(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                   unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta,
                   const RunInfo& runs, hipStream_t st, float4* plane = nullptr);
bool sparse_supported(uint32_t n_tri);
uint32_t sparse_workgroups(uint32_t n_tri);   // workgroups of kSpCand = 512 triangles
constexpr uint32_t kSparseTrianglesPerWorkgroup = 512;
void launch_emit_big(const SceneDev& sc, uint32_t R, const BigItem* biglist, uint32_t n_big, uint32_t max_cnt, uint64_t limit,
                     float4* out, hipStream_t st);

// ---- viewer prepass (m2s_prepass.hip) --------------------------------------------------------------------------
// kernel-side uniforms: the shader's uniforms plus its per-dispatch invariants, prepared by prepass_prepare()
struct PrepassK {
    float M[16], V[16], P[16];   // u_modelToWorld, u_worldToView, u_viewToClip (column-major)
    float MinvT[16];             // transpose(inverse(u_modelToWorld))
    float mr_inv[9];             // inverse(mat3(u_modelToWorld)), [col*3 + row]
    float ms2[3];                // (modelScale * modelScale)
    float res[2], near_far[2], std_dev;
    int32_t render_mode;
    uint32_t format, ply_has_pbr, depth_test, arrival_order;
    const float* depth;          // device pointer (window-space depth, row 0 = bottom) or nullptr
    uint32_t depth_w, depth_h;
    uint32_t global_w;           // width in invocations of the reference's dispatch (gl_GlobalInvocationID of a linear index)
};
}  // namespace m2s
struct m2s_prepass_params;
namespace m2s {
void prepass_prepare(const m2s_prepass_params& p, uint64_t n, PrepassK* out);
// perm (or nullptr): record index of every position — the prepass then reads record perm[i] at position i (m2s_prepass_sorted);
// dense: every one of the n positions is known to survive (the sort in front applied the frustum test): written at its own position, no append
hipError_t launch_prepass(const PrepassK& k, const float4* rec, uint32_t n, float4* quads, float* depths, unsigned long long* chain,
                          uint32_t epoch, unsigned long long* counter, unsigned long long* total, uint32_t* status, hipStream_t st,
                          const uint32_t* perm = nullptr, bool dense = false);

size_t sort_prepass_temp_bytes(uint32_t n);
hipError_t sort_prepass(const float* depths, const float4* quads, uint32_t n, uint32_t* keys_out, uint32_t* vals_out, void* temp,
                        size_t temp_bytes, float4* sorted, hipStream_t st);
size_t sort_temp_bytes(uint32_t n);
// *key_offset_out (may be NULL: then keys_out is made whole before returning): keys_out holds `key - offset` (the sort ran over the bits in
// which the keys differ); launch_add_to_keys puts it back
hipError_t sort_by_depth(const float4* rec, uint32_t n, const float view[16], uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out,
                         void* temp, size_t temp_bytes, float4* sorted, float4* plane, bool plane_valid, hipEvent_t* stage_ev, hipStream_t st,
                         uint32_t* key_offset_out, uint32_t* pinned_mm /* two pinned host words for the key range, or NULL */);
void launch_add_to_keys(uint32_t* keys, uint32_t n, uint32_t offset, hipStream_t st);
// keys = the depth bits the PREPASS stores (model, view as in PrepassK), stable radix sort, no gather: vals_out = the permutation
// cull: the prepass's frustum test applied to the keys (survivors first: *n_visible of them; *clash: call again without cull); pinned_mm4: FOUR words
hipError_t sort_prepass_permutation(const float4* rec, uint32_t n, const float model[16], const float view[16], const float proj[16], bool cull,
                                    uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, void* temp, size_t temp_bytes, float4* plane, bool plane_valid,
                                    hipEvent_t* ev, hipStream_t st, uint32_t* pinned_mm4, uint32_t* n_visible, bool* clash);

// sample sort across ranks (m2s_dist.cpp): evenly spaced samples of sorted keys; split points of sorted keys
void launch_pick_samples(const uint32_t* keys, uint64_t n, uint32_t s, unsigned long long* out, hipStream_t st);
void launch_lower_bounds(const uint32_t* keys, uint64_t n, const unsigned long long* splitters, uint32_t m, unsigned long long* out, hipStream_t st);

// The runtime loads a file's code object inside the first launch of one of its kernels (0.2-0.6 ms each: the first conversion of
// a PROCESS).  These make it happen now — hipFuncGetAttributes on one kernel of the file — without launching anything:
// m2s_upload_scene asks for the pipeline it has just decided on, m2s_prepare(M2S_PREPARE_KERNELS) for all of them.
hipError_t preload_fused2();
hipError_t preload_fused3();
hipError_t preload_sparse();
hipError_t preload_multipass();
void launch_scratch_warm(hipStream_t st);   // the queue's scratch memory set up now, not inside the first multi-pass conversion (m2s_emit2.hip)
hipError_t preload_export();
hipError_t preload_prepass();
hipError_t preload_sort();

inline uint32_t n_count_blocks(uint32_t n_tri) { return (n_tri + kTriPerBlock - 1) / kTriPerBlock; }
// fused kernel: triangles per wave.  64 as soon as that fills the GPU's 3072 wave slots once.  A smaller scene gets just
// enough triangles per wave to occupy every slot ONCE (one round of waves instead of two: the C2 stand-in, 69 312
// triangles, ran 16 per wave = 4332 waves = 1.4 rounds in round 1), in multiples of 8, at least 8.  With the slots
// filled, fewer triangles per wave only adds instructions (500 k triangles: 0.126 -> 0.156 ms at 32).
#ifndef M2S_TPW_SLOTS
#define M2S_TPW_SLOTS 3072u
#endif
inline uint32_t fused_tpw(uint32_t n_tri) {
    const uint32_t per = (n_tri + M2S_TPW_SLOTS - 1u) / M2S_TPW_SLOTS;
    const uint32_t r8 = (per + 7u) & ~7u;
    return r8 < 8u ? 8u : r8 > 64u ? 64u : r8;
}
inline uint32_t n_fused_waves(uint32_t n_tri) { const uint32_t w = fused_tpw(n_tri); return (n_tri + w - 1) / w; }
// entries (batches + 1) a BatchTable for n_tri triangles may need; 0: the scene takes uniform 64-triangle batches
// (room for one batch per wave slot of the lean kernel — 4096 — where that is more than one per wave of k_fused2)
inline uint32_t batch_table_capacity(uint32_t n_tri) { return fused_tpw(n_tri) < 64u ? (n_fused_waves(n_tri) > 4096u ? n_fused_waves(n_tri) : 4096u) + n_tri / 64u + 16u : 0u; }

}  // namespace m2s
