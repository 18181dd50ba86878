// m2s_prepass.hip — the viewer prepass for gfx950: frustum / depth cull of the 96-byte records, 3D
// covariance -> screen-space conic and quad axes, ordered append of the survivors.
//
// Replaces   GaussiansPrepass::execute            (src/renderer/renderPasses/GaussiansPrepass.cpp:8-56)
//            gaussianSplattingPrepassCS.glsl:58-204 + common.glsl:12-92
//
// Shape of the work: stream 96 B in, <= 100 B out per Gaussian, ~300 flops in between: HBM-bound.
//   load     one lane per Gaussian reads its 96-byte record directly (a wave's 64 records are 6144 contiguous bytes, so
//            all fetched lines are fully used)
//   math     one lane per Gaussian, fp32, the shader's operation order (one rounding per operation, no contraction),
//            so that the cull decisions and every stored float equal the reference's (sin / exp of the two debug
//            render modes excepted: library functions)
//   append   the reference takes an atomic counter per survivor (arrival order, nondeterministic).  Here: ballot +
//            popcount per wave, the waves of a workgroup add up through LDS, ONE decoupled look-back per workgroup over a
//            chain word per workgroup (same word format and helper as the conversion kernels), survivors staged in LDS
//            and written as contiguous float4 runs -> output in INPUT order, reproducible, no same-address atomics.
//            (One chain word per WAVE was 2x slower: nothing is known about a wave's survivors before its math is
//            done, so all ~5000 resident waves publish at about the same time and each had to walk back over
//            thousands of not-yet-prefixed words.  With 8 waves per word the walk is one 512-word poll.)
//            m2s_prepass_params.arrival_order = 1 selects the reference's own semantics instead — survivors in arrival
//            order — with one atomic per workgroup: no wait at all, the kernel then runs at the device's copy rate.
#include "../../include/m2s.h"
#include "m2s_fused_common.h"
#include "m2s_viewmath.h"

#include <cmath>
#include <cstring>

#pragma clang fp contract(off)

namespace m2s {

namespace {

struct M3 { float c[3][3]; };   // c[col][row], like the shader's mat3

// mat3 * mat3 in the shader's (glm's) association: a0r*b_c0 + a1r*b_c1 + a2r*b_c2, left to right
__device__ __forceinline__ M3 m3_mul(const M3& a, const M3& b) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) r.c[c][i] = a.c[0][i] * b.c[c][0] + a.c[1][i] * b.c[c][1] + a.c[2][i] * b.c[c][2];
    return r;
}
__device__ __forceinline__ M3 m3_transpose(const M3& a) {
    M3 r;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) r.c[c][i] = a.c[i][c];
    return r;
}
// (mat4 * vec4 and the projection + frustum test of :67-77: m2s_viewmath.h, shared with the depth sort's key kernel)
__device__ __forceinline__ float min_glsl(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max_glsl(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ float clamp01(float x) { return min_glsl(max_glsl(x, 0.0f), 1.0f); }

// common.glsl:12-19
__device__ __forceinline__ float random2d(float cx, float cy) {
    const float a = 12.9898f, b = 78.233f, c = 43758.5453f;
    const float dt = cx * a + cy * b;
    const float sn = dt - 3.14f * floorf(dt / 3.14f);
    const float v = sinf(sn) * c;
    return v - floorf(v);
}

}  // namespace

// gaussianSplattingPrepassCS.glsl:58-204 for one Gaussian (g: its six vec4, gid: its index = the reference's linear invocation
// id).  Returns whether it survives; q / depth_vs are what the shader stores for a survivor.
__device__ __forceinline__ bool prepass_one(const PrepassK& k, const float4 (&g)[6], uint32_t gid, bool valid, float4 (&q)[6], float& depth_vs) {
    const float4 gpos = g[0], gcol = g[1], gscl = g[2], gnrm = g[3], grot = g[4], gpbr = g[5];
    // ---- the shader ------------------------------------------------------------------------------------------
    bool vis = valid;
    float4 ws, vs, pos2d;
    if (!view_project(k.M, k.V, k.P, gpos.x, gpos.y, gpos.z, ws, vs, pos2d)) vis = false;   // :67-77

    if (k.depth_test == 1u && gcol.w > .95f && k.format == 0u && vis) {          // :80-92
        const float u = (pos2d.x / pos2d.w) * 0.5f + 0.5f, v = (pos2d.y / pos2d.w) * 0.5f + 0.5f;
        // GL_NEAREST, CLAMP_TO_EDGE (renderer.cpp:290-296): texel floor(u*W), clamped; NaN -> 0
        const float fu = floorf(u * (float)k.depth_w), fv = floorf(v * (float)k.depth_h);
        const uint32_t ti = fu >= 0.0f ? (fu < (float)k.depth_w ? (uint32_t)fu : k.depth_w - 1u) : 0u;
        const uint32_t tj = fv >= 0.0f ? (fv < (float)k.depth_h ? (uint32_t)fv : k.depth_h - 1u) : 0u;
        const float depth = k.depth[(size_t)tj * k.depth_w + ti];
        const float my_depth = (pos2d.z / pos2d.w) * 0.5f + 0.5f;
        if (my_depth > depth + 0.00002f) vis = false;
    }

    const float multiplier = (k.format == 0u || k.format == 3u) ? k.std_dev : 1.0f;   // :94
    // :95-96  modelScale = (|M[0]|, |M[0]|, |M[1]|) as written; k.ms2 = its square (uniform, prepared on the host)
    const float scale[3] = { (gscl.x * multiplier) * k.ms2[0], (gscl.y * multiplier) * k.ms2[1], (gscl.z * multiplier) * k.ms2[2] };

    M3 rot;                                                                      // :100  castQuatToMat3 on the stored vec4
    {
        const float x = grot.x, y = grot.y, z = grot.z, w = grot.w;
        rot.c[0][0] = 1.f - 2.f * (z * z + w * w);
        rot.c[0][1] = 2.f * (y * z - x * w);
        rot.c[0][2] = 2.f * (y * w + x * z);
        rot.c[1][0] = 2.f * (y * z + x * w);
        rot.c[1][1] = 1.f - 2.f * (y * y + w * w);
        rot.c[1][2] = 2.f * (z * w - x * y);
        rot.c[2][0] = 2.f * (y * w - x * z);
        rot.c[2][1] = 2.f * (z * w + x * y);
        rot.c[2][2] = 1.f - 2.f * (y * y + z * z);
    }
    M3 mri;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) mri.c[c][i] = k.mr_inv[c * 3 + i];
    rot = m3_mul(rot, mri);                                                      // :102-108
    M3 cov3d;                                                                    // :110  computeCov3D
    {
        M3 sm;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int i = 0; i < 3; ++i) sm.c[c][i] = (c == i) ? scale[c] : 0.0f;
        const M3 mm = m3_mul(sm, rot);
        cov3d = m3_mul(m3_transpose(mm), mm);
    }

    float4 out_color = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 nrm = make_float4(1.f, 0.f, 0.f, 0.f);
    if (k.format == 0u || (k.format == 1u && k.ply_has_pbr != 0u) || k.format == 3u) {   // :118-122
        const float4 nw = m4_mul(k.MinvT, gnrm.x, gnrm.y, gnrm.z, 1.0f);
        nrm = make_float4(nw.x * 0.5f + 0.5f, nw.y * 0.5f + 0.5f, nw.z * 0.5f + 0.5f, gcol.w);
    } else if (k.format == 1u) {                                                 // :124-131
        const uint32_t mi = (uint32_t)((gscl.y < gscl.z) && (gscl.y < gscl.x)) + (uint32_t)((gscl.z < gscl.y) && (gscl.z < gscl.x)) * 2u;
        const float a0 = mi == 0u ? rot.c[0][0] : mi == 1u ? rot.c[1][0] : rot.c[2][0];
        const float a1 = mi == 0u ? rot.c[0][1] : mi == 1u ? rot.c[1][1] : rot.c[2][1];
        const float a2 = mi == 0u ? rot.c[0][2] : mi == 1u ? rot.c[1][2] : rot.c[2][2];
        nrm = make_float4(a0 * 0.5f + 0.5f, a1 * 0.5f + 0.5f, a2 * 0.5f + 0.5f, gcol.w);
    }
    if (k.render_mode == 0 || k.render_mode == 6) out_color = gcol;               // :133-149
    else if (k.render_mode == 1) {
        const float nd = ((-vs.z) - k.near_far[0]) / (k.near_far[1] - k.near_far[0]);   // common.glsl:78-82
        const float cd = clamp01(expf(-20.0f * clamp01(nd)));
        out_color = make_float4(cd, cd, cd, gcol.w);
    } else if (k.render_mode == 2) out_color = nrm;
    if (k.render_mode == 3) {
        // gl_GlobalInvocationID of the reference's dispatch (GaussiansPrepass.cpp:44-49; 16x16 local size)
                const float fx = (float)(gid % k.global_w), fy = (float)(gid / k.global_w);
        out_color = make_float4(random2d(fx, fy), random2d(fy, fx), random2d(fy * 1.234f, fx * 1.234f), 1.0f);
    }

    pos2d.x = pos2d.x / pos2d.w; pos2d.y = pos2d.y / pos2d.w; pos2d.z = pos2d.z / pos2d.w;   // :151

    const float p00 = k.P[0], p11 = k.P[5], p32 = k.P[14];
    const float tz_sq = vs.z * vs.z;                                             // :154-159
    const float jsx = -(p00 * k.res[0]) / (2.0f * vs.z);
    const float jsy = -(p11 * k.res[1]) / (2.0f * vs.z);
    const float jtx = (p00 * vs.x * k.res[0]) / (2.0f * tz_sq);
    const float jty = (p11 * vs.y * k.res[1]) / (2.0f * tz_sq);
    const float jtz = ((k.near_far[1] - k.near_far[0]) * p32) / (2.0f * tz_sq);
    M3 J, W;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < 3; ++i) { J.c[c][i] = 0.0f; W.c[c][i] = k.V[c * 4 + i]; }   // :161-165
    J.c[0][0] = jsx; J.c[1][1] = jsy; J.c[2][0] = jtx; J.c[2][1] = jty; J.c[2][2] = jtz;
    const M3 JW = m3_mul(J, W);
    const M3 Vp = m3_mul(m3_mul(JW, cov3d), m3_transpose(JW));                   // :168
    float c00 = Vp.c[0][0];
    const float c01 = Vp.c[0][1], c10 = Vp.c[1][0];
    float c11 = Vp.c[1][1];                                                      // :170
    c00 += 0.3f;                                                                 // :173-174
    c11 += 0.3f;
    const float mid = c00 + c11;
    const float da = c00 - c11, db = 2.0f * c01;
    const float delta = sqrtf(da * da + db * db);                                // :178
    const float lambda1 = 0.5f * (mid + delta), lambda2 = 0.5f * (mid - delta);
    if (lambda2 < 0.0f) vis = false;                                             // :183

    const float dvy = (-c00 + c01 + lambda1) / (c01 - c11 + lambda1);            // :185
    const float inv_len = 1.0f / sqrtf(1.0f * 1.0f + dvy * dvy);
    const float dx = 1.0f * inv_len, dy = dvy * inv_len;
    const float major_r = min_glsl(3.0f * sqrtf(lambda1), 1024.0f), minor_r = min_glsl(3.0f * sqrtf(lambda2), 1024.0f);
    const float hx = k.res[0] * 0.5f, hy = k.res[1] * 0.5f;                      // :186-190
    const float4 quad_scale = make_float4((major_r * dx) / hx, (major_r * dy) / hy, (minor_r * dy) / hx, (minor_r * (-dx)) / hy);
    const float det = c00 * c11 - c01 * c10;                                     // common.glsl:61-76
    float i00 = 0.0f, i01 = 0.0f, i11 = 0.0f;
    if (det != 0.0f) { i00 = c11 / det; i01 = -c01 / det; i11 = c00 / det; }

    q[0] = pos2d;                                                                // :194-202
    q[1] = quad_scale;
    q[2] = out_color;
    q[3] = make_float4(i00, i01, i11, -vs.z);
    q[4] = make_float4(nrm.x, nrm.y, nrm.z, gpbr.x);
    q[5] = make_float4(ws.x, ws.y, ws.z, gpbr.y);
    depth_vs = vs.z;                                                             // :204
    return vis;
}

constexpr uint32_t kPPDelayGrid = 2048;     // grids at least this large (~3 rounds of resident workgroups) delay their look-back
constexpr int kPPWaves = 8;                 // waves per workgroup = per chain word (4: -20 %, 16: -8 %)
// Gaussians per lane (a wave takes kPPRec x 64 consecutive records).  2 was measured: more bytes in flight per workgroup help
// when many Gaussians are culled (close-up, input order: 0.129 -> 0.115 ms; 685 k records: 0.043 -> 0.037 ms) but not with
// everything in view (0.145 -> 0.143 ms), and cost in arrival order (0.103 -> 0.110 ms) and on small inputs (fewer
// workgroups than CUs).  Kept at 1.
constexpr int kPPRec = 1;
constexpr uint32_t kPPTile = kPPWaves * 64u * kPPRec;   // records per workgroup

__global__ void __launch_bounds__(kPPWaves * 64) k_prepass(const PrepassK k, const float4* __restrict__ rec, uint32_t n, float4* __restrict__ quads,
                                                    float* __restrict__ depths, unsigned long long* __restrict__ chain, uint32_t epoch,
                                                    unsigned long long* __restrict__ counter, unsigned long long* __restrict__ total,
                                                    uint32_t* __restrict__ status, const uint32_t* __restrict__ perm, uint32_t dense) {
    __shared__ float4 s_rec[kPPWaves][64 * 6];   // survivors of ONE 64-record group, staged for contiguous stores
    __shared__ float s_depth[kPPWaves][64];
    __shared__ uint32_t s_cnt[kPPWaves];
    __shared__ unsigned long long s_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t first = blockIdx.x * kPPTile + (uint32_t)wave * (64u * kPPRec);   // this wave's first record
    float4* S = s_rec[wave];

    // ---- load + math.  Each lane reads its own records (six 16-byte loads each, 96-byte lane stride; a group's 6 KiB are
    // contiguous, so every fetched line is fully used — 8 % faster than 1 KiB-coalesced loads transposed through LDS).  All
    // groups' loads are issued before the first math: kPPRec x 96 bytes in flight per lane.  The survivors of group 0 go to
    // the staging area at once; those of the later groups wait in registers until group 0 has been written out.
    // With a permutation (m2s_prepass_sorted: the depth sort was taken first) position i holds record perm[i]: the same six 16-byte
    // loads per lane, from a 96-byte record anywhere in the buffer — the gather of the sort and the prepass's read in one pass.
    float4 g[kPPRec][6];
    bool valid[kPPRec];
    uint32_t src[kPPRec];
#pragma unroll
    for (int r = 0; r < kPPRec; ++r) {
        const uint32_t gid = first + (uint32_t)r * 64u + (uint32_t)lane;
        valid[r] = gid < n;
        src[r] = valid[r] ? (perm ? perm[gid] : gid) : 0u;
        const float4* gsrc = rec + (size_t)src[r] * 6;
#pragma unroll
        for (int j = 0; j < 6; ++j) g[r][j] = gsrc[j];      // (non-temporal LOADS of the records: measured, no difference)
    }
    float4 q[kPPRec][6];
    float dvs[kPPRec];
    bool vis[kPPRec];
    uint32_t cnt[kPPRec], rank[kPPRec], wcnt = 0;
#pragma unroll
    for (int r = 0; r < kPPRec; ++r) {
        vis[r] = prepass_one(k, g[r], perm ? src[r] : first + (uint32_t)r * 64u + (uint32_t)lane, valid[r], q[r], dvs[r]);   // (gid = the RECORD's index)
        if (dense) {
            // the sort in front of this launch already applied the frustum test (same function: m2s_viewmath.h) and put the survivors first:
            // every one of the n positions survives, position i is written at i.  A disagreement would be a bug: reported, never silent.
            if (valid[r] && !vis[r]) __hip_atomic_store(&status[1], 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            vis[r] = valid[r];
        }
        const unsigned long long mask = __ballot(vis[r]);
        cnt[r] = (uint32_t)__popcll(mask);
        rank[r] = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        wcnt += cnt[r];
        if (r == 0 && vis[0]) {
            float4* o = S + rank[0] * 6;
#pragma unroll
            for (int j = 0; j < 6; ++j) o[j] = q[0][j];
            s_depth[wave][rank[0]] = dvs[0];
        }
    }
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    if (lane == 0) s_cnt[wave] = wcnt;
    __syncthreads();
    if (dense) {
        // no append: the workgroup's first position is its base, and the total is known to the host
        if (wave == 0 && lane == 0) s_base = (unsigned long long)blockIdx.x * kPPTile;
    } else
    if (wave == 0) {
        const uint32_t bid = blockIdx.x, last = gridDim.x - 1u;
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kPPWaves; ++w) tot += s_cnt[w];
        if (k.arrival_order) {                         // the reference's append: one atomic per WORKGROUP (it takes one per Gaussian)
            if (lane == 0) s_base = atomicAdd(counter, (unsigned long long)tot);
        } else {                                       // input order: one look-back per workgroup
            if (lane == 0 && bid != last) chain_store(&chain[bid], kFlagAgg | etag | tot);
            // Polls issued before the predecessors' words are visible are wasted round trips through memory (4 KiB of uncached
            // reads per workgroup and poll) that only delay the successful one.  On a grid of several rounds of resident
            // workgroups, first sleep ~5 us (two uncached round trips under load), then watch ONE word — the immediate
            // predecessor's — and only then take the wide look-back.  Measured on 2.74 M records: 0.171 -> 0.145 ms (delay
            // 0 / 100 / 200 / 300 / 400 x 64 clocks: 0.171 / 0.151 / 0.146 / 0.156 / 0.170 ms); small grids are not delayed
            // (170 k records: 0.0166 ms without, 0.0189 ms with).
            if (gridDim.x >= kPPDelayGrid) {
                __builtin_amdgcn_s_sleep(100);
                __builtin_amdgcn_s_sleep(100);
            }
            if (bid != 0u) {
                uint32_t spins = 0;
                while (chain_flag(chain_load(&chain[bid - 1u]), epoch) == 0u && ++spins < kSpinLimit) __builtin_amdgcn_s_sleep(20);
            }
            const unsigned long long b = bid == 0u ? 0ull : lookback(chain, bid, lane, epoch, status);
            if (lane == 0) {
                if (bid != last) chain_store(&chain[bid], kFlagPrefix | etag | ((b + tot) & kValMask));
                else __hip_atomic_store(total, b + tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // the counter read-back
                s_base = b;
            }
        }
    }
    __syncthreads();
    unsigned long long base = s_base;
#pragma unroll
    for (int w = 0; w < kPPWaves; ++w) base += (w < wave) ? s_cnt[w] : 0u;
    // ---- stores: group after group through the staging area, each a contiguous run of float4 ---------------------------------
#pragma unroll
    for (int r = 0; r < kPPRec; ++r) {
        if (r > 0) {
            wave_lds_sync();                          // the previous group has left the staging area
            if (vis[r]) {
                float4* o = S + rank[r] * 6;
#pragma unroll
                for (int j = 0; j < 6; ++j) o[j] = q[r][j];
                s_depth[wave][rank[r]] = dvs[r];
            }
            wave_lds_sync();
        }
        float4* dst = quads + (size_t)base * 6;
        const uint32_t n4 = cnt[r] * 6u;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const uint32_t idx = (uint32_t)j * 64u + (uint32_t)lane;
            if (idx < n4) nt_store(&dst[idx], S[idx]);       // (the quads are read by the next pass, not by this one: non-temporal, like the records)
        }
        if ((uint32_t)lane < cnt[r]) __builtin_nontemporal_store(s_depth[wave][lane], &depths[base + lane]);
        base += cnt[r];
    }
}

hipError_t launch_prepass(const PrepassK& k, const float4* rec, uint32_t n, float4* quads, float* depths, unsigned long long* chain,
                          uint32_t epoch, unsigned long long* counter, unsigned long long* total, uint32_t* status, hipStream_t st,
                          const uint32_t* perm, bool dense) {
    const uint32_t nb = (n + kPPTile - 1u) / kPPTile;
    hipLaunchKernelGGL(k_prepass, dim3(nb), dim3(kPPWaves * 64), 0, st, k, rec, n, quads, depths, chain, epoch & 0xFFFFu, counter, total, status, perm,
                       dense ? 1u : 0u);
    return hipGetLastError();
}

// ---- host: the uniforms and loop invariants of the shader ----------------------------------------------------------
// transpose(inverse(u_modelToWorld)) (:120) and inverse(mat3(u_modelToWorld)) (:102-108) do not depend on the Gaussian:
// computed once here, with the operation order of the vendored glm 1.0.1 (glm/detail/func_matrix.inl:322-405) because the
// reference's CPU-side execution of the shader — what the parity tests compare against — goes through glm.
namespace {
inline float M4(const float* m, int c, int r) { return m[c * 4 + r]; }
}

void prepass_prepare(const m2s_prepass_params& p, uint64_t n, PrepassK* out) {
    PrepassK& k = *out;
    memcpy(k.M, p.model_to_world, sizeof k.M);
    memcpy(k.V, p.world_to_view, sizeof k.V);
    memcpy(k.P, p.view_to_clip, sizeof k.P);
    const float* m = p.model_to_world;
    {   // 4x4 inverse by cofactors, then transpose
        const float C00 = M4(m, 2, 2) * M4(m, 3, 3) - M4(m, 3, 2) * M4(m, 2, 3), C02 = M4(m, 1, 2) * M4(m, 3, 3) - M4(m, 3, 2) * M4(m, 1, 3);
        const float C03 = M4(m, 1, 2) * M4(m, 2, 3) - M4(m, 2, 2) * M4(m, 1, 3), C04 = M4(m, 2, 1) * M4(m, 3, 3) - M4(m, 3, 1) * M4(m, 2, 3);
        const float C06 = M4(m, 1, 1) * M4(m, 3, 3) - M4(m, 3, 1) * M4(m, 1, 3), C07 = M4(m, 1, 1) * M4(m, 2, 3) - M4(m, 2, 1) * M4(m, 1, 3);
        const float C08 = M4(m, 2, 1) * M4(m, 3, 2) - M4(m, 3, 1) * M4(m, 2, 2), C10 = M4(m, 1, 1) * M4(m, 3, 2) - M4(m, 3, 1) * M4(m, 1, 2);
        const float C11 = M4(m, 1, 1) * M4(m, 2, 2) - M4(m, 2, 1) * M4(m, 1, 2), C12 = M4(m, 2, 0) * M4(m, 3, 3) - M4(m, 3, 0) * M4(m, 2, 3);
        const float C14 = M4(m, 1, 0) * M4(m, 3, 3) - M4(m, 3, 0) * M4(m, 1, 3), C15 = M4(m, 1, 0) * M4(m, 2, 3) - M4(m, 2, 0) * M4(m, 1, 3);
        const float C16 = M4(m, 2, 0) * M4(m, 3, 2) - M4(m, 3, 0) * M4(m, 2, 2), C18 = M4(m, 1, 0) * M4(m, 3, 2) - M4(m, 3, 0) * M4(m, 1, 2);
        const float C19 = M4(m, 1, 0) * M4(m, 2, 2) - M4(m, 2, 0) * M4(m, 1, 2), C20 = M4(m, 2, 0) * M4(m, 3, 1) - M4(m, 3, 0) * M4(m, 2, 1);
        const float C22 = M4(m, 1, 0) * M4(m, 3, 1) - M4(m, 3, 0) * M4(m, 1, 1), C23 = M4(m, 1, 0) * M4(m, 2, 1) - M4(m, 2, 0) * M4(m, 1, 1);
        const float F[6][4] = { { C00, C00, C02, C03 }, { C04, C04, C06, C07 }, { C08, C08, C10, C11 },
                                { C12, C12, C14, C15 }, { C16, C16, C18, C19 }, { C20, C20, C22, C23 } };
        float Vv[4][4];
        for (int r = 0; r < 4; ++r) { Vv[r][0] = M4(m, 1, r); Vv[r][1] = Vv[r][2] = Vv[r][3] = M4(m, 0, r); }
        float inv[4][4];
        for (int i = 0; i < 4; ++i) {
            const float sa = (i & 1) ? -1.0f : 1.0f, sb = -sa;
            inv[0][i] = (Vv[1][i] * F[0][i] - Vv[2][i] * F[1][i] + Vv[3][i] * F[2][i]) * sa;
            inv[1][i] = (Vv[0][i] * F[0][i] - Vv[2][i] * F[3][i] + Vv[3][i] * F[4][i]) * sb;
            inv[2][i] = (Vv[0][i] * F[1][i] - Vv[1][i] * F[3][i] + Vv[3][i] * F[5][i]) * sa;
            inv[3][i] = (Vv[0][i] * F[2][i] - Vv[1][i] * F[4][i] + Vv[2][i] * F[5][i]) * sb;
        }
        const float d0 = M4(m, 0, 0) * inv[0][0], d1 = M4(m, 0, 1) * inv[1][0], d2 = M4(m, 0, 2) * inv[2][0], d3 = M4(m, 0, 3) * inv[3][0];
        const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
        for (int c = 0; c < 4; ++c)
            for (int i = 0; i < 4; ++i) k.MinvT[i * 4 + c] = inv[c][i] * ood;      // transposed on the way out
    }
    {   // 3x3 inverse of the upper-left block
#define A(c_, r_) M4(m, c_, r_)
        const float ood = 1.0f / (+A(0, 0) * (A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) - A(1, 0) * (A(0, 1) * A(2, 2) - A(2, 1) * A(0, 2)) +
                                  A(2, 0) * (A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2)));
        float* r = k.mr_inv;   // r[c*3 + row]
        r[0 * 3 + 0] = +(A(1, 1) * A(2, 2) - A(2, 1) * A(1, 2)) * ood;
        r[1 * 3 + 0] = -(A(1, 0) * A(2, 2) - A(2, 0) * A(1, 2)) * ood;
        r[2 * 3 + 0] = +(A(1, 0) * A(2, 1) - A(2, 0) * A(1, 1)) * ood;
        r[0 * 3 + 1] = -(A(0, 1) * A(2, 2) - A(2, 1) * A(0, 2)) * ood;
        r[1 * 3 + 1] = +(A(0, 0) * A(2, 2) - A(2, 0) * A(0, 2)) * ood;
        r[2 * 3 + 1] = -(A(0, 0) * A(2, 1) - A(2, 0) * A(0, 1)) * ood;
        r[0 * 3 + 2] = +(A(0, 1) * A(1, 2) - A(1, 1) * A(0, 2)) * ood;
        r[1 * 3 + 2] = -(A(0, 0) * A(1, 2) - A(1, 0) * A(0, 2)) * ood;
        r[2 * 3 + 2] = +(A(0, 0) * A(1, 1) - A(1, 0) * A(0, 1)) * ood;
#undef A
    }
    // :95  vec3(length(M[0]), length(M[0]), length(M[1])) squared, vec4 lengths: (x*x + y*y) + (z*z + w*w)
    const float l0 = sqrtf((m[0] * m[0] + m[1] * m[1]) + (m[2] * m[2] + m[3] * m[3]));
    const float l1 = sqrtf((m[4] * m[4] + m[5] * m[5]) + (m[6] * m[6] + m[7] * m[7]));
    k.ms2[0] = l0 * l0; k.ms2[1] = l0 * l0; k.ms2[2] = l1 * l1;
    k.res[0] = (float)p.resolution[0]; k.res[1] = (float)p.resolution[1];         // glm::ivec2 -> vec2 (GaussiansPrepass.cpp:23)
    k.near_far[0] = p.near_far[0]; k.near_far[1] = p.near_far[1];
    k.std_dev = p.gaussian_std / (float)p.resolution_target;                      // GaussiansPrepass.cpp:18
    k.render_mode = p.render_mode;
    k.format = p.format;
    k.ply_has_pbr = p.ply_has_pbr;
    k.depth_test = p.depth_test_mesh;
    k.arrival_order = p.arrival_order;
    k.depth = nullptr; k.depth_w = p.depth_w; k.depth_h = p.depth_h;
    // GaussiansPrepass.cpp:44-49: groupsX = ceil(sqrt(groups of 256)), 16 invocations wide each
    const uint32_t groups = (uint32_t)((n + 255u) / 256u);
    const uint32_t gx = (uint32_t)std::ceil(std::sqrt((float)groups));
    k.global_w = gx ? gx * 16u : 16u;
}

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_prepass() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_prepass)); }

}  // namespace m2s
