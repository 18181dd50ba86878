// m2s_sparse.hip — single-pass conversion kernel for meshes with more triangles than fragments (T >> N; gfx950).
//
// BASELINE config 5 (50 M triangles, 24 M Gaussians at R = 2048) lives here: 58 % of its triangles cover no pixel centre, and
// k_fused2 (m2s_fused2.hip) still pays the whole exact triangle phase — geometry-shader setup with its IEEE divisions and
// square roots, snapped raster setup, coverage, fragment constants: ~900 vector instructions per batch of 64 — for every one
// of them, one lane per triangle, so most lanes of most batches compute a result that is "nothing".  Same output, bit for
// bit, as every other pipeline; the difference is who runs the heavy phase:
//
//   tier 1  (every triangle, ~110 instructions, no division, no square root)   load 36 B of positions; pick the projection
//           axis from the unnormalised face normal (only when the pick is numerically clear); approximate window coordinates
//           (one multiply per coordinate); pixel box with a safety margin; if the box holds no pixel centre — or every centre
//           in it is outside one edge by more than the error bound — the triangle CANNOT emit a fragment under the exact
//           arithmetic either and is dropped.  Conservative by construction: anything unclear survives
//           (tools/tier1_check.py: a numpy transcription against the oracle's exact per-triangle counts).
//   compaction  the survivors of the workgroup's 512 candidate triangles are listed in LDS (triangle order).
//   tier 2  (survivors only, dense lanes)   "rounds" of 64 survivors, handed out by an LDS counter: the exact triangle phase
//           of k_fused2, unchanged (geo_setup, raster_setup, coverage mask) — the decision arithmetic is the oracle's.
//   fragment phase   as in k_fused2: one entry stream per workgroup, strips of 64 handed out by an LDS counter, records
//           staged in LDS and written as coalesced non-temporal runs; ONE chain word and one look-back per workgroup.
//
// Rounds replace k_fused2's per-wave batches as the unit of the in-workgroup protocol: round r needs the inclusive prefix of
// round r-1 (fragments, entries) before it can place its entries, publishes its own and carries on; rounds are claimed in
// order by running waves, so every wait is for a wave that is already past its own waits.  Every wait is bounded and raises the
// error flag instead of hanging; so does a workgroup whose entries do not fit the LDS stream — the host (run_pass) then repeats
// the conversion with k_fused2 and remembers that.  Triangles too large for an 8 x 8 pixel box are only counted here and
// handed to k_emit_big through the deferred-triangle list (as k_fused2 does with its big ones).
#include "m2s_fused_common.h"

#pragma clang fp contract(off)

namespace m2s {

#ifndef M2S_SPARSE_WAVES
#define M2S_SPARSE_WAVES 4                    // waves per workgroup (1: every wave on its own, resources released per wave)
#endif
#ifndef M2S_SPARSE_SUB
#define M2S_SPARSE_SUB 8                      // sub-batches of 64 candidate triangles per workgroup (a multiple of the waves)
#endif
#ifndef M2S_SPARSE_ENTRIES
#define M2S_SPARSE_ENTRIES 2560               // entry stream capacity per workgroup (2 B each)
#endif
constexpr int kSpWaves = M2S_SPARSE_WAVES;
constexpr int kSpThreads = kSpWaves * 64;
constexpr int kSpSub = M2S_SPARSE_SUB;
constexpr int kSpPer = kSpSub / kSpWaves;     // sub-batches per wave in tier 1
constexpr uint32_t kSpCand = (uint32_t)kSpSub * 64u;   // candidate triangles per workgroup (= slots: every candidate may survive)
constexpr int kSpRounds = kSpSub;
constexpr uint32_t kSpEntries = M2S_SPARSE_ENTRIES;
#ifndef M2S_SPARSE_STAGE
#define M2S_SPARSE_STAGE 32
#endif
constexpr int kSpStage = M2S_SPARSE_STAGE;    // records staged per wave and round (32: half a strip)
constexpr uint32_t kSpWait = 1u << 24;        // LDS polls before giving up
static_assert(kSpSub % kSpWaves == 0 && kSpSub <= 16, "sub-batches: a multiple of the wave count, slot index must fit 10 bits");

#ifdef M2S_TIMING
// debug build only: per-workgroup cycle counts of wave 0, read back by tools/sparse_timing.py
//   [0] total  [1] until tier 1 is listed  [2] in rounds  [3] of that: waiting for the previous round's prefix  [4] waiting for the
//   last round's count  [5] in strips  [6] of that: waiting for expansions  [7] waiting for the base  [8] rounds of the workgroup
//   [9] entries  [10] survivors  [11] rounds taken by wave 0  [12] strips taken by wave 0
constexpr int kSpTimingSlots = 32, kSpTimingBlocks = 16384;   // [16..21] first round of wave 0, [22..28] its first strip: see tools/sparse_timing.py
__device__ unsigned long long g_sp_timing[kSpTimingSlots * kSpTimingBlocks];
#define SP_T(slot, v) do { if (wave == 0 && lane == 0 && lb < kSpTimingBlocks) g_sp_timing[(slot) * kSpTimingBlocks + lb] = (v); } while (0)
#define SP_NOW() __builtin_amdgcn_s_memtime()
// per XCD: [x] earliest start, [8 + x] latest end of a workgroup (s_memrealtime, 100 MHz, common to all XCDs), [16 + x] sum of
// wave-0 lifetimes, [24 + x] workgroups; reset by the reader (tools/xcd_spans.py)
__device__ unsigned long long g_sp_xcd[32];
#else
#define SP_T(slot, v) do {} while (0)
#define SP_NOW() 0ull
#endif

struct SpLds {
    float4 tri[kSpCand * 4];                  // TriShadeS per survivor slot (slot = position in the survivor list)
    uint32_t tskip[kSpCand];                  // per slot: (record index - stream position) of its fragments
    uint16_t surv[kSpCand];                   // survivor list: candidate index inside the workgroup's range
    uint16_t entries[kSpEntries];             // slot << 6 | bit of the 8 x 8 coverage mask
    float4 stage[kSpWaves][kSpStage * 6];     // record staging, one per wave
    unsigned long long base;                  // record index of stream position 0
    unsigned long long pre_w[kSpRounds + 1];  // fragments (all kinds) of the rounds before k
    uint32_t pre_c[kSpRounds + 1];            // entries of the rounds before k
    uint32_t counted[kSpRounds];              // 1: pre_w / pre_c [k + 1] are valid
    uint32_t expanded[kSpRounds];             // 1: that round's TriShadeS, tskip and entries are in place
    uint32_t ns[kSpSub];                      // survivors per sub-batch
    uint32_t claimed_round, claimed_strip, base_state, irregular, error, done_waves;
};
static_assert(sizeof(SpLds) * (12 / kSpWaves) <= 163840, "twelve waves per CU (three per SIMD) must fit the 160 KB of LDS");

__device__ __forceinline__ uint32_t sp_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sp_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

// ---- tier 1 ------------------------------------------------------------------------------------------------------------
// true: the triangle certainly yields no fragment under the exact arithmetic of geo_setup / raster_setup / the coverage rule.
// (bx, by, bz) = bbox minimum; syz / sxz / sxy = R / range of the three projection planes (converterGS.glsl:360-396).
//
// Why this is safe.  (1) Axis: the exact normal is normalize(cross(normalize(longest edge), another edge)); every pair of edges
// of a triangle has the same cross product up to sign, so its direction is that of c = e1 x e2.  In fp32 either evaluation is
// off by <= ~2^-22 |e_i||e_j| per component, i.e. by <= 2^-22 / sin(angle) relative to |c|; the pick is trusted only if all
// three angles have sin >= 1e-3 (|c|^2 >= 1e-6 Lmax^4) AND the largest |component| beats the second by 1/64 — a thousand times
// the error.  Otherwise: not dropped.  (2) Window coordinates: the exact path rounds (p - bmin) / range * 2 - 1, * R/2 + R/2 in
// four steps and snaps to 1/256; ours is (p - bmin) * (R * rcp(range)).  Both are within a few 2^-24 R of the real value:
// |ours - snapped| <= 0.5/256 + 2.5e-3 (R <= 4096) < d = 0.006 px.  The pixel box is widened by 3/256 > d on every side, so it
// contains the exact box.  (3) Candidate centres: an edge function E(c) = (xb - xa)(cy - ya) - (yb - ya)(cx - xa) moves by at
// most 2d (|cy - ya| + |cx - xa| + |xb - xa| + |yb - ya|) + 4 d^2 when its two vertices move by d per coordinate; a covered centre
// has all three exact edge values >= 0 in the triangle's orientation, so ours are all >= -tol (or all <= tol): if neither holds
// the centre is not covered.  NaN / Inf anywhere makes every comparison false = "not dropped" (or the exact path drops it too).
__device__ __forceinline__ bool tier1_empty(const float p[9], float bx, float by, float bz, float syz, float sxz, float sxy, float Rm1) {
    const float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
    const float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
    const float e3x = p[6] - p[3], e3y = p[7] - p[4], e3z = p[8] - p[5];
    const float cx = fma_(e1y, e2z, -(e1z * e2y)), cy = fma_(e1z, e2x, -(e1x * e2z)), cz = fma_(e1x, e2y, -(e1y * e2x));
    const float ax = fabsf(cx), ay = fabsf(cy), az = fabsf(cz);
    const float mx = fmaxf(ax, fmaxf(ay, az)), md = __builtin_amdgcn_fmed3f(ax, ay, az);
    const float lm = fmaxf(dot3_(e1x, e1y, e1z, e1x, e1y, e1z), fmaxf(dot3_(e2x, e2y, e2z, e2x, e2y, e2z), dot3_(e3x, e3y, e3z, e3x, e3y, e3z)));
    const float cc = dot3_(cx, cy, cz, cx, cy, cz);
    const bool clear = (md < mx * 0.984375f) && (cc >= 1e-6f * lm * lm) && (cc > 1e-30f) && (lm < 1e18f);   // false for NaN
    if (!clear) return false;
    const bool first = (ax > ay) && (ax > az);
    const bool second = !first && (ay > az);
    const bool zb = first || second;                   // B = z, else y
    const float bA = first ? by : bx, bB = zb ? bz : by;
    const float s = first ? syz : (second ? sxz : sxy);
    float x[3], y[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float pa = first ? p[3 * i + 1] : p[3 * i + 0];
        const float pb = zb ? p[3 * i + 2] : p[3 * i + 1];
        x[i] = (pa - bA) * s;
        y[i] = (pb - bB) * s;
    }
    const float mg = 3.0f / 256.0f;
    const float ix0 = fmaxf(ceilf(fminf(x[0], fminf(x[1], x[2])) - 0.5f - mg), 0.0f);
    const float ix1 = fminf(floorf(fmaxf(x[0], fmaxf(x[1], x[2])) - 0.5f + mg), Rm1);
    const float iy0 = fmaxf(ceilf(fminf(y[0], fminf(y[1], y[2])) - 0.5f - mg), 0.0f);
    const float iy1 = fminf(floorf(fmaxf(y[0], fmaxf(y[1], y[2])) - 0.5f + mg), Rm1);
    if (ix0 > ix1 || iy0 > iy1) return true;           // no pixel centre in the (widened) box
    // candidate centres: (ix0 + 0.5 + kx, iy0 + 0.5 + ky), kx in [0, wx], ky in [0, wy].  An edge function is linear, so its
    // extremes over the centres are at the corners: if for some edge even the largest value is below -tol, no centre is inside
    // in the positive orientation; likewise the smallest above +tol for the negative one.  Neither orientation possible: empty.
    const float wx = ix1 - ix0, wy = iy1 - iy0;
    const float pcx = ix0 + 0.5f, pcy = iy0 + 0.5f;
    bool out_pos = false, out_neg = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = (i + 1) % 3;
        const float dx = x[j] - x[i], dy = y[j] - y[i], qx = pcx - x[i], qy = pcy - y[i];
        const float E = fma_(dx, qy, -(dy * qx));
        const float sx = dx * wy, sy = -(dy * wx);
        const float tol = fma_(0.012f, ((fabsf(qy) + wy) + (fabsf(qx) + wx)) + (fabsf(dx) + fabsf(dy)), 2e-4f);
        out_pos = out_pos || ((E + fmaxf(sx, 0.0f)) + fmaxf(sy, 0.0f) < -tol);
        out_neg = out_neg || ((E + fminf(sx, 0.0f)) + fminf(sy, 0.0f) > tol);
    }
    if (out_pos && out_neg) return true;
    return false;
}

// The workgroup's base: sum of the totals of all workgroups before it.  Whoever needs it first resolves it.
__device__ __forceinline__ bool sp_get_base(SpLds& S, const unsigned long long* chain, uint32_t lb, int lane, uint32_t epoch, uint32_t* status,
                                            unsigned long long& base) {
    uint32_t st = sp_load(&S.base_state);
    if (st != 2) {
        uint32_t got = 1;
        if (lane == 0) {
            uint32_t expect = 0;
            got = __hip_atomic_compare_exchange_strong(&S.base_state, &expect, 1u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u;
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got == 0) {   // this wave resolves
            const unsigned long long b = lb == 0 ? 0ull : lookback(chain, lb, lane, epoch, status);
            if (lane == 0) S.base = b;
            sp_store(&S.base_state, 2u);
        } else {
            uint32_t spins = 0;
            while (sp_load(&S.base_state) != 2) {
                if (sp_load(&S.error)) return false;      // (another wave gave up: the launch is discarded, nobody waits)
                if (++spins > kSpWait) { if (lane == 0) sp_store(&S.error, 5u); return false; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    base = S.base;
    return true;
}

__global__ void __launch_bounds__(kSpThreads, 3) k_sparse(SceneDev sc, uint32_t R, unsigned long long* __restrict__ chain, unsigned long long limit,
                                                         float4* __restrict__ out, unsigned long long* __restrict__ total_out,
                                                         uint32_t* __restrict__ status /* [0]=any big, [1]=error */, uint32_t epoch,
                                                         BigItem* __restrict__ biglist, uint32_t* __restrict__ bigmeta, RunInfo runs,
                                                         float4* __restrict__ plane /* or nullptr: positions of the records, 16 B each (m2s_set_keep_positions) */) {
    __shared__ SpLds S;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_wg = (sc.n_tri + kSpCand - 1u) / kSpCand;
    // XCD runs exactly as in k_fused2 (here in units of kSpCand triangles: the table was recorded by a k_sparse launch)
    const uint32_t hb = blockIdx.x, xcd = hb & 7u, rnd = hb >> 3;
    const bool in_runs = runs.base != nullptr;
    const uint32_t rmask = (1u << runs.shift) - 1u;
    uint32_t lb = hb;
    if (in_runs) {
        uint32_t rr = ((rnd >> runs.shift) << 3) + xcd;            // dispatch slot of the run ...
        if (runs.order) rr = ((const __attribute__((address_space(4))) uint32_t*)runs.order)[rr];   // ... heaviest runs first (launch_run_order)
        lb = (rr << runs.shift) + (rnd & rmask);
    }
    const bool band_first = in_runs && (rnd & rmask) == 0u;
    if (lb >= n_wg) return;
    const uint32_t t_wg = lb * kSpCand;
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    [[maybe_unused]] const unsigned long long tk0 = SP_NOW();      // phase timers: live only in -DM2S_TIMING builds
#ifdef M2S_TIMING
    const unsigned long long tk_real0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) atomicMin(&g_sp_xcd[xcd], tk_real0);
#endif
    [[maybe_unused]] unsigned long long tk_rounds = 0, tk_pre = 0, tk_cnt = 0, tk_strips = 0, tk_exp = 0, tk_base = 0, my_rounds = 0, my_strips = 0;

    if (threadIdx.x < (unsigned)kSpRounds) { S.counted[threadIdx.x] = 0; S.expanded[threadIdx.x] = 0; }
    if (threadIdx.x == 0) {
        S.claimed_round = 0; S.claimed_strip = 0; S.irregular = 0; S.error = 0; S.done_waves = 0;
        S.base_state = (lb == 0 || band_first) ? 2u : 0u;
        S.base = band_first ? runs.base[lb >> runs.shift] : 0ull;
        S.pre_w[0] = 0; S.pre_c[0] = 0;
    }

    // ======================= tier 1: every candidate, cheap and conservative =======================
    const uint32_t lastT = min(t_wg + kSpCand, sc.n_tri) - 1u;
    bool uniform_mesh;
    const uint32_t m0 = mesh_of_range(sc, t_wg, lastT, uniform_mesh);   // one scalar load (was: a binary search)
    unsigned long long passm[kSpPer];
    bool pass[kSpPer];
    {
        float pp[kSpPer][9];
        bool val[kSpPer];
#pragma unroll
        for (int k = 0; k < kSpPer; ++k) {     // all position loads of the wave's sub-batches in one round trip
            const uint32_t t = t_wg + (wave + (uint32_t)k * kSpWaves) * 64u + (uint32_t)lane;
            val[k] = t < sc.n_tri;
#pragma unroll
            for (int i = 0; i < 9; ++i) pp[k][i] = 0.0f;
            if (val[k]) load_positions(sc.tri, t, pp[k]);
        }
        const ConstMeshPtr mp0 = kConstMesh(sc.meshes + m0);
        const float bx = mp0->bmin[0], by = mp0->bmin[1], bz = mp0->bmin[2];
        const float ex = mp0->bmax[0] - bx, ey = mp0->bmax[1] - by, ez = mp0->bmax[2] - bz;
        const float ryz = fmaxf(ey, ez), rxz = fmaxf(ex, ez), rxy = fmaxf(ex, ey);
        // the cheap test is only used inside one mesh and with ordinary ranges (else: everything survives)
        const bool cull_ok = uniform_mesh && fminf(ryz, fminf(rxz, rxy)) > 1e-30f && fmaxf(ryz, fmaxf(rxz, rxy)) < 1e30f;
        const float Rf = (float)R;
        const float syz = Rf * fast_rcp(ryz), sxz = Rf * fast_rcp(rxz), sxy = Rf * fast_rcp(rxy);
#pragma unroll
        for (int k = 0; k < kSpPer; ++k) {
            pass[k] = val[k];
            if (cull_ok && val[k]) pass[k] = !tier1_empty(pp[k], bx, by, bz, syz, sxz, sxy, Rf - 1.0f);
            passm[k] = __ballot(pass[k]);
            if (lane == 0) S.ns[wave + (uint32_t)k * kSpWaves] = (uint32_t)__popcll(passm[k]);
        }
    }
    __syncthreads();
    uint32_t NS = 0;
    {
        uint32_t my_off[kSpPer];
#pragma unroll
        for (int j = 0; j < kSpSub; ++j) {
            if ((uint32_t)(j % kSpWaves) == wave) my_off[j / kSpWaves] = NS;
            NS += S.ns[j];
        }
#pragma unroll
        for (int k = 0; k < kSpPer; ++k)
            if (pass[k]) S.surv[my_off[k] + lanes_below(passm[k])] = (uint16_t)((wave + (uint32_t)k * kSpWaves) * 64u + (uint32_t)lane);
    }
    __syncthreads();
    NS = __builtin_amdgcn_readfirstlane(NS);
    SP_T(1, SP_NOW() - tk0);
    const uint32_t nr = (NS + 63u) / 64u;
    const bool knows = lb == 0 || band_first;            // this workgroup's base is known without a look-back
    const unsigned long long before = band_first ? runs.base[lb >> runs.shift] : 0ull;
    if (nr == 0 && wave == 0 && lane == 0)               // nothing survived: the aggregate is zero
        chain_store(&chain[lb], (knows ? kFlagPrefix : kFlagAgg) | etag | (before & kValMask));

    // ======================= tier 2: rounds of 64 survivors, the exact triangle phase =======================
    bool alive = true;
    [[maybe_unused]] const unsigned long long tr0 = SP_NOW();
    // A wave always holds the inputs of its NEXT round in registers: it claims a round, requests that round's positions and
    // texture coordinates, and only then computes the round it claimed before — the 2-3 k cycles of the global round trip
    // are covered by ~4 k cycles of its own arithmetic instead of being exposed at the top of every round.
    auto claim_round = [&]() -> uint32_t {
        uint32_t r = 0;
        if (lane == 0) r = __hip_atomic_fetch_add(&S.claimed_round, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return __builtin_amdgcn_readfirstlane(r);
    };
    float pn[9];
    float4 uvn0 = make_float4(0, 0, 0, 0);
    float2 uvn1 = make_float2(0, 0);
    uint32_t tn = 0;
    bool validn = false;
    auto request_round = [&](uint32_t r) {
        const uint32_t sl = r * 64u + (uint32_t)lane;
        validn = r < nr && sl < NS;
        tn = t_wg + (validn ? (uint32_t)S.surv[sl] : 0u);
#pragma unroll
        for (int i = 0; i < 9; ++i) pn[i] = 0.0f;
        if (validn) {
            load_positions(sc.tri, tn, pn);
            uvn0 = sc.tri.B0[tn];
            uvn1 = sc.tri.B1[tn];
        }
    };
    uint32_t r_next = claim_round();
    request_round(r_next);
    for (;;) {
        const uint32_t r = r_next;
        if (r >= nr) break;
        ++my_rounds;
        [[maybe_unused]] const bool stamp = my_rounds == 1;
        [[maybe_unused]] const unsigned long long q0 = SP_NOW();
        const uint32_t slot = r * 64u + (uint32_t)lane;
        const bool valid = validn;
        const uint32_t t = tn;
        float p[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) p[i] = pn[i];
        const float4 uvb0 = uvn0;
        const float2 uvb1 = uvn1;
        r_next = claim_round();
        request_round(r_next);
        Geo g;
        Raster rs;
        rs.x0 = rs.y0 = 0; rs.x1 = rs.y1 = -1; rs.ext = 0; rs.bias = 0; rs.area2 = 1;
#pragma unroll
        for (int i = 0; i < 3; i++) { rs.a[i] = rs.b[i] = 0; rs.c[i] = 0; }
        bool ok = false;
        uint32_t m = m0;
        if (valid) {
            // (a workgroup inside one mesh — the common case — reads the mesh uniforms with scalar loads)
            if (uniform_mesh) geo_setup_mp(p, kConstMesh(sc.meshes + m0), g);
            else { m = find_mesh(sc, sc.tri_first + t); geo_setup_mp(p, sc.meshes + m, g); }
            ok = raster_setup(g, R, rs);
        }
        if (stamp) SP_T(16, SP_NOW() - q0);   // inputs arrived + geo_setup + raster_setup
        const int w = rs.x1 - rs.x0 + 1, rows = rs.y1 - rs.y0 + 1;
        int kind = kNone;
        unsigned long long mask = 0;
        uint32_t cnt = 0;
        if (ok) {
            if (w <= 8 && rows <= 8 && rs.ext <= 2304) {
                kind = kSmall;
                const long long Px0 = 256ll * rs.x0 + 128, Py0 = 256ll * rs.y0 + 128;
                int e0 = (int)((long long)rs.a[0] * Px0 + (long long)rs.b[0] * Py0 + rs.c[0]) + ((rs.bias >> 0) & 1) - 1;
                int e1 = (int)((long long)rs.a[1] * Px0 + (long long)rs.b[1] * Py0 + rs.c[1]) + ((rs.bias >> 1) & 1) - 1;
                int e2 = (int)((long long)rs.a[2] * Px0 + (long long)rs.b[2] * Py0 + rs.c[2]) + ((rs.bias >> 2) & 1) - 1;
                const int ax0 = rs.a[0] * 256, ax1 = rs.a[1] * 256, ax2 = rs.a[2] * 256;
                const int by0 = rs.b[0] * 256, by1 = rs.b[1] * 256, by2 = rs.b[2] * 256;
                for (int dy = 0; dy < rows; ++dy) {
                    int r0 = e0, r1 = e1, r2 = e2;
                    for (int dx = 0; dx < w; ++dx) {
                        if ((r0 | r1 | r2) >= 0) mask |= 1ull << (dy * 8 + dx);
                        r0 += ax0; r1 += ax1; r2 += ax2;
                    }
                    e0 += by0; e1 += by1; e2 += by2;
                }
                cnt = (uint32_t)__popcll(mask);
            } else {
                kind = kBig;   // emitted by the second stage (k_emit_big); counted here: its slice of the ordered output
                if (rows <= kRowsCount) {
                    RowWalker rw;
                    row_walker_init(rs, rs.y0, rw);
                    for (int y = rs.y0; y <= rs.y1; ++y) {
                        int xa, xb;
                        row_walker_next(rw, xa, xb);
                        cnt += (uint32_t)max(xb - xa + 1, 0);
                    }
                }
            }
        }
        {
            unsigned long long bigm = __ballot(kind == kBig && rows > kRowsCount);
            while (bigm) {
                const int src = __ffsll((long long)bigm) - 1;
                bigm &= bigm - 1;
                const Raster br = shfl_raster(rs, src);
                uint32_t part = 0;
                for (int y = br.y0 + lane; y <= br.y1; y += 64) {
                    int xa, xb;
                    row_span(br, y, xa, xb);
                    part += (uint32_t)max(xb - xa + 1, 0);
                }
                part = wave_sum(part);
                if (lane == src) cnt = part;
            }
        }
        if (stamp) SP_T(17, SP_NOW() - q0);   // + coverage
        if (cnt == 0) kind = kNone;
        const uint32_t cntc = kind == kSmall ? cnt : 0;
        const bool anybig = __ballot(kind == kBig) != 0ull;
        const uint32_t incl = wave_incl_scan(cnt, lane);
        const uint32_t inclc = wave_incl_scan(cntc, lane);
        const unsigned long long total_w = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t total_c = __builtin_amdgcn_readlane(inclc, 63);
        const unsigned long long toff = incl - cnt;
        const uint32_t ctoff = inclc - cntc;

        // the rounds before this one: inclusive prefix of round r - 1 (claimed earlier, by a wave that is past its own wait)
        unsigned long long pw = 0;
        uint32_t pc = 0;
        if (r) {
            [[maybe_unused]] const unsigned long long tw0 = SP_NOW();
            uint32_t spins = 0;
            while (sp_load(&S.counted[r - 1]) == 0) {
                // a wave that gave up (entries do not fit, a wait timed out) has abandoned the round it had claimed ahead:
                // that round is never counted — do not wait for it (this loop used to spin its full 2^24 polls)
                if (sp_load(&S.error)) { alive = false; break; }
                if (++spins > kSpWait) { alive = false; if (lane == 0) sp_store(&S.error, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            pw = S.pre_w[r];
            pc = S.pre_c[r];
            tk_pre += SP_NOW() - tw0;
        }
        if (lane == 0) {
            S.pre_w[r + 1] = pw + total_w; S.pre_c[r + 1] = pc + total_c;
            // deferred triangles make record index != base + stream position for everything AFTER them.  The flag is set before
            // the release store of counted[r] below, so it is visible to every strip whose counted prefix includes this round;
            // strips that start earlier (the strip loop runs ahead of the counting) hold only entries of earlier rounds, whose
            // tskip is 0 either way.  It travels with the counts, not with the later expansion: a strip made of ANOTHER
            // round's entries does not wait for this round's expansion
            if (anybig) __hip_atomic_fetch_or(&S.irregular, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        sp_store(&S.counted[r], 1u);
        if (r == nr - 1u && lane == 0)   // the workgroup's aggregate (or inclusive prefix, where the base is known)
            chain_store(&chain[lb], (knows ? kFlagPrefix : kFlagAgg) | etag | (((knows ? before : 0ull) + pw + total_w) & kValMask));
        if (alive && (unsigned long long)pc + total_c > kSpEntries) {   // does not fit the LDS stream
            alive = false;
            if (lane == 0) sp_store(&S.error, 2u);
        }

        if (stamp) SP_T(18, SP_NOW() - q0);   // + scans, previous round's prefix, publication
        if (alive) {
            if (cntc) {
                TriShade ts;
                if (uniform_mesh) tri_shade_setup(p, g, rs, kConstMesh(sc.meshes + m0), uvb0, uvb1, ts);
                else tri_shade_setup(p, g, rs, sc.meshes + m, uvb0, uvb1, ts);
                TriShadeS c;
                c.a1 = (short)ts.a1; c.b1 = (short)ts.b1; c.a2 = (short)ts.a2; c.b2 = (short)ts.b2;
                c.e1 = (int)ts.e1; c.e2 = (int)ts.e2;
                c.inva = ts.inva; c.sx = ts.sx; c.sy = ts.sy; c.lod0 = ts.lod0;
                c.rot = ts.rot;
                c.lod1 = ts.lod1; c.lod2 = ts.lod2; c.org = ts.org; c.mesh = ts.mesh | m;
                const float4* src = reinterpret_cast<const float4*>(&c);
#pragma unroll
                for (int k = 0; k < 4; ++k) S.tri[slot * 4 + k] = src[k];
                S.tskip[slot] = (uint32_t)((pw + toff) - ((unsigned long long)pc + ctoff));
            }
            if (stamp) SP_T(19, SP_NOW() - q0);   // + fragment constants written
            if (anybig) {   // deferred triangles: reserve their slice of the output, list them for k_emit_big
                unsigned long long base;
                if (sp_get_base(S, chain, lb, lane, epoch, status, base)) {
                    if (kind == kBig) {
                        const uint32_t bslot = atomicAdd(&bigmeta[0], 1u);
                        atomicMax(&bigmeta[1], cnt);
                        atomicAdd(&bigmeta[2], cnt);
                        BigItem it;
                        it.t = t; it.cnt = cnt; it.off = base + pw + toff;
                        biglist[bslot] = it;
                    }
                    if (lane == 0) __hip_atomic_store(&status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else alive = false;
            }
            if (kind == kSmall) {
                unsigned long long mm = mask;
                uint32_t ci = pc + ctoff;
                const uint32_t tag = slot << 6;
                while (mm) {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    S.entries[ci++] = (uint16_t)(tag | (uint32_t)bit);
                }
            }
        }
        if (stamp) SP_T(20, SP_NOW() - q0);   // + entries written
        sp_store(&S.expanded[r], 1u);   // release (set even on error so that nobody waits for it)
        if (!alive) break;
    }

    // ======================= fragment phase: strips of the workgroup's stream =======================
    [[maybe_unused]] uint32_t stream_total = 0;
    unsigned long long out_total = 0;
    tk_rounds = SP_NOW() - tr0;
    [[maybe_unused]] const unsigned long long ts0 = SP_NOW();
    float4* stage = S.stage[wave];
    unsigned long long base = 0;
    bool have_base = false;
    // A wave gets here when no round is left to CLAIM; other waves may still be working on theirs.  Strips do not wait for
    // them: strip s (entries [64 s, 64 s + 64)) is shaded as soon as the rounds counted so far, in order, cover it and the
    // rounds it overlaps have expanded — only the last, partial strip needs the final total.  A wave that drew one round
    // fewer than its neighbours shades the front of the stream meanwhile instead of idling.
    while (alive && sp_load(&S.error) == 0) {
        uint32_t s = 0;
        if (lane == 0) s = __hip_atomic_fetch_add(&S.claimed_strip, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s = __builtin_amdgcn_readfirstlane(s);
        const uint32_t pos0 = s * 64u;
        uint32_t n = 0;
        bool finished = false;
        {   // lane k looks after round k: is it counted, where do its entries end, has it expanded them?
            [[maybe_unused]] const unsigned long long tw0 = SP_NOW();
            uint32_t spins = 0;
            for (;;) {
                bool cnt_k = false, exp_k = false;
                uint32_t end_k = 0, beg_k = 0;
                if ((uint32_t)lane < nr) {
                    cnt_k = sp_load(&S.counted[lane]) != 0;
                    exp_k = sp_load(&S.expanded[lane]) != 0;
                    if (cnt_k) { beg_k = S.pre_c[lane]; end_k = S.pre_c[lane + 1]; }
                }
                const unsigned long long cm = __ballot(cnt_k);
                const uint32_t ncnt = (uint32_t)__builtin_ctzll(~cm);                    // leading counted rounds (counted in order)
                const uint32_t avail = ncnt ? __builtin_amdgcn_readlane(end_k, (int)ncnt - 1) : 0u;   // entries known so far
                bool go = false;
                if (avail >= pos0 + 64u) { n = 64u; go = true; }
                else if (ncnt == nr) {
                    if (pos0 >= avail) { finished = true; break; }
                    n = avail - pos0; go = true;
                }
                if (go) {
                    const bool wait = cnt_k && (uint32_t)lane < ncnt && (beg_k < pos0 + n) && (end_k > pos0) && !exp_k;
                    if (__ballot(wait) == 0ull) {
                        if (ncnt == nr) { stream_total = avail; }
                        break;
                    }
                }
                if (sp_load(&S.error)) { alive = false; break; }      // (rounds abandoned by a wave that gave up are never counted)
                if (++spins > kSpWait) { alive = false; if (lane == 0) sp_store(&S.error, 4u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            tk_exp += SP_NOW() - tw0;
        }
        if (finished) break;
        ++my_strips;
        if (!alive || sp_load(&S.error)) break;
        const bool have = (uint32_t)lane < n;
        uint32_t en = 0;
        if (have) en = S.entries[pos0 + lane];
        const uint32_t slot = en >> 6, bit = en & 63u;
        float4 rec[6];
        uint32_t skip = 0;
        uint32_t my_mesh = 0;
        if (have) my_mesh = reinterpret_cast<const uint32_t*>(&S.tri[slot * 4 + 3])[3] & 0xFFFFFFu;
        const uint32_t m_first = __builtin_amdgcn_readfirstlane(my_mesh);   // lane 0 always holds a fragment
        const bool uniform = sc.n_meshes == 1 || __ballot(have && my_mesh != m_first) == 0ull;
        [[maybe_unused]] const unsigned long long z0 = SP_NOW();
        [[maybe_unused]] unsigned long long st3[3] = { z0, z0, z0 };
        if (have) {
            const TriShadeS& ts = *reinterpret_cast<const TriShadeS*>(&S.tri[slot * 4]);
            const uint32_t tt = t_wg + (uint32_t)S.surv[slot];
            skip = S.tskip[slot];
            const uint32_t org = ts.org;
            const int x = (int)(org & 0xFFFu) + (int)(bit & 7u), y = (int)(org >> 12) + (int)(bit >> 3);
#ifdef M2S_TIMING
            if (uniform) shade_from_tri(sc.tri, tt, x, y, kConstMesh(sc.meshes + m_first), ts, rec, st3);
            else shade_from_tri(sc.tri, tt, x, y, sc.meshes + my_mesh, ts, rec, st3);
#else
            if (uniform) shade_from_tri(sc.tri, tt, x, y, kConstMesh(sc.meshes + m_first), ts, rec);
            else shade_from_tri(sc.tri, tt, x, y, sc.meshes + my_mesh, ts, rec);
#endif
        }
        if (my_strips == 1) { SP_T(22, st3[0] - z0); SP_T(23, st3[1] - z0); SP_T(24, st3[2] - z0); SP_T(25, SP_NOW() - z0); }
        if (!have_base) {
            [[maybe_unused]] const unsigned long long tw0 = SP_NOW();
            if (!sp_get_base(S, chain, lb, lane, epoch, status, base)) break;
            have_base = true;
            tk_base += SP_NOW() - tw0;
        }
        if (sp_load(&S.irregular) == 0) {
            const unsigned long long o0 = base + pos0;
            uint32_t nvalid = n;
            if (o0 + 64ull > limit) {
                if (o0 >= limit) nvalid = 0;
                else if (limit - o0 < nvalid) nvalid = (uint32_t)(limit - o0);
            }
            // the caller is going to sort these records by depth: their positions also go to a compact plane (one coalesced 1 KB store
            // per strip) — the sort's key pass then reads 16 B per record instead of every 128-byte line of the 96-byte records
            if (plane != nullptr && (uint32_t)lane < nvalid) nt_store(&plane[o0 + (uint32_t)lane], rec[0]);
#pragma unroll 1
            for (int part = 0; part < 64 / kSpStage; ++part) {
                if (have && (lane / kSpStage) == part) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) stage[(lane % kSpStage) * 6 + k] = rec[k];
                }
                wave_lds_sync();
                float4* __restrict__ dsto = out + (o0 + (uint32_t)kSpStage * part) * 6;
                const uint32_t nv = nvalid > (uint32_t)kSpStage * part ? min((uint32_t)kSpStage, nvalid - (uint32_t)kSpStage * part) : 0u;
#pragma unroll
                for (int j = 0; j < (kSpStage * 6 + 63) / 64; ++j) {
                    const uint32_t q = (uint32_t)lane + 64u * j;
                    const uint32_t rr = q / 6u;
                    if (rr < nv) nt_store(&dsto[q], stage[q]);
                }
                wave_lds_sync();
            }
        } else if (have) {
            const unsigned long long oidx = base + skip + pos0 + lane;
            if (oidx < limit) {
                float4* __restrict__ dsto = out + oidx * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) nt_store(&dsto[k], rec[k]);
                if (plane != nullptr) nt_store(&plane[oidx], rec[0]);
            }
        }
    }
    tk_strips = SP_NOW() - ts0;
    SP_T(0, SP_NOW() - tk0); SP_T(2, tk_rounds); SP_T(3, tk_pre); SP_T(4, tk_cnt); SP_T(5, tk_strips); SP_T(6, tk_exp); SP_T(7, tk_base);
    SP_T(8, (unsigned long long)nr); SP_T(9, (unsigned long long)stream_total); SP_T(10, (unsigned long long)NS); SP_T(11, my_rounds); SP_T(12, my_strips);
#ifdef M2S_TIMING
    if (threadIdx.x == 0) {
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        atomicMax(&g_sp_xcd[8u + xcd], now);
        atomicAdd(&g_sp_xcd[16u + xcd], now - tk_real0);
        atomicAdd(&g_sp_xcd[24u + xcd], 1ull);
    }
#endif
    // ======================= epilogue: the workgroup's inclusive prefix / the counter (by the last wave to get here) ======
    uint32_t last = 0;
    if (lane == 0) last = __hip_atomic_fetch_add(&S.done_waves, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (uint32_t)kSpWaves - 1u ? 1u : 0u;
    last = __builtin_amdgcn_readfirstlane(last);
    if (last && sp_load(&S.error) == 0) {
        // (every wave has left the strip loop, hence every round is counted: the totals below are final)
        if (nr) { stream_total = S.pre_c[nr]; out_total = S.pre_w[nr]; }
        if (!have_base) have_base = sp_get_base(S, chain, lb, lane, epoch, status, base);
        if (have_base && lane == 0) {
            chain_store(&chain[lb], kFlagPrefix | etag | ((base + out_total) & kValMask));
            if (lb + 1u == n_wg) *total_out = base + out_total;
            if (runs.out && (lb & rmask) == 0u) runs.out[lb >> runs.shift] = base;     // (the run table of the next launches at this R)
        }
    }
    // status[1] != 0 is what the host acts on; the value says why (1 / 3 / 4 / 5: a wait gave up, 2: entries do not fit) and where.
    // The launch is discarded, but its other workgroups still run: leave a chain word behind (if the round that publishes the
    // aggregate was abandoned there is none) so that no successor's look-back waits out its spin limit on this workgroup.
    if (last && sp_load(&S.error) && lane == 0) {
        chain_store(&chain[lb], kFlagPrefix | etag);
        __hip_atomic_store(&status[1], S.error | (lb << 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_sparse(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out, unsigned long long* total,
                   uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta, const RunInfo& runs, hipStream_t st, float4* plane) {
    if (!sc.n_tri) return;
    const uint32_t n_wg = sparse_workgroups(sc.n_tri);
    uint32_t nb = (n_wg + 7u) & ~7u;
    RunInfo r = runs;
    if (r.base) { nb = ((n_wg + (8u << r.shift) - 1u) / (8u << r.shift)) * (8u << r.shift); r.out = nullptr; }
    hipLaunchKernelGGL(k_sparse, dim3(nb), dim3(kSpThreads), 0, st, sc, R, chain, (unsigned long long)limit, out, total, status,
                       epoch & 0xFFFFu, biglist, bigmeta, r, plane);
}

bool sparse_supported(uint32_t n_tri) { return fused_tpw(n_tri) == 64u; }
static_assert(kSpCand == kSparseTrianglesPerWorkgroup, "m2s_device.h");
uint32_t sparse_workgroups(uint32_t n_tri) { return (n_tri + kSpCand - 1u) / kSpCand; }

#ifdef M2S_TIMING
extern "C" int m2s_debug_read_timing_sparse(unsigned long long* dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sp_timing), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
extern "C" int m2s_debug_read_xcd_spans_sparse(unsigned long long* dst /* [32] */) {
    const int e = (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_sp_xcd), 32 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
    unsigned long long init[32];
    for (int i = 0; i < 32; ++i) init[i] = i < 8 ? ~0ull : 0ull;
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sp_xcd), init, sizeof init, 0, hipMemcpyHostToDevice);
    return e;
}
#endif

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_sparse() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_sparse)); }

}  // namespace m2s
