// m2s_fused2.hip — single-pass conversion kernel, workgroup-cooperative ("team") form (gfx950).
//
// Its predecessor (k_fused, rounds 1-5, removed in round 6) ran every WAVE alone: triangle phase for its 64
// triangles, look-back, then strips of 64 of ITS OWN fragments — and the last strip of every wave is partial: on
// the C3 workload 64 triangles give 9..452 fragments, so only 85 % of the fragment lanes do useful work, and every
// wave pays its own look-back round trip.  Here the four waves of a workgroup still run the triangle phase
// independently (one batch of 64 triangles each, aggregates published as early as possible), but their fragments go
// into ONE entry stream for the workgroup (LDS), in canonical order, and the fragment phase takes strips of 64 from
// that stream (an LDS atomic hands them out): only the workgroup's last strip is partial (96 % useful lanes), the
// look-back is needed once per workgroup — record index = workgroup base + stream position — and a wave whose batch
// was light helps with the fragments of a heavy one.
//
// Synchronisation inside the workgroup is LDS-only: per-wave "counted" and "expanded" flags (release stores / acquire
// loads); a wave waits for the COUNTS of the waves before it (to know where its entries go) and, per strip, for the
// EXPANSION of the waves whose entries the strip contains.  No __syncthreads.  Every wait is bounded and raises the
// error flag instead of hanging; so does a workgroup whose fragments do not fit the LDS stream (kEntries) — the host
// then repeats the conversion with the multi-pass pipeline (m2s_pass.cpp, run_pass) and remembers that for the scene and R.
#include "m2s_fused_common.h"
#include <cstdio>

#pragma clang fp contract(off)

namespace m2s {

// Four waves = one per SIMD.  (Six, -DM2S_TEAM_WAVES=6: 98 % strip fill and two workgroups per CU by LDS, but the six
// waves of a workgroup land 2/2/1/1 on the four SIMDs and the kernel takes 0.197 ms instead of 0.135.)
#ifndef M2S_TEAM_WAVES
#define M2S_TEAM_WAVES 4
#endif
#ifndef M2S_FUSED2_WAVES
#define M2S_FUSED2_WAVES 3                 // waves per SIMD the kernel is compiled for
#endif
#ifndef M2S_FUSED2_ENTRIES
#define M2S_FUSED2_ENTRIES 1024            // entry-stream capacity per wave of the team (x4 bytes x kTeam of LDS)
#endif
#ifndef M2S_FUSED2_STAGE
#define M2S_FUSED2_STAGE 32                // records staged per wave and round (32 = half a strip, 16 = a quarter)
#endif
constexpr int kTeam = M2S_TEAM_WAVES;      // waves (= batches of 64 triangles) per workgroup
constexpr int kTeamThreads = kTeam * 64;
constexpr uint32_t kEntries = (uint32_t)M2S_FUSED2_ENTRIES * kTeam;   // entry stream capacity per workgroup (4 B each)
constexpr int kStageRec = M2S_FUSED2_STAGE;
constexpr uint32_t kInvalidEntry = 0xFFFFFFFFu;
constexpr uint32_t kWaitLimit = 1u << 24;  // LDS polls before giving up

#ifdef M2S_TIMING
// debug build only: per-workgroup cycle counts of wave 0, read back by tools/team_timing.py
//   [0] total, [1] waiting for counts, [2] waiting for entries, [3] waiting for the base, [4] strips, [5] entries of the workgroup,
//   [6] start and [7] end of wave 0 (s_memrealtime: 100 MHz, common to all XCDs), [8] XCD (blockIdx & 7), [11] until the counts are published
constexpr int kF2TimingSlots = 16, kF2TimingBlocks = 8192;
__device__ unsigned long long g_f2_timing[kF2TimingSlots * kF2TimingBlocks];
#define F2_T(slot, v) do { if (wave == 0 && lane == 0 && lb < kF2TimingBlocks) g_f2_timing[(slot) * kF2TimingBlocks + lb] = (v); } while (0)
#define F2_TW(w, slot, v) do { if (wave == (w) && lane == 0 && lb < kF2TimingBlocks) g_f2_timing[(slot) * kF2TimingBlocks + lb] = (v); } while (0)
#define F2_NOW() __builtin_amdgcn_s_memtime()
#else
#define F2_TW(w, slot, v) do {} while (0)
#define F2_T(slot, v) do {} while (0)
#define F2_NOW() 0ull
#endif

// control words of the workgroup's unit of work (kTeam batches)
struct F2Ctl {
    unsigned long long base;               // record index of stream position 0
    unsigned long long total_w[kTeam];     // fragments (all kinds) per batch
    uint32_t total_c[kTeam];               // entries per batch
    uint32_t counted[kTeam];               // 1: total_w / total_c of that wave are valid
    uint32_t expanded[kTeam];              // 1: that wave's TriShade, tskip and entries are in place
    uint32_t t0[kTeam];                    // first triangle of each wave's batch (batches need not be equally long: BatchTable)
    uint32_t claimed;                      // next strip to hand out
    uint32_t base_state;                   // 0 unknown, 1 being resolved, 2 known
    uint32_t irregular;                    // 1: record index != base + stream position somewhere (deferred triangles)
    uint32_t error;
};
struct F2Lds {
    float4 tri[kTeam][64 * 5];             // TriShade of the four batches
    uint32_t tskip[kTeam][64];             // per triangle: (record index - stream position) of its fragments
    uint32_t entries[kEntries];            // lane << 24 | y << 12 | x  (the owning wave follows from the stream position)
    float4 stage[kTeam][kStageRec * 6];    // record staging, one per wave (half a strip, or a quarter)
    F2Ctl ctl;
};

__device__ __forceinline__ uint32_t lds_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

// The workgroup's base: sum of the totals of all batches before its first one.  Whoever needs it first resolves it.
__device__ __forceinline__ bool f2_get_base(F2Ctl& S, const unsigned long long* chain, unsigned long long* chain_w, uint32_t b0, int lane,
                                            uint32_t epoch, uint32_t* status, unsigned long long& base) {
    uint32_t st = lds_load(&S.base_state);
    if (st != 2) {
        uint32_t got = 1;
        if (lane == 0) {
            uint32_t expect = 0;
            got = __hip_atomic_compare_exchange_strong(&S.base_state, &expect, 1u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u;
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got == 0) {   // this wave resolves
            const unsigned long long b = b0 == 0 ? 0ull : lookback(chain, b0, lane, epoch, status);
            if (lane == 0) {
                S.base = b;
                // the first batch's inclusive prefix: successors' look-backs stop here
                chain_store(&chain_w[b0], kFlagPrefix | ((unsigned long long)epoch << kEpochShift) | ((b + S.total_w[0]) & kValMask));
            }
            lds_store(&S.base_state, 2u);
        } else {
            uint32_t spins = 0;
            while (lds_load(&S.base_state) != 2) {
                if (++spins > kWaitLimit) { if (lane == 0) lds_store(&S.error, 1u); return false; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    base = S.base;
    return true;
}

__global__ void __launch_bounds__(kTeamThreads, M2S_FUSED2_WAVES) k_fused2(SceneDev sc, uint32_t R, unsigned long long* __restrict__ chain,
                                                      unsigned long long limit, float4* __restrict__ out,
                                                      unsigned long long* __restrict__ total_out,
                                                      uint32_t* __restrict__ status /* [0]=any big, [1]=error */, uint32_t epoch,
                                                      BigItem* __restrict__ biglist, uint32_t* __restrict__ bigmeta,
                                                      uint32_t tpw /* triangles per wave: 64, 32 or 16 (fused_tpw) */,
                                                      RunInfo runs, BatchTable bt) {
    __shared__ F2Lds S;
    F2Ctl& C = S.ctl;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Batches are `tpw` consecutive triangles each — or, for a scene small enough to be converted by ONE generation of
    // workgroups, the entries of a table whose batches carry equal estimated WORK (bt.first[b] .. bt.first[b + 1], at most 64
    // triangles, starts at multiples of 8): with one generation the kernel lasts as long as its slowest workgroup.
    const uint32_t n_batches = bt.first ? bt.n : (sc.n_tri + tpw - 1u) / tpw;
    // Hardware workgroup h runs on XCD h % 8, and every XCD has a private L2.  A launch in RUNS (RunInfo, m2s_device.h) gives XCD x
    // the runs x, x + 8, ... of consecutive units: neighbouring triangles — neighbouring texels — meet in ONE L2 instead of
    // eight, and the look-back chain restarts at every run, whose base comes from a table.  Without runs: plain order, one chain.
    const uint32_t hb = blockIdx.x, xcd = hb & 7u, round = hb >> 3;
    const bool in_runs = runs.base != nullptr;
    const uint32_t rmask = (1u << runs.shift) - 1u;
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    [[maybe_unused]] const unsigned long long tk0 = F2_NOW();     // phase timers: live only in -DM2S_TIMING builds
#ifdef M2S_TIMING
    const unsigned long long tk_real0 = __builtin_amdgcn_s_memrealtime();   // (100 MHz, the same on every XCD: s_memtime is per XCD)
#endif
    [[maybe_unused]] unsigned long long tk_cnt = 0, tk_ent = 0, tk_base = 0, n_strips = 0, tk_tri = 0;
    uint32_t lb = hb;
    if (in_runs) {
        uint32_t rr = ((round >> runs.shift) << 3) + xcd;      // dispatch slot of the run ...
        if (runs.order) rr = ((const __attribute__((address_space(4))) uint32_t*)runs.order)[rr];   // ... heaviest runs first (launch_run_order)
        lb = (rr << runs.shift) + (round & rmask);
    }
    const bool band_first = in_runs && (round & rmask) == 0u;     // first unit of its run: its base is the run's, known
    if (lb * (uint32_t)kTeam >= n_batches) return;
    // control words: each wave initialises its own; the shared ones are set by wave 0 BEFORE it publishes `counted`,
    // and every other wave reads them only after it has seen counted[0] (acquire) — no barrier needed.
    // LDS is not zero on entry: counted[]/expanded[] of OTHER waves may hold garbage until those waves get here.
    // One barrier at the very start (all four waves arrive immediately) makes the flags trustworthy.
    if (lane == 0) { C.counted[wave] = 0; C.expanded[wave] = 0; }
    if (wave == 0 && lane == 0) {
        C.claimed = 0; C.irregular = 0; C.error = 0;
        C.base_state = (lb == 0 || band_first) ? 2u : 0u;
        C.base = band_first ? runs.base[lb >> runs.shift] : 0ull;   // scalar load from device memory
    }
    __syncthreads();
    const uint32_t b0 = lb * kTeam;                    // the workgroup's first batch
    const uint32_t nb_here = min((uint32_t)kTeam, n_batches - b0);
    const uint32_t b = b0 + wave;                      // this wave's batch (may not exist in the last workgroup)
    const bool has_batch = wave < nb_here;
    const unsigned long long band_base = band_first ? C.base : 0ull;     // (of the unit's run)

    // ======================= triangle phase: one batch per wave =======================
    uint32_t t0 = b * tpw, nt = tpw;
    if (bt.first && has_batch) {
        const __attribute__((address_space(4))) uint32_t* q = (const __attribute__((address_space(4))) uint32_t*)bt.first;
        t0 = q[b];
        nt = q[b + 1] - t0;
    }
    const uint32_t t = t0 + lane;
    const bool valid = has_batch && (uint32_t)lane < nt && t < sc.n_tri;
    if (lane == 0) C.t0[wave] = t0;        // (read by other waves only after this wave's `expanded` flag)
    float p[9];
    Geo g;
    Raster rs;
    rs.x0 = rs.y0 = 0; rs.x1 = rs.y1 = -1; rs.ext = 0; rs.bias = 0; rs.area2 = 1;
#pragma unroll
    for (int i = 0; i < 3; i++) { rs.a[i] = rs.b[i] = 0; rs.c[i] = 0; }
    bool ok = false;
    uint32_t m = 0;
    float4 uvb0 = make_float4(0, 0, 0, 0);
    float2 uvb1 = make_float2(0, 0);
    bool uniform_mesh_w = false;   // the wave's batch lies inside one mesh (m0w): mesh uniforms through scalar loads
    uint32_t m0w = 0;
    if (has_batch) {
        const uint32_t lastT = min(t0 + nt, sc.n_tri) - 1;
        bool uniform_mesh;
        const uint32_t m0 = mesh_of_range(sc, t0, lastT, uniform_mesh);   // one scalar load (was: a binary search)
        uniform_mesh_w = uniform_mesh; m0w = m0;
        m = m0;
        if (valid) {
            load_positions(sc.tri, t, p);
            uvb0 = sc.tri.B0[t];
            uvb1 = sc.tri.B1[t];
            // (a wave inside one mesh — the common case — reads the mesh uniforms with scalar loads)
            if (uniform_mesh) geo_setup_mp(p, kConstMesh(sc.meshes + m0), g);
            else { m = find_mesh(sc, sc.tri_first + t); geo_setup_mp(p, sc.meshes + m, g); }
            ok = raster_setup(g, R, rs);
        }
    }
    const int w = rs.x1 - rs.x0 + 1, rows = rs.y1 - rs.y0 + 1;
    int kind = kNone;
    unsigned long long mask = 0;
    uint32_t cnt = 0;
    if (ok) {
        if (w <= 8 && rows <= 8 && rs.ext <= 2304) {
            kind = kSmall;       // (its coverage: below, by the whole wave at once)
        } else if (rows <= kRowsCount) {
            kind = rows <= kFusedRows ? kMedium : kBig;
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            for (int y = rs.y0; y <= rs.y1; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                cnt += (uint32_t)max(xb - xa + 1, 0);
            }
            if (cnt > kBigCount) kind = kBig;
        } else {
            kind = kBig;
        }
    }
    {   // coverage of the small triangles, the wave walking rows and columns together (small_coverage, m2s_devfn.h)
        RasterSmall rsm;
        if (kind == kSmall) {
            // (edge values at the centre of the box-origin pixel: below 2^31 for such boxes, so the low words of raster_setup's 64-bit
            //  constants give them)
            const uint32_t Px0 = 256u * (uint32_t)rs.x0 + 128u, Py0 = 256u * (uint32_t)rs.y0 + 128u;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                rsm.a[i] = rs.a[i]; rsm.b[i] = rs.b[i];
                rsm.e[i] = (int)((uint32_t)rs.a[i] * Px0 + (uint32_t)rs.b[i] * Py0 + (uint32_t)rs.c[i]);
            }
            rsm.bias = rs.bias; rsm.area2 = 0;
        }
        uint32_t mlo = 0, mhi = 0;
        small_coverage(kind == kSmall, w, rows, rsm, mlo, mhi);
        if (kind == kSmall) {
            mask = (unsigned long long)mlo | ((unsigned long long)mhi << 32);
            cnt = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
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
    if (cnt == 0) kind = kNone;
    const uint32_t cntc = (kind == kSmall || kind == kMedium) ? cnt : 0;
    const bool anybig = __ballot(kind == kBig) != 0ull;

    // (a triangle has at most 4096^2 = 2^24 fragments, so the wave's 64 counts sum to < 2^31: 32-bit scans)
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t inclc = wave_incl_scan(cntc, lane);
    const unsigned long long total_w = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t total_c = __builtin_amdgcn_readlane(inclc, 63);
    const unsigned long long toff = incl - cnt;
    const uint32_t ctoff = inclc - cntc;

    // publish: the chain word of this batch (global batch 0 knows its prefix) and the counts for the team
    if (has_batch && lane == 0) {
        // the first batch of the grid / of a band knows its inclusive prefix; everybody else publishes the aggregate
        const bool knows = b == 0 || (band_first && wave == 0);
        chain_store(&chain[b], (knows ? kFlagPrefix : kFlagAgg) | etag | (((knows ? band_base : 0ull) + total_w) & kValMask));
    }
    if (lane == 0) {
        C.total_w[wave] = total_w; C.total_c[wave] = total_c;
        // deferred triangles make record index != base + stream position for everything after them.  The flag travels WITH
        // the counts — the fragment phase starts only after every wave has counted — and not with the later expansion: a
        // strip made of ANOTHER wave's entries does not wait for this wave's expansion and must not read "not yet set"
        // (round 3: it used to be set after the look-back below; a strip of the next wave's entries could run before that)
        if (anybig) __hip_atomic_fetch_or(&C.irregular, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    lds_store(&C.counted[wave], 1u);
    tk_tri = F2_NOW() - tk0;

    // ======================= where do my entries go?  counts of the waves before me =======================
    uint32_t stream0 = 0;              // stream position of my first entry
    unsigned long long out0 = 0;       // fragments (all kinds) of the batches before mine in this workgroup
    bool alive = true;
    {
        const unsigned long long tw0 = F2_NOW();
        for (uint32_t k = 0; k < wave && alive; ++k) {
            uint32_t spins = 0;
            while (lds_load(&C.counted[k]) == 0) {
                if (++spins > kWaitLimit) { alive = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            stream0 += C.total_c[k];
            out0 += C.total_w[k];
        }
        tk_cnt = F2_NOW() - tw0;
    }
    if (!alive && lane == 0) lds_store(&C.error, 1u);
    if (alive && (unsigned long long)stream0 + total_c > kEntries) {   // does not fit the LDS stream
        alive = false;
        if (lane == 0) lds_store(&C.error, 2u);
    }

    float4* stage = S.stage[wave];
    unsigned long long base = 0;
    bool have_base = false;
    // The look-back (one round trip through memory shared by all XCDs, ~5 k cycles) is taken by ONE wave — the LAST one,
    // and BEFORE it expands its own triangles: its entries sit at the end of the workgroup's stream and are the last ones
    // a strip asks for, while the other three waves are already expanding and shading the front of the stream.  By the
    // time anybody reaches a store, the base is there (round 1 took the look-back after the last wave's expansion: wave 0
    // then waited 3.9 k cycles for the base on average).  The predecessors' aggregates are published right after THEIR
    // counting, i.e. at about the time this workgroup has counted too.
    if (alive && wave == (uint32_t)kTeam - 1 && lds_load(&C.error) == 0) {
        const unsigned long long tb0 = F2_NOW();
        have_base = f2_get_base(C, chain, chain, b0, lane, epoch, status, base);
        if (!have_base) alive = false;
        tk_base += F2_NOW() - tb0;
    }

    // ======================= my TriShade, tskip and entries =======================
#if defined(M2S_PROBE_STOP) && M2S_PROBE_STOP >= 2      // instruction-count probes (wrong output): tools/ab/r4_valu.sh
    if (false) {
#else
    if (alive) {
#endif
        if (cntc) {
            TriShade ts;
            if (uniform_mesh_w) tri_shade_setup(p, g, rs, kConstMesh(sc.meshes + m0w), uvb0, uvb1, ts);
            else tri_shade_setup(p, g, rs, sc.meshes + m, uvb0, uvb1, ts);
            ts.mesh |= m;
            const float4* src = reinterpret_cast<const float4*>(&ts);
#pragma unroll
            for (int k = 0; k < 5; ++k) S.tri[wave][lane * 5 + k] = src[k];
            S.tskip[wave][lane] = (uint32_t)((out0 + toff) - ((unsigned long long)stream0 + ctoff));
        }
        if (anybig) {   // deferred triangles: reserve their slice of the output, list them for k_emit_big
            unsigned long long base;
            if (f2_get_base(C, chain, chain, b0, lane, epoch, status, base)) {
                if (kind == kBig) {
                    const uint32_t slot = atomicAdd(&bigmeta[0], 1u);
                    atomicMax(&bigmeta[1], cnt);
                    atomicAdd(&bigmeta[2], cnt);
                    BigItem it2;
                    it2.t = t; it2.cnt = cnt; it2.off = base + out0 + toff;
                    biglist[slot] = it2;
                }
                if (lane == 0) __hip_atomic_store(&status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else alive = false;
        }
        const uint32_t tag = (uint32_t)lane << 24;
        if (kind == kSmall) {
            const uint32_t org = ((uint32_t)rs.y0 << 12) | (uint32_t)rs.x0;
            unsigned long long mm = mask;
            uint32_t ci = stream0 + ctoff;
            while (mm) {
                const int bit = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                S.entries[ci++] = tag + org + (((uint32_t)(bit >> 3) << 12) | (uint32_t)(bit & 7));
            }
        } else if (kind == kMedium) {
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            uint32_t ci = stream0 + ctoff;
            for (int y = rs.y0; y <= rs.y1; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                for (int x = xa; x <= xb; ++x) S.entries[ci++] = tag | ((uint32_t)y << 12) | (uint32_t)x;
            }
        }
    }
    lds_store(&C.expanded[wave], 1u);   // release: TriShade, tskip, entries (set even on error so that nobody waits for it)

    // ======================= fragment phase: strips of the workgroup's stream =======================
    // the stream's length needs every wave's count
    uint32_t stream_total = 0;
    unsigned long long out_total = 0;
    uint32_t cum[kTeam + 1];
    cum[0] = 0;
    {
        const unsigned long long tw0 = F2_NOW();
        for (uint32_t k = 0; k < (uint32_t)kTeam; ++k) {
            uint32_t spins = 0;
            while (lds_load(&C.counted[k]) == 0) {
                if (++spins > kWaitLimit) { alive = false; if (lane == 0) lds_store(&C.error, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
            stream_total += C.total_c[k];
            out_total += C.total_w[k];
            cum[k + 1] = stream_total;
        }
        tk_cnt += F2_NOW() - tw0;
    }
    [[maybe_unused]] const unsigned long long tsl0 = F2_NOW();
#if defined(M2S_PROBE_STOP) && M2S_PROBE_STOP >= 1
    stream_total = 0;
#endif
    while (alive && lds_load(&C.error) == 0) {
        uint32_t s = 0;
        if (lane == 0) s = __hip_atomic_fetch_add(&C.claimed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s = __builtin_amdgcn_readfirstlane(s);
        const uint32_t pos0 = s * 64u;
        if (pos0 >= stream_total) break;
        const uint32_t n = min(64u, stream_total - pos0);
        {   // the waves whose entries this strip contains must have expanded them
            const unsigned long long tw0 = F2_NOW();
            for (uint32_t k = 0; k < (uint32_t)kTeam && alive; ++k) {
                if (cum[k + 1] <= pos0 || cum[k] >= pos0 + n) continue;
                uint32_t spins = 0;
                while (lds_load(&C.expanded[k]) == 0) {
                    if (++spins > kWaitLimit) { alive = false; if (lane == 0) lds_store(&C.error, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            tk_ent += F2_NOW() - tw0;
        }
        if (!alive || lds_load(&C.error)) break;
        ++n_strips;
        uint32_t en = kInvalidEntry;
        if ((uint32_t)lane < n) en = S.entries[pos0 + lane];
        const bool have = en != kInvalidEntry;
        uint32_t ow = 0;                                   // owner wave of my entry: cum[ow] <= pos < cum[ow + 1]
#pragma unroll
        for (int k = 1; k < kTeam; ++k) ow += (pos0 + (uint32_t)lane >= cum[k]) ? 1u : 0u;
        const uint32_t tl = (en >> 24) & 63u;
        float4 rec[6];
        uint32_t skip = 0;
        // do all fragments of the strip belong to one mesh?
        uint32_t my_mesh = 0;
        if (have) my_mesh = reinterpret_cast<const uint32_t*>(&S.tri[ow][tl * 5 + 4])[3] & 0xFFFFFFu;
        const uint32_t m_first = __builtin_amdgcn_readfirstlane(my_mesh);   // lane 0 always holds a fragment
        const bool uniform = sc.n_meshes == 1 || __ballot(have && my_mesh != m_first) == 0ull;
        if (have) {
            const TriShade& ts = *reinterpret_cast<const TriShade*>(&S.tri[ow][tl * 5]);
            const uint32_t tt = C.t0[ow] + tl;
            skip = S.tskip[ow][tl];
            // a strip inside one mesh (the common case): wave-uniform descriptor pointer in the constant address space
            const float2* uvl = nullptr;
            if (uniform) shade_from_tri(sc.tri, tt, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), kConstMesh(sc.meshes + m_first), ts, rec, nullptr, uvl);
            else shade_from_tri(sc.tri, tt, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), sc.meshes + my_mesh, ts, rec, nullptr, uvl);
        }
        if (!have_base) {
            const unsigned long long tb0 = F2_NOW();
            if (!f2_get_base(C, chain, chain, b0, lane, epoch, status, base)) break;
            have_base = true;
            tk_base += F2_NOW() - tb0;
        }
        if (lds_load(&C.irregular) == 0) {
            const unsigned long long o0 = base + pos0;
            uint32_t nvalid = n;
            if (o0 + 64ull > limit) {
                if (o0 >= limit) nvalid = 0;
                else if (limit - o0 < nvalid) nvalid = (uint32_t)(limit - o0);
            }
#pragma unroll 1
            for (int part = 0; part < 64 / kStageRec; ++part) {
                if (have && (lane / kStageRec) == part) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) stage[(lane % kStageRec) * 6 + k] = rec[k];
                }
                wave_lds_sync();
                float4* __restrict__ dsto = out + (o0 + (uint32_t)kStageRec * part) * 6;
                const uint32_t nv = nvalid > (uint32_t)kStageRec * part ? min((uint32_t)kStageRec, nvalid - (uint32_t)kStageRec * part) : 0u;
#pragma unroll
                for (int j = 0; j < (kStageRec * 6 + 63) / 64; ++j) {
                    const uint32_t q = (uint32_t)lane + 64u * j;
                    const uint32_t r = q / 6u;
                    if (r < nv) nt_store(&dsto[q], stage[q]);
                }
                wave_lds_sync();
            }
        } else if (have) {
            const unsigned long long oidx = base + skip + pos0 + lane;
            if (oidx < limit) {
                float4* __restrict__ dsto = out + oidx * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) nt_store(&dsto[k], rec[k]);
            }
        }
    }
    [[maybe_unused]] const unsigned long long tsl1 = F2_NOW();
    // ======================= epilogue: the workgroup's inclusive prefix / the counter =======================
    // (by the wave of the last batch; the base is resolved here if no strip needed it, e.g. a workgroup without fragments)
    if (alive && wave == nb_here - 1 && lds_load(&C.error) == 0) {
        if (!have_base) have_base = f2_get_base(C, chain, chain, b0, lane, epoch, status, base);
        if (have_base && lane == 0) {
            chain_store(&chain[b0 + nb_here - 1], kFlagPrefix | etag | ((base + out_total) & kValMask));
            if (b0 + nb_here == n_batches) *total_out = base + out_total;
            // by-product of a launch without runs: where every run's output starts (the table of the NEXT launches at this R)
            if (runs.out && (lb & rmask) == 0u) runs.out[lb >> runs.shift] = base;
        }
    }
    // status[1] != 0 is what the host acts on; 2 = "a workgroup's entries do not fit", 1 = a bounded wait gave up
    { const uint32_t e = lds_load(&C.error); if (e && lane == 0) __hip_atomic_store(&status[1], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    F2_T(0, F2_NOW() - tk0); F2_T(1, tk_cnt); F2_T(2, tk_ent); F2_T(3, tk_base); F2_T(4, n_strips); F2_T(5, (unsigned long long)stream_total);
    F2_T(6, tk_real0); F2_T(7, __builtin_amdgcn_s_memrealtime()); F2_T(8, (unsigned long long)(blockIdx.x & 7u));
    F2_T(11, tk_tri);
    F2_T(13, tsl1 - tsl0); F2_TW(3, 14, tsl1 - tsl0); F2_TW(3, 15, n_strips);
}

void launch_fused2(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                   unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta,
                   const RunInfo& runs, const BatchTable& bt, hipStream_t st) {
    // small scenes: fewer triangles per wave, more workgroups (see m2s_device.h); bt.tpw: the caller's batch size (scenes of 11-18
    // fragments per triangle: smaller batches keep a workgroup's entries inside its LDS stream)
    const uint32_t tpw = (!bt.first && bt.tpw) ? bt.tpw : fused_tpw(sc.n_tri);
    const uint32_t n_batches = bt.first ? bt.n : (sc.n_tri + tpw - 1u) / tpw;
    if (!n_batches) return;
    uint32_t nb = (n_batches + kTeam - 1) / kTeam;
    RunInfo r = runs;
    if (tpw != 64u || kTeam != 4 || bt.first) r = RunInfo{ nullptr, nullptr, 0u, nullptr };
    if (r.base) r.out = nullptr;
    if (r.base) nb = ((nb + (8u << r.shift) - 1u) / (8u << r.shift)) * (8u << r.shift);   // whole groups of eight runs; surplus workgroups exit at once
    else nb = (nb + 7u) & ~7u;
    hipLaunchKernelGGL(k_fused2, dim3(nb), dim3(kTeamThreads), 0, st, sc, R, chain, (unsigned long long)limit, out, total, status,
                       epoch & 0xFFFFu, biglist, bigmeta, tpw, r, bt);
}

// units of k_fused2 for a scene that can be converted in runs (64-triangle batches, no batch table); 0: no runs
uint32_t fused2_band_workgroups(uint32_t n_tri) {
    if (fused_tpw(n_tri) != 64u || kTeam != 4) return 0;
    return (n_fused_waves(n_tri) + 3u) / 4u;
}

#ifdef M2S_TIMING
extern "C" int m2s_debug_read_timing2(unsigned long long* dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_f2_timing), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_fused2() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fused2)); }

}  // namespace m2s
