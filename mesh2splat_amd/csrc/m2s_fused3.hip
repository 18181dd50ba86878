// m2s_fused3.hip — single-pass conversion kernel, the LEAN form of the team kernel (gfx950; round 5).
//
// k_fused2 (m2s_fused2.hip) is compiled for every scene the library accepts: triangles of up to 16 pixel rows expanded in the
// workgroup (a 30-register row walker), edge functions in 64 bits, three separately sized maps sampled in one round trip (75
// registers of taps and texels).  That generality is what its 157 vector registers — three waves per SIMD — pay for, and a
// scene like BASELINE config 3 uses none of it: every triangle fits an 8 x 8 pixel box, every material samples one interleaved
// ("combo") texture.  This file is the same protocol — one batch of triangles per wave, ONE entry stream per workgroup, strips
// of 64 fragments handed out by an LDS counter, one look-back per workgroup, records staged in LDS and written as coalesced
// non-temporal runs; the same bytes out — compiled for exactly that case:
//
//   * in the workgroup only triangles whose pixel box is at most 8 x 8 (coverage = one 64-bit mask); everything larger is
//     counted here (wave-cooperatively, one row per lane) and handed to k_emit_big through the deferred-triangle list, as
//     k_fused2 does with triangles of more than 16 rows.  The host uses this kernel only while such triangles are rare
//     (run_pass: a launch that deferred many switches the scene back to k_fused2 at that R);
//   * fragment constants as TriShadeS (64 B: 16-bit edge coefficients, 32-bit edge values — exact for such boxes), barycentrics
//     in 32-bit integers; entries are 16 bits (lane << 6 | bit of the mask);
//   * only meshes that sample their combo texture or no map at all (decided at upload: SceneDev has no other mesh);
//   * fragment constants are computed and stored BEFORE the wave waits for anything: across the waits a lane keeps its mask,
//     its counts and its triangle index, nothing else; the strip loop re-reads the workgroup's protocol words from LDS.
//
// kLeanWaves waves per SIMD (M2S_FUSED3_WAVES, default 4: 128 registers, 40 KB of LDS per workgroup).
#include "m2s_fused_common.h"

#pragma clang fp contract(off)

namespace m2s {

#ifndef M2S_FUSED3_WAVES
#define M2S_FUSED3_WAVES 4                 // waves per SIMD the kernel is compiled for (= workgroups per CU)
#endif
#ifndef M2S_FUSED3_ENTRIES
#define M2S_FUSED3_ENTRIES 1024            // entry-stream capacity per wave of the team (x2 bytes x 4 of LDS)
#endif
#ifndef M2S_FUSED3_STAGE
#define M2S_FUSED3_STAGE (M2S_FUSED3_WAVES >= 5 ? 16 : 32)   // records staged per wave and round (32 = half a strip)
#endif
constexpr int kL3Team = 4;                 // waves (= batches) per workgroup
constexpr int kL3Threads = kL3Team * 64;
constexpr uint32_t kL3Entries = (uint32_t)M2S_FUSED3_ENTRIES * kL3Team;
constexpr int kL3Stage = M2S_FUSED3_STAGE;
constexpr uint32_t kL3Wait = 1u << 24;     // LDS polls before giving up
constexpr uint32_t kL3Stop = 1u << 28;     // added to the strip counter by a wave that gives up: nobody draws a strip after that

struct F3Ctl {
    unsigned long long base;               // record index of stream position 0
    unsigned long long total_w[kL3Team];   // fragments (all kinds) per batch
    uint32_t total_c[kL3Team];             // entries per batch
    uint32_t counted[kL3Team];             // 1: total_w / total_c of that wave are valid
    uint32_t expanded[kL3Team];            // 1: that wave's TriShadeS, tskip and entries are in place
    uint32_t t0[kL3Team];                  // first triangle of each wave's batch
    uint32_t claimed;                      // next strip to hand out
    uint32_t base_state;                   // 0 unknown, 1 being resolved, 2 known
    uint32_t flags;                        // bits 0-7: error (1 a wait gave up, 2 entries do not fit); bit 8: deferred triangles
                                           // somewhere in this workgroup (record index != base + stream position after them)
    uint32_t pad_;
};
struct F3Lds {
    float4 tri[kL3Team][64 * 4];           // TriShadeS of the four batches
    uint32_t tskip[kL3Team][64];           // per triangle: (record index - stream position) of its fragments
    uint16_t entries[kL3Entries];          // lane << 6 | bit of the 8 x 8 mask  (the owning wave follows from the stream position)
    float4 stage[kL3Team][kL3Stage * 6];   // record staging, one per wave
    F3Ctl ctl;
};
static_assert(sizeof(F3Lds) * M2S_FUSED3_WAVES <= 163840, "M2S_FUSED3_WAVES workgroups per CU must fit the 160 KB of LDS");
constexpr uint32_t kF3Irregular = 1u << 8;

__device__ __forceinline__ uint32_t l3_load(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void l3_store(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// a wave-uniform word, re-read where it is used (relaxed atomic: not hoisted out of the strip loop), into a scalar register
__device__ __forceinline__ uint32_t l3_uniform(const uint32_t* p) {
    return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
// a wave gives up: remember why, and stop the strip counter
__device__ __forceinline__ void l3_fail(F3Ctl& C, uint32_t why, int lane) {
    if (lane == 0) {
        __hip_atomic_fetch_or(&C.flags, why, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&C.claimed, kL3Stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// The workgroup's base: sum of the totals of all batches before its first one.  Whoever needs it first resolves it.
__device__ __forceinline__ bool f3_get_base(F3Ctl& C, unsigned long long* chain, uint32_t b0, int lane, uint32_t epoch, uint32_t* status,
                                            unsigned long long& base) {
    uint32_t st = l3_load(&C.base_state);
    if (st != 2) {
        uint32_t got = 1;
        if (lane == 0) {
            uint32_t expect = 0;
            got = __hip_atomic_compare_exchange_strong(&C.base_state, &expect, 1u, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) ? 0u : 1u;
        }
        got = __builtin_amdgcn_readfirstlane(got);
        if (got == 0) {   // this wave resolves
            const unsigned long long b = b0 == 0 ? 0ull : lookback(chain, b0, lane, epoch, status);
            if (lane == 0) {
                C.base = b;
                // the first batch's inclusive prefix: successors' look-backs stop here
                chain_store(&chain[b0], kFlagPrefix | ((unsigned long long)epoch << kEpochShift) | ((b + C.total_w[0]) & kValMask));
            }
            l3_store(&C.base_state, 2u);
        } else {
            uint32_t spins = 0;
            while (l3_load(&C.base_state) != 2) {
                if (++spins > kL3Wait) { l3_fail(C, 1u, lane); return false; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
    }
    base = C.base;
    return true;
}

#ifdef M2S_TIMELINE
// measurement build only (tools/timeline_probe.py): 100 MHz timestamps per wave — start, counts known, entries expanded, first strip,
// last strip done, end — and the strips the wave shaded
__device__ unsigned long long g_tl_f3[16384 * 4 * 8];
#define TLF(k) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0 && blockIdx.x < 16384u) g_tl_f3[((size_t)blockIdx.x * 4 + wave) * 8 + (k)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define TLFV(k, v) do { if (lane == 0 && blockIdx.x < 16384u) g_tl_f3[((size_t)blockIdx.x * 4 + wave) * 8 + (k)] = (v); } while (0)
#else
#define TLF(k) do { } while (0)
#define TLFV(k, v) do { } while (0)
#endif

__global__ void __launch_bounds__(kL3Threads, M2S_FUSED3_WAVES) k_fused3(SceneDev sc, uint32_t R, unsigned long long* __restrict__ chain,
                                                      unsigned long long limit, float4* __restrict__ out,
                                                      unsigned long long* __restrict__ total_out,
                                                      uint32_t* __restrict__ status /* [0]=any big, [1]=error */, uint32_t epoch,
                                                      BigItem* __restrict__ biglist, uint32_t* __restrict__ bigmeta,
                                                      uint32_t tpw /* triangles per wave: 64 .. 8 (fused_tpw) */,
                                                      RunInfo runs, BatchTable bt) {
    __shared__ F3Lds S;
    F3Ctl& C = S.ctl;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // batches, units, XCD runs: exactly as in k_fused2 (the two kernels share the run tables of a scene)
    const uint32_t n_batches = bt.first ? bt.n : (sc.n_tri + tpw - 1u) / tpw;
    const uint32_t hb = blockIdx.x, xcd = hb & 7u, round = hb >> 3;
    const bool in_runs = runs.base != nullptr;
    const uint32_t rmask = (1u << runs.shift) - 1u;
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    uint32_t lb = hb;
    if (in_runs) {
        uint32_t rr = ((round >> runs.shift) << 3) + xcd;          // dispatch slot of the run ...
        if (runs.order) rr = ((const __attribute__((address_space(4))) uint32_t*)runs.order)[rr];   // ... heaviest runs first
        lb = (rr << runs.shift) + (round & rmask);
    }
    const bool band_first = in_runs && (round & rmask) == 0u;     // first unit of its run: its base is the run's, known
    if (lb * (uint32_t)kL3Team >= n_batches) return;
    TLF(0);
    // LDS is not zero on entry: one barrier at the very start makes the flags trustworthy (see k_fused2)
    if (lane == 0) { C.counted[wave] = 0; C.expanded[wave] = 0; }
    if (wave == 0 && lane == 0) {
        C.claimed = 0; C.flags = 0;
        C.base_state = (lb == 0 || band_first) ? 2u : 0u;
        C.base = band_first ? runs.base[lb >> runs.shift] : 0ull;
    }
    __syncthreads();
    const uint32_t b0 = lb * kL3Team;
    const uint32_t nb_here = min((uint32_t)kL3Team, n_batches - b0);
    const uint32_t b = b0 + wave;
    const bool has_batch = wave < nb_here;
    const unsigned long long band_base = band_first ? C.base : 0ull;

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
    int kind = kNone;
    unsigned long long mask = 0;
    uint32_t cnt = 0;
    {
        float p[9];
        Geo g;
        RasterHead h;
        h.x0 = h.y0 = 0; h.x1 = h.y1 = -1; h.ext = 0;
        bool box = false;
        uint32_t m = 0;
        float4 uvb0 = make_float4(0, 0, 0, 0);
        float2 uvb1 = make_float2(0, 0);
        bool uniform_mesh = false;     // the wave's batch lies inside one mesh (m0): mesh uniforms through scalar loads
        uint32_t m0 = 0;
        if (has_batch) {
            const uint32_t lastT = min(t0 + nt, sc.n_tri) - 1;
            m0 = mesh_of_range(sc, t0, lastT, uniform_mesh);
            m = m0;
            if (valid) {
                load_positions(sc.tri, t, p);
                uvb0 = sc.tri.B0[t];
                uvb1 = sc.tri.B1[t];
                if (uniform_mesh) geo_setup_mp(p, kConstMesh(sc.meshes + m0), g);
                else { m = find_mesh(sc, sc.tri_first + t); geo_setup_mp(p, sc.meshes + m, g); }
                box = raster_head(g, R, h);
            }
        }
        const int w = h.x1 - h.x0 + 1, rows = h.y1 - h.y0 + 1;
        // small: shaded in this workgroup, all of its integer setup in 32 bits (raster_small)
        RasterSmall rs;
        bool small = box && w <= 8 && rows <= 8 && h.ext <= 2304;
        if (small) small = raster_small(h, rs);
        else if (box) kind = kBig;         // (unless it has no area: decided where the wave counts it, below)
        {   // coverage of the small triangles, the wave walking rows and columns together (small_coverage, m2s_devfn.h)
            uint32_t mlo = 0, mhi = 0;
            small_coverage(small, w, rows, rs, mlo, mhi);
            mask = (unsigned long long)mlo | ((unsigned long long)mhi << 32);
            cnt = (uint32_t)__popc(mlo) + (uint32_t)__popc(mhi);
            if (small) kind = kSmall;
        }
        // fragment constants of the triangles this workgroup shades itself: in place before any wait (see the file header)
        if (cnt != 0 && kind == kSmall) {
            TriShadeS c;
            if (uniform_mesh) tri_shade_small(p, g, h, rs, kConstMesh(sc.meshes + m0), uvb0, uvb1, m, c);
            else tri_shade_small(p, g, h, rs, sc.meshes + m, uvb0, uvb1, m, c);
            const float4* src = reinterpret_cast<const float4*>(&c);
#pragma unroll
            for (int k = 0; k < 4; ++k) S.tri[wave][lane * 4 + k] = src[k];
        }
        if (__ballot(kind == kBig) != 0ull) {   // larger triangles (rare): the whole wave counts one of them, a pixel row per lane
            // (inline on purpose: as a non-inlined function this path gave the kernel a stack — 64 bytes of scratch per lane — and the
            //  FIRST launch of a kernel that uses scratch pays the runtime's scratch allocation: first conversion 0.19 -> 0.31 ms)
            Raster rb;                           // their full (64-bit) setup, only here
            rb.x0 = rb.y0 = 0; rb.x1 = rb.y1 = -1; rb.ext = 0; rb.bias = 0; rb.area2 = 1;
#pragma unroll
            for (int i = 0; i < 3; i++) { rb.a[i] = rb.b[i] = 0; rb.c[i] = 0; }
            if (kind == kBig && !raster_setup(g, R, rb)) kind = kNone;
            unsigned long long bigm = __ballot(kind == kBig);
            while (bigm) {
                const int src = __ffsll((long long)bigm) - 1;
                bigm &= bigm - 1;
                const Raster br = shfl_raster(rb, src);
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
    }
    if (cnt == 0) kind = kNone;
    const uint32_t cntc = kind == kSmall ? cnt : 0;
    const bool anybig = __ballot(kind == kBig) != 0ull;

    // (a triangle has at most 4096^2 = 2^24 fragments, so the wave's 64 counts sum to < 2^31: 32-bit scans)
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t inclc = wave_incl_scan(cntc, lane);
    const unsigned long long total_w = __builtin_amdgcn_readlane(incl, 63);
    const uint32_t total_c = __builtin_amdgcn_readlane(inclc, 63);
    const uint32_t toff = incl - cnt;
    const uint32_t ctoff = inclc - cntc;

    // publish: the chain word of this batch (global batch 0 and the first batch of a run know their prefix) and the counts
    if (has_batch && lane == 0) {
        const bool knows = b == 0 || (band_first && wave == 0);
        chain_store(&chain[b], (knows ? kFlagPrefix : kFlagAgg) | etag | (((knows ? band_base : 0ull) + total_w) & kValMask));
    }
    if (lane == 0) {
        C.total_w[wave] = total_w; C.total_c[wave] = total_c;
        // the "deferred triangles" flag travels WITH the counts (see k_fused2: a strip of another wave's entries must see it)
        if (anybig) __hip_atomic_fetch_or(&C.flags, kF3Irregular, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    l3_store(&C.counted[wave], 1u);
    TLF(1);

    // ======================= where do my entries go?  counts of the waves before me =======================
    uint32_t stream0 = 0;              // stream position of my first entry
    unsigned long long out0 = 0;       // fragments (all kinds) of the batches before mine in this workgroup
    bool alive = true;
    for (uint32_t k = 0; k < wave && alive; ++k) {
        uint32_t spins = 0;
        while (l3_load(&C.counted[k]) == 0) {
            if (++spins > kL3Wait) { alive = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        stream0 += C.total_c[k];
        out0 += C.total_w[k];
    }
    if (!alive) l3_fail(C, 1u, lane);
    if (alive && (unsigned long long)stream0 + total_c > kL3Entries) {   // does not fit the LDS stream
        alive = false;
        l3_fail(C, 2u, lane);
    }
    // the look-back is taken by the LAST wave, before its own expansion: its entries are the last ones a strip asks for
    unsigned long long base = 0;
    bool have_base = false;
    if (alive && wave == (uint32_t)kL3Team - 1 && (l3_load(&C.flags) & 0xFFu) == 0) {
        have_base = f3_get_base(C, chain, b0, lane, epoch, status, base);
        if (!have_base) alive = false;
    }

    // ======================= my tskip and entries =======================
    if (alive) {
        if (cntc) S.tskip[wave][lane] = (uint32_t)((out0 + toff) - ((unsigned long long)stream0 + ctoff));
        if (anybig) {   // deferred triangles: reserve their slice of the output, list them for k_emit_big
            unsigned long long bb;
            if (f3_get_base(C, chain, b0, lane, epoch, status, bb)) {
                if (kind == kBig) {
                    const uint32_t slot = atomicAdd(&bigmeta[0], 1u);
                    atomicMax(&bigmeta[1], cnt);
                    atomicAdd(&bigmeta[2], cnt);
                    BigItem it2;
                    it2.t = t; it2.cnt = cnt; it2.off = bb + out0 + toff;
                    biglist[slot] = it2;
                }
                if (lane == 0) __hip_atomic_store(&status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else alive = false;
        }
        if (kind == kSmall) {
            unsigned long long mm = mask;
            uint32_t ci = stream0 + ctoff;
            const uint32_t tag = (uint32_t)lane << 6;
            while (mm) {
                const int bit = __ffsll((long long)mm) - 1;
                mm &= mm - 1;
                S.entries[ci++] = (uint16_t)(tag | (uint32_t)bit);
            }
        }
    }
    l3_store(&C.expanded[wave], 1u);   // release: TriShadeS, tskip, entries (set even on error so that nobody waits for it)
    TLF(2);
    [[maybe_unused]] unsigned long long tl_strips = 0;

    // ======================= fragment phase: strips of the workgroup's stream =======================
    // the stream's length needs every wave's count
    for (uint32_t k = 0; k < (uint32_t)kL3Team && alive; ++k) {
        uint32_t spins = 0;
        while (l3_load(&C.counted[k]) == 0) {
            if (++spins > kL3Wait) { alive = false; l3_fail(C, 1u, lane); break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    float4* const stage = S.stage[wave];
    uint32_t exp_seen = 0;             // bit k: wave k's expansion has been seen (after that its flag is not polled again)
    TLF(3);
    while (alive) {
        uint32_t s = 0;
        if (lane == 0) s = __hip_atomic_fetch_add(&C.claimed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        s = __builtin_amdgcn_readfirstlane(s);
        // the workgroup's protocol words are re-read from LDS for every strip (wave-uniform: they live in scalar registers for the
        // length of one strip instead of in vector registers for the whole loop)
        const uint32_t c1 = l3_uniform(&C.total_c[0]), c2 = c1 + l3_uniform(&C.total_c[1]), c3 = c2 + l3_uniform(&C.total_c[2]),
                       stream_total = c3 + l3_uniform(&C.total_c[3]);
        const uint32_t pos0 = s * 64u;
        if (s >= kL3Stop / 64u || pos0 >= stream_total) break;
#ifdef M2S_TIMELINE
        ++tl_strips;
#endif
        const uint32_t n = min(64u, stream_total - pos0);
        if (exp_seen != 15u) {   // the waves whose entries this strip contains must have expanded them
            const uint32_t lo[4] = { 0u, c1, c2, c3 }, hi[4] = { c1, c2, c3, stream_total };
#pragma unroll
            for (int k = 0; k < kL3Team; ++k) {
                if ((exp_seen >> k) & 1u) continue;
                if (hi[k] <= pos0 || lo[k] >= pos0 + n) continue;
                uint32_t spins = 0;
                while (l3_load(&C.expanded[k]) == 0) {
                    if (++spins > kL3Wait) { alive = false; l3_fail(C, 1u, lane); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                exp_seen |= 1u << k;
            }
            if (!alive) break;
        }
#ifdef M2S_FUSED3_PRIO
        __builtin_amdgcn_s_setprio(M2S_FUSED3_PRIO);       // A/B: the front of a strip (up to its texel requests) ahead of other waves' arithmetic
#endif
        const uint32_t pos = pos0 + (uint32_t)lane;
        const bool have = (uint32_t)lane < n;
        uint32_t en = 0;
        if (have) en = S.entries[pos];
        const uint32_t ow = (pos >= c1 ? 1u : 0u) + (pos >= c2 ? 1u : 0u) + (pos >= c3 ? 1u : 0u);   // owner wave of my entry
        const uint32_t tl = (en >> 6) & 63u, bit = en & 63u;
        float4 rec[6];
        // fragments of one strip almost always belong to one mesh; if not, the meshes take turns (every turn shades through a
        // wave-uniform descriptor: scalar loads, one instantiation of the shader)
        uint32_t my_mesh = 0;
        if (have) my_mesh = reinterpret_cast<const uint32_t*>(&S.tri[ow][tl * 4 + 3])[3] & 0xFFFFFFu;
        unsigned long long todo = __ballot(have);
        while (todo) {
            const uint32_t m_now = __builtin_amdgcn_readlane(my_mesh, __ffsll((long long)todo) - 1);
            const bool mine = have && my_mesh == m_now;
            if (mine) {
                const TriShadeS& ts = *reinterpret_cast<const TriShadeS*>(&S.tri[ow][tl * 4]);
                const uint32_t tt = C.t0[ow] + tl;
                const uint32_t org = ts.org;
                const int x = (int)(org & 0xFFFu) + (int)(bit & 7u), y = (int)(org >> 12) + (int)(bit >> 3);
                // (readfirstlane, not m_now: inside this branch the optimiser knows my_mesh == m_now and substitutes the per-lane
                //  value — the descriptor loads then become vector loads through a per-lane pointer, the texel reads 64-bit addresses)
                shade_from_tri<ConstMeshPtr, TriShadeS, true>(sc.tri, tt, x, y, kConstMesh(sc.meshes + __builtin_amdgcn_readfirstlane(my_mesh)), ts, rec);
            }
            todo &= ~__ballot(mine);
        }
        if (!have_base) {
            if (!f3_get_base(C, chain, b0, lane, epoch, status, base)) break;
            have_base = true;
        }
        const uint32_t fl = l3_load(&C.flags);
        if (fl & 0xFFu) break;
        // (the base, like the counts, is re-read per strip: two scalar registers for the length of the stores)
        const unsigned long long base_now = (unsigned long long)l3_uniform(reinterpret_cast<const uint32_t*>(&C.base)) |
                                            ((unsigned long long)l3_uniform(reinterpret_cast<const uint32_t*>(&C.base) + 1) << 32);
        if ((fl & kF3Irregular) == 0) {
            const unsigned long long o0 = base_now + pos0;
            uint32_t nvalid = n;
            if (o0 + 64ull > limit) {
                if (o0 >= limit) nvalid = 0;
                else if (limit - o0 < nvalid) nvalid = (uint32_t)(limit - o0);
            }
#pragma unroll 1
            for (int part = 0; part < 64 / kL3Stage; ++part) {
                if (have && (lane / kL3Stage) == part) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) stage[(lane % kL3Stage) * 6 + k] = rec[k];
                }
                wave_lds_sync();
                float4* __restrict__ dsto = out + (o0 + (uint32_t)kL3Stage * part) * 6;
                const uint32_t nv = nvalid > (uint32_t)kL3Stage * part ? min((uint32_t)kL3Stage, nvalid - (uint32_t)kL3Stage * part) : 0u;
#pragma unroll
                for (int j = 0; j < (kL3Stage * 6 + 63) / 64; ++j) {
                    const uint32_t q = (uint32_t)lane + 64u * j;
                    const uint32_t r = q / 6u;
                    if (r < nv) nt_store(&dsto[q], stage[q]);
                }
                wave_lds_sync();
            }
        } else if (have) {
            const unsigned long long oidx = base_now + S.tskip[ow][tl] + pos;
            if (oidx < limit) {
                float4* __restrict__ dsto = out + oidx * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) nt_store(&dsto[k], rec[k]);
            }
        }
    }
    TLF(4);
    TLFV(6, tl_strips);
    // ======================= epilogue: the workgroup's inclusive prefix / the counter =======================
    // (by the wave of the last batch; the base is resolved here if no strip needed it, e.g. a workgroup without fragments)
    const uint32_t err = l3_load(&C.flags) & 0xFFu;
    if (alive && wave == nb_here - 1 && err == 0) {
        if (!have_base) have_base = f3_get_base(C, chain, b0, lane, epoch, status, base);
        if (have_base && lane == 0) {
            const unsigned long long out_total = (C.total_w[0] + C.total_w[1]) + (C.total_w[2] + C.total_w[3]);
            chain_store(&chain[b0 + nb_here - 1], kFlagPrefix | etag | ((base + out_total) & kValMask));
            if (b0 + nb_here == n_batches) *total_out = base + out_total;
            // by-product of a launch without runs: where every run's output starts (the table of the NEXT launches at this R)
            if (runs.out && (lb & rmask) == 0u) runs.out[lb >> runs.shift] = base;
        }
    }
    // status[1] != 0 is what the host acts on; 2 = "a workgroup's entries do not fit", 1 = a bounded wait gave up
    if (err && lane == 0) __hip_atomic_store(&status[1], err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    TLF(5);
    TLFV(7, (unsigned long long)lb);
}

void launch_fused3(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                   unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta,
                   const RunInfo& runs, const BatchTable& bt, hipStream_t st) {
    const uint32_t tpw = fused_tpw(sc.n_tri);
    const uint32_t n_batches = bt.first ? bt.n : n_fused_waves(sc.n_tri);
    if (!n_batches) return;
    uint32_t nb = (n_batches + kL3Team - 1) / kL3Team;
    RunInfo r = runs;
    if (tpw != 64u || bt.first) r = RunInfo{ nullptr, nullptr, 0u, nullptr };
    if (r.base) r.out = nullptr;
    if (r.base) nb = ((nb + (8u << r.shift) - 1u) / (8u << r.shift)) * (8u << r.shift);   // whole groups of eight runs; surplus workgroups exit at once
    else nb = (nb + 7u) & ~7u;
    hipLaunchKernelGGL(k_fused3, dim3(nb), dim3(kL3Threads), 0, st, sc, R, chain, (unsigned long long)limit, out, total, status,
                       epoch & 0xFFFFu, biglist, bigmeta, tpw, r, bt);
}

#ifdef M2S_TIMELINE
}  // namespace m2s
extern "C" int m2s_debug_timeline_f3(void* dst, size_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(m2s::g_tl_f3), std::min(bytes, sizeof(m2s::g_tl_f3))) == hipSuccess ? 0 : 1;
}
extern "C" int m2s_debug_timeline_f3_clear() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(m2s::g_tl_f3)) != hipSuccess) return 1;
    return hipMemset(p, 0, sizeof(m2s::g_tl_f3)) == hipSuccess ? 0 : 2;
}
namespace m2s {
#endif
hipError_t preload_fused3() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fused3)); }

}  // namespace m2s
