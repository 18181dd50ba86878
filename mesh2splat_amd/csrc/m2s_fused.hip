// m2s_fused.hip — single-pass conversion kernel, one-wave-per-batch form (gfx950).  (m2s_fused2.hip is the
// workgroup-cooperative form of the same kernel and the default; this one has no per-workgroup capacity limit and
// takes over when a workgroup's fragments do not fit there.)
//
// One launch does what the reference's VS+GS+rasteriser+FS draw does (converterGS.glsl:326-443,
// converterFS.glsl:44-104, ConversionPass.cpp:114-116) for triangles whose fragments fit the in-wave budget.
// Every WAVE owns 64 (32 / 16 for small scenes) consecutive triangles from load to store:
//
//   triangle phase  (1 lane / triangle)
//       coalesced 36 B position load -> GS setup -> exact coverage:
//         * bbox <= 8x8 px : 64-bit coverage mask from incremental int32 edge functions
//         * <= 16 rows, <= 96 fragments : row spans from the division-free row walker
//         * larger         : counted (row walker up to 128 rows, else wave-cooperatively), emission deferred
//       per-triangle fragment constants (edge functions, 1/area, Scale, Quaternion, LODs) -> LDS
//   ordering        wave scan of the counts + decoupled look-back over a chain of 64-bit {flag, epoch, value}
//                   words: the wave learns the index of its first record in the global, canonically ordered
//                   output WITHOUT a second pass and without atomics on a shared cursor (the reference: one
//                   atomicCounterIncrement per fragment, converterFS.glsl:46).
//   fragment phase  (1 lane / Gaussian) masks/spans -> LDS entry list, shading from the LDS triangle record,
//                   records staged half a wave at a time in LDS and written as contiguous 3 KiB runs with
//                   16 B/lane non-temporal stores.
//
// Inter-wave protocol (MI355X_MICROARCH.md "R2 granule"): each chain word is written by ONE relaxed agent-scope
// 8-byte atomic store and read by relaxed agent-scope 8-byte atomic loads; the word carries both the flag and the
// payload, so no fence is needed.  A wave only ever waits for waves of workgroups dispatched before its own
// (hardware dispatches a grid in increasing workgroup order); every spin is bounded and sets an error flag instead
// of hanging.
#include "m2s_fused_common.h"

#pragma clang fp contract(off)

namespace m2s {

#ifdef M2S_TIMING
// debug build only: per-wave phase timestamps (s_memtime), read back by tools/fused_timing.py
constexpr int kTimingSlots = 16, kTimingWaves = 16384;
__device__ unsigned long long g_timing[kTimingSlots * kTimingWaves];
#define M2S_STAMP(i) do { if (lane == 0 && wid < kTimingWaves) g_timing[(i) * kTimingWaves + wid] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define M2S_STAMP(i) do {} while (0)
#endif

#ifndef M2S_XCD_RUN
#define M2S_XCD_RUN 1
#endif
constexpr uint32_t kXcdRun = M2S_XCD_RUN;  // logical workgroups per XCD run (1 = plain round-robin)
constexpr int kWaveEntryCap = 192;  // per-wave LDS entry-list window (fragments)

// Everything one wave needs in LDS: exactly 10 KiB -> 40 KiB per 256-thread workgroup, 4 workgroups per CU.
struct WaveLds {
    float4 tri[64 * 5];                 // TriShade of the wave's 64 triangles
    uint32_t tskip[64];                 // fragments of deferred (big) triangles preceding each triangle
    uint4 park[64];                     // per-triangle expansion state {mask.lo, mask.hi, ctoff | cnt<<18 | kind<<30, origin
                                        // pixel}: parked here so it is not live in VGPRs during the fragment loop
    uint32_t entries[kWaveEntryCap];    // slot<<24 | y<<12 | x
    float4 stage[32 * 6];               // half-wave record staging
};

// Each WAVE owns 64 consecutive triangles from load to store; waves never synchronise with each
// other inside the workgroup (no __syncthreads), so a stalled wave never holds back its neighbours.
#ifndef M2S_FUSED_WAVES
#define M2S_FUSED_WAVES 3
#endif
__global__ void __launch_bounds__(kBlock, M2S_FUSED_WAVES) k_fused(SceneDev sc, uint32_t R, unsigned long long* __restrict__ chain,
                                                  unsigned long long limit, float4* __restrict__ out,
                                                  unsigned long long* __restrict__ total_out,
                                                  uint32_t* __restrict__ status /* [0]=any big, [1]=error */, uint32_t epoch,
                                                  BigItem* __restrict__ biglist, uint32_t* __restrict__ bigmeta,
                                                  uint32_t tpw /* triangles per wave: 64, 32 or 16 (fused_tpw) */) {
    __shared__ WaveLds lds_all[kBlock / 64];
    const int lane = threadIdx.x & 63;
    WaveLds& L = lds_all[threadIdx.x >> 6];
    // Optional XCD-aware placement (kXcdRun > 1): hardware workgroup b runs on XCD b % 8 (private L2 each);
    // runs of kXcdRun consecutive LOGICAL workgroups (= consecutive triangles = neighbouring texture
    // regions) go to one XCD.  It cuts L2->fabric fetches, but MEASURED SLOWER on the C3 workload
    // (k_fused 0.237 ms at run 1, 0.249 at 4, 0.262 at 16, 0.287 at 64): the look-back chain follows the
    // logical order, and a wave's predecessors are then dispatched up to kXcdRun rounds later, which
    // lengthens every wave's wait for its base offset.  Default 1 = plain round-robin.
    const uint32_t hb = blockIdx.x, xcd = hb & 7u, round = hb >> 3;
    const uint32_t lb = ((round / kXcdRun) * 8u + xcd) * kXcdRun + (round % kXcdRun);
    // global wave id == chain index (made scalar explicitly: the compiler cannot prove threadIdx.x>>6 uniform)
    const uint32_t wid = lb * (kBlock / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_waves = (sc.n_tri + tpw - 1u) / tpw;
    if (wid >= n_waves) return;
    const uint32_t t0 = wid * tpw;
    const uint32_t t = t0 + lane;
    const bool valid = (uint32_t)lane < tpw && t < sc.n_tri;
    const uint32_t lastT = min(t0 + tpw, sc.n_tri) - 1;
    bool uniform_mesh;
    const uint32_t m0 = mesh_of_range(sc, t0, lastT, uniform_mesh);   // one scalar load (was: a binary search)

    M2S_STAMP(0);
    // ---------------- triangle phase ----------------
    float p[9];
    Geo g;
    Raster rs;
    rs.x0 = rs.y0 = 0; rs.x1 = rs.y1 = -1; rs.ext = 0; rs.bias = 0; rs.area2 = 1;
#pragma unroll
    for (int i = 0; i < 3; i++) { rs.a[i] = rs.b[i] = 0; rs.c[i] = 0; }
    bool ok = false;
    uint32_t m = m0;
    float4 uvb0 = make_float4(0, 0, 0, 0);
    float2 uvb1 = make_float2(0, 0);
    if (valid) {
        load_positions(sc.tri, t, p);
        uvb0 = sc.tri.B0[t];   // only needed for the LODs of covered triangles, but requesting it here
        uvb1 = sc.tri.B1[t];   // puts it in the same memory round trip as the positions
        // (a wave inside one mesh — the common case — reads the mesh uniforms with scalar loads)
        if (uniform_mesh) geo_setup_mp(p, kConstMesh(sc.meshes + m0), g);
        else { m = find_mesh(sc, sc.tri_first + t); geo_setup_mp(p, sc.meshes + m, g); }
        ok = raster_setup(g, R, rs);
    }
    M2S_STAMP(1);  // positions loaded + GS/raster setup
    const int w = rs.x1 - rs.x0 + 1, rows = rs.y1 - rs.y0 + 1;
    int kind = kNone;
    unsigned long long mask = 0;
    uint32_t cnt = 0;
    if (ok) {
        if (w <= 8 && rows <= 8 && rs.ext <= 2304) {
            // small: every quantity fits int32 relative to the bbox origin pixel (|E| < 2^24)
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
                    if ((r0 | r1 | r2) >= 0) mask |= 1ull << (dy * 8 + dx);  // all three >= 0
                    r0 += ax0; r1 += ax1; r2 += ax2;
                }
                e0 += by0; e1 += by1; e2 += by2;
            }
            cnt = (uint32_t)__popcll(mask);
        } else if (rows <= kFusedRows) {
            kind = kMedium;
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            for (int y = rs.y0; y <= rs.y1; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                cnt += (uint32_t)max(xb - xa + 1, 0);
            }
            if (cnt > kBigCount) kind = kBig;
        } else {
            kind = kBig;   // emitted by the second stage; still counted here (its slice of the ordered output)
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
    {   // big triangles spanning very many rows: counted by the whole wave, one row per lane
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
    const uint32_t cntc = (kind == kSmall || kind == kMedium) ? cnt : 0;  // emitted by this kernel
    const uint32_t org = ((uint32_t)rs.y0 << 12) | (uint32_t)rs.x0;       // bbox origin pixel
    const bool anybig = __ballot(kind == kBig) != 0ull;

    M2S_STAMP(2);  // coverage counted
    // ---------------- ordering: wave scan + decoupled look-back ----------------
    // (a triangle has at most 4096^2 = 2^24 fragments, so the wave's 64 counts sum to < 2^31: 32-bit scans)
    const uint32_t incl = wave_incl_scan(cnt, lane);
    const uint32_t inclc = wave_incl_scan(cntc, lane);
    const unsigned long long total_w = __builtin_amdgcn_readlane(incl, 63);  // wave-uniform, in SGPRs
    const uint32_t total_c = __builtin_amdgcn_readlane(inclc, 63);
    const unsigned long long toff = incl - cnt;     // wave-local index of the first fragment (all kinds)
    const uint32_t ctoff = inclc - cntc;         // same, counting only fragments emitted here

    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    if (lane == 0) chain_store(&chain[wid], (wid == 0 ? kFlagPrefix : kFlagAgg) | etag | (total_w & kValMask));
    // The look-back itself is deferred to the first store: shading does not need the global offset, so
    // the chain latency overlaps with the first strip's work.
    unsigned long long base = 0;
    bool have_base = (wid == 0);
    auto resolve_base = [&]() {
        base = lookback(chain, wid, lane, epoch, status);
        if (lane == 0) {
            chain_store(&chain[wid], kFlagPrefix | etag | ((base + total_w) & kValMask));
            if (wid == n_waves - 1) *total_out = base + total_w;
        }
        have_base = true;
    };
    if (wid == 0 && lane == 0 && n_waves == 1) *total_out = total_w;
    if (total_c == 0 && !have_base) resolve_base();  // nothing to shade: settle the chain now

    M2S_STAMP(3);  // scanned + aggregate published
    // ---------------- per-triangle fragment constants -> LDS ----------------
    if (cntc) {
        TriShade ts;
        if (uniform_mesh) tri_shade_setup(p, g, rs, kConstMesh(sc.meshes + m0), uvb0, uvb1, ts);
        else tri_shade_setup(p, g, rs, sc.meshes + m, uvb0, uvb1, ts);
        ts.mesh |= m;   // low 24 bits; the top byte carries the combo sampler's mip levels
        const float4* src = reinterpret_cast<const float4*>(&ts);
#pragma unroll
        for (int k = 0; k < 5; ++k) L.tri[lane * 5 + k] = src[k];
        L.tskip[lane] = (uint32_t)(toff - ctoff);
    }
    L.park[lane] = make_uint4((uint32_t)mask, (uint32_t)(mask >> 32), ctoff | (cntc << 18) | ((uint32_t)kind << 30), org);
    // Deferred (big) triangles: they own their slice of the ordered output (it is part of this wave's total),
    // but are emitted by k_emit_big.  Needs the global base, so these (rare) waves resolve it right away.
    if (anybig) {
        if (!have_base) resolve_base();
        if (kind == kBig) {
            const uint32_t slot = atomicAdd(&bigmeta[0], 1u);
            atomicMax(&bigmeta[1], cnt);
            atomicAdd(&bigmeta[2], cnt);
            BigItem it;
            it.t = t; it.cnt = cnt; it.off = base + toff;
            biglist[slot] = it;
        }
        // status lives in host-mapped memory: plain (idempotent) system-scope stores, no PCIe atomics needed
        if (lane == 0) __hip_atomic_store(&status[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // ---------------- fragment phase, in windows of kWaveEntryCap ----------------
    for (uint32_t win = 0; win < total_c; win += kWaveEntryCap) {
        const uint32_t wend = win + kWaveEntryCap;
        {
            const uint4 pk = L.park[lane];
            const uint32_t kd = pk.z >> 30, cn = (pk.z >> 18) & 0xFFFu, cto = pk.z & 0x3FFFFu;
            if (kd == kSmall && cto < wend && cto + cn > win) {
                unsigned long long mm = ((unsigned long long)pk.y << 32) | pk.x;
                const uint32_t tag = ((uint32_t)lane << 24) | pk.w;
                uint32_t ci = cto;
                while (mm) {
                    const int bit = __ffsll((long long)mm) - 1;
                    mm &= mm - 1;
                    if (ci >= win && ci < wend) L.entries[ci - win] = tag + (((uint32_t)(bit >> 3) << 12) | (uint32_t)(bit & 7));
                    ++ci;
                }
            } else if (kd == kMedium && cto < wend && cto + cn > win) {
                expand_medium(sc.tri.A0, sc.tri.A1, sc.tri.A2, sc.meshes + (reinterpret_cast<const uint32_t*>(&L.tri[lane * 5 + 4])[3] & 0xFFFFFFu), t0 + lane, R, (uint32_t)lane, cto, win, wend, L.entries);
            }
        }
        wave_lds_sync();  // entries + tri visible to the whole wave
        if (win == 0) M2S_STAMP(5);  // first window expanded
        const uint32_t nwin = min((uint32_t)kWaveEntryCap, total_c - win);
        for (uint32_t e0 = 0; e0 < nwin; e0 += 64) {
            const uint32_t e = e0 + lane;
            const bool have = e < nwin;
            float4 rec[6];
            uint32_t skipped = 0;  // fragments of deferred (big) triangles preceding this fragment's triangle
            uint32_t en = 0, slot = 0, mymesh = m0;
            if (have) {
                en = L.entries[e];
                slot = en >> 24;
                if (!uniform_mesh) mymesh = reinterpret_cast<const uint32_t*>(&L.tri[slot * 5 + 4])[3] & 0xFFFFFFu;
                if (anybig) skipped = L.tskip[slot];
            }
            if (have) {
                // the TriShade fields are read from LDS where they are used rather than copied (80 B = 20 VGPRs) up front:
                // 166 -> 160 VGPRs, no scratch, k_fused 0.1845 -> 0.1816 ms
                const TriShade& ts = *reinterpret_cast<const TriShade*>(&L.tri[slot * 5]);
                // Two copies of the shading code on purpose.  Wave inside one mesh (the common case): the descriptor
                // pointer is an SGPR pair in the constant address space, so sizes / level offsets / texel base come
                // from scalar loads and the texel fetches use the saddr + 32-bit offset form.  Wave straddling a mesh
                // boundary: per-lane pointer.  (A single call with a selected pointer made EVERY descriptor read a
                // per-lane vector load: 38 instead of 19 vector loads per strip.)
#ifdef M2S_TIMING
                unsigned long long st3[3] = { 0, 0, 0 };
                if (uniform_mesh) shade_from_tri(sc.tri, t0 + slot, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), kConstMesh(sc.meshes + m0), ts, rec, st3);
                else shade_from_tri(sc.tri, t0 + slot, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), sc.meshes + mymesh, ts, rec, st3);
                if (win == 0 && e0 == 0 && lane == 0 && wid < kTimingWaves)
                    for (int k = 0; k < 3; ++k) g_timing[(12 + k) * kTimingWaves + wid] = st3[k];
#else
                if (uniform_mesh) shade_from_tri(sc.tri, t0 + slot, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), kConstMesh(sc.meshes + m0), ts, rec);
                else shade_from_tri(sc.tri, t0 + slot, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), sc.meshes + mymesh, ts, rec);
#endif
            }
            if (win == 0 && e0 == 0) M2S_STAMP(6);  // first strip shaded
            if (!have_base) resolve_base();
            if (win == 0 && e0 == 0) M2S_STAMP(7);  // base resolved
            const unsigned long long oidx = base + skipped + win + e;
            if (!anybig) {
                // the strip's records are consecutive in the output: stage half a wave at a time, then
                // 16 B/lane fully coalesced stores (3 KiB contiguous per half)
                const unsigned long long o0 = base + win + e0;
                uint32_t nvalid = min(64u, nwin - e0);
                if (o0 + 64ull > limit) {   // (wave-uniform) only strips that straddle the cap need the 64-bit clipping
                    if (o0 >= limit) nvalid = 0;
                    else if (limit - o0 < nvalid) nvalid = (uint32_t)(limit - o0);
                }
#pragma unroll 1
                for (int half = 0; half < 2; ++half) {
                    if (have && (lane >> 5) == half) {
#pragma unroll
                        for (int k = 0; k < 6; ++k) L.stage[(lane & 31) * 6 + k] = rec[k];
                    }
                    wave_lds_sync();
                    float4* __restrict__ dsto = out + (o0 + 32u * half) * 6;
                    const uint32_t nv = nvalid > 32u * half ? min(32u, nvalid - 32u * half) : 0u;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const uint32_t q = (uint32_t)lane + 64u * j;
                        const uint32_t r = q / 6u;
                        // non-temporal: the records are never re-read by this kernel; keeping them out of the
                        // 4 MiB L2 leaves it to the texture / vertex lines (measured: k_fused 0.236 -> 0.200 ms)
                        if (r < nv) nt_store(&dsto[q], L.stage[q]);
                    }
                    wave_lds_sync();
                }
            } else if (have && oidx < limit) {
                float4* __restrict__ dsto = out + oidx * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) nt_store(&dsto[k], rec[k]);
            }
        }
        wave_lds_sync();  // the next window overwrites entries
    }
    M2S_STAMP(8);  // done
#ifdef M2S_TIMING
    if (lane == 0 && wid < kTimingWaves) { g_timing[9 * kTimingWaves + wid] = total_c; g_timing[10 * kTimingWaves + wid] = __builtin_amdgcn_s_getreg(6164) /*XCC_ID*/; }
#endif
}

void launch_fused(const SceneDev& sc, uint32_t R, unsigned long long* chain, uint64_t limit, float4* out,
                  unsigned long long* total, uint32_t* status, uint32_t epoch, BigItem* biglist, uint32_t* bigmeta, hipStream_t st) {
    // Small scenes: fewer triangles per wave, so that there are enough waves to occupy the GPU (3072 wave slots) and
    // each wave's serial chain of phases is shorter.  The fragment phase still runs full 64-lane strips.
    const uint32_t tpw = fused_tpw(sc.n_tri);
    uint32_t nb = (n_fused_waves(sc.n_tri) + kBlock / 64 - 1) / (kBlock / 64);
    if (!nb) return;
    nb = (nb + 8 * kXcdRun - 1) / (8 * kXcdRun) * (8 * kXcdRun);  // whole XCD runs; surplus workgroups exit at once
    hipLaunchKernelGGL(k_fused, dim3(nb), dim3(kBlock), 0, st, sc, R, chain, (unsigned long long)limit, out, total, status, epoch & 0xFFFFu, biglist, bigmeta, tpw);
}

#ifdef M2S_TIMING
extern "C" int m2s_debug_read_timing(unsigned long long* dst, size_t n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_timing), n * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_fused() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_fused)); }

}  // namespace m2s
