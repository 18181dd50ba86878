// m2s_emit2.hip — the multi-pass pipeline, second generation (gfx950): two kernels, every triangle set up ONCE.
//
// The first generation (m2s_kernels.hip: k_count -> k_scan_partials -> k_offsets -> k_emit) is what scenes with mid-size
// and large triangles take (tens to millions of fragments per triangle: Sponza-like content, coarse meshes at high
// density).  Measured on the C4 stand-in (248 832 triangles, 8.27 M fragments, 7 M stored; profiles/r02): k_emit 0.336 ms
// = 2.1 TB/s of algorithmic bytes, waves waiting 70 % of their cycles; k_count 0.062 ms with ONE workgroup per CU; two
// tiny kernels plus three dependent-launch gaps in between.  What was wrong with it:
//   * k_emit's workgroup of 4 waves marches in lock step: expansion (a few dozen busy lanes), barrier, then four rounds of
//     shade -> barrier -> store -> barrier; 48 KB of LDS per workgroup;
//   * every workgroup sets its triangles up again (positions, mesh lookup, GS, raster setup, Scale / quaternion / LODs);
//   * k_count walks 4 triangles per thread in 243 workgroups: a latency chain with 4 waves per CU.
//
// Here:
//   k_count_scan  1 thread / triangle, 256 triangles / workgroup.  GS + raster setup + exact count (row walker), the
//                 fragment stage's per-triangle constants (TriShade) and the three edge functions are written to a
//                 112-byte TriSetup record, and the SAME kernel turns the counts into output offsets with a decoupled
//                 look-back over the context's chain words (one word per workgroup): no scan kernel, no offsets kernel.
//                 It also fills start[]: the triangle that owns output record m * 512.
//                 (Round 6, from the kernel's per-wave timeline — tools/timeline_probe.py on a -DM2S_TIMELINE build: the rows are
//                 walked in 32 bits (RowWalker32, m2s_devfn.h), positions / texture coordinates / geometry-stage output wait in
//                 28 KB of LDS across the row loops instead of in registers, the TriSetup records leave as contiguous 1 KB runs, and
//                 the kernel is two halves — count_block_a up to the published aggregate, count_block_b from the look-back on — so
//                 that a launch of the resident workgroups can take a few EXTRA blocks by ticket: see count_block_a.)
//   k_emit2       every WAVE owns 512 consecutive output records and never synchronises with another wave: it loads the
//                 TriSetup of the (typically 5-60) triangles overlapping its slice, expands them into an LDS entry list
//                 (row walker; triangles of more than 32 rows wave-cooperatively), and shades strips of 64 entries
//                 exactly like the single-pass kernel does: record staged half a wave at a time, 16 B/lane non-temporal
//                 stores.  10 KB of LDS per wave.  Output-partitioned, so balanced for ANY triangle size.
//   fine blocks   (round 5) Output partitioning has a bad case of its own: where triangles are much SMALLER than a pixel (foliage,
//                 distant cloth: a fragment every dozen triangles) a slice of 512 records spans thousands of triangles, i.e. dozens of
//                 64-triangle batches — each a dependent chain of global round trips (offsets, TriSetup, attributes, texels) —
//                 walked by ONE wave: the heterogeneous scene's k_emit2 took 0.26 ms for 4.3 M fragments, 2.2 times the time per
//                 fragment of the C4 stand-in, waiting for the ~30 waves inside its two foliage meshes.  k_count_scan therefore
//                 classes every block of 256 triangles: FINE if the whole block yields at most 2048 fragments (and none of its
//                 triangles is taller than 32 pixel rows).  Fine blocks are emitted triangle-partitioned — one workgroup per block,
//                 one thread per triangle, the block's whole entry list in LDS, no inter-workgroup dependency because the offsets are
//                 known — by extra workgroups of the SAME launch (emit_fine_block); the output-partitioned slices step over them
//                 (one scalar load per block).  The pipeline choice has become a per-256-triangle decision taken on the device.
//                 (Round 6 tried the two obvious next steps and measured both SLOWER, same bytes: the workgroup of k_count_scan that
//                 counted a fine block emits it — the count stage lasts as long as its slowest workgroup and k_emit2 lost the
//                 workgroups that filled its tail, tag r6-fine-fold-in-count —; and the conversion cut into two chunks of blocks,
//                 count(B) beside emit(A) on a second stream — three cross-queue waits on the critical path of a 0.16 ms
//                 conversion, tag r6-two-stream-chunks —; and count and emit as two roles of ONE launch with in-kernel dependencies
//                 between chunks — waiting emitters hold workgroup slots and stall the in-order dispatcher, 1.6-2x slower, tag
//                 r6-one-launch-multipass.  profiles/r06/negative_*.log.)
// Output: bit-identical to every other pipeline (same device functions, same operation order).
#include <cstdlib>
#include "m2s_fused_common.h"
#include <algorithm>

#pragma clang fp contract(off)

namespace m2s {

#ifndef M2S_EMIT2_SLICE
#define M2S_EMIT2_SLICE 512
#endif
constexpr int kSlice = M2S_EMIT2_SLICE;   // output records per wave in k_emit2
constexpr int kCountBlock = 256;       // triangles per workgroup in k_count_scan
// waves per SIMD the two kernels are compiled for (A/B switches; see DESIGN.md for the measurements behind the defaults)
#ifndef M2S_COUNT_WAVES
#define M2S_COUNT_WAVES 4
#endif
#ifndef M2S_COUNT_ROWS
#define M2S_COUNT_ROWS kRowsCount   // k_count_scan: triangles of more pixel rows are counted by the whole wave, 64 rows at a time (A/B switch)
#endif
#ifndef M2S_EMIT2_WAVES
#define M2S_EMIT2_WAVES 4   // round 3: 122 VGPRs since both mip levels are read without a branch: four waves per SIMD, no scratch
#endif

// What k_emit2 needs to know about a triangle: the fragment stage's constants plus the third edge function and the
// pixel bounding box (the first two edges and the box origin are in the TriShade).  7 x 16 bytes.
struct TriSetup {
    TriShade ts;
    int a0, b0;          // edge function opposite vertex 0 ...
    long long e0;        // ... and its value at the centre of pixel (x0, y0)
    uint32_t ext;        // x1 | y1 << 12 | bias << 24
    uint32_t tall;       // triangles of more than kRowsCount pixel rows: 1 + their slot in the tall-triangle table (0: none)
    uint32_t pad[2];
};
static_assert(sizeof(TriSetup) == 112, "TriSetup must be seven float4");

// Tall-triangle table (round 5), behind the TriSetup array in the same allocation: a 16-byte header (word 0: slots handed out in
// this conversion) and kTallCap slots of 64 words.  Word j of a slot = fragments of its triangle in the pixel rows BEFORE row
// y0 + 64 j.  k_count_scan counts such a triangle 64 rows at a time anyway and writes the running sums down; a slice of k_emit2
// that starts deep inside the triangle (the floor of a Sponza-like scene: half a million fragments, a thousand slices, a
// thousand rows) then begins at the 64-row chunk that holds its first record instead of walking every chunk above it.  A
// conversion with more tall triangles than slots leaves the rest without one: they are walked from the top, as before.
constexpr uint32_t kTallCap = 16384;
constexpr uint32_t kTallChunks = 64;          // 4096 rows / 64
__device__ __forceinline__ uint32_t* tall_header(const float4* setup, uint32_t n_tri) {
    return reinterpret_cast<uint32_t*>(const_cast<float4*>(setup) + (size_t)max(n_tri, 1u) * 7);
}
// ... and behind that table one byte per block of kCountBlock triangles: 1 = fine block (see the file header)
constexpr uint32_t kFineMax = 2048;           // fragments of a fine block: its entry list fits the workgroup's LDS
__device__ __forceinline__ uint8_t* block_class(const float4* setup, uint32_t n_tri) {
    return reinterpret_cast<uint8_t*>(tall_header(setup, n_tri) + 4 + (size_t)kTallCap * kTallChunks);
}

#ifdef M2S_TIMELINE
// Measurement build only (tools/timeline_probe.py; never part of the shipping library): every wave of the two kernels leaves
// 100 MHz timestamps behind — k_count_scan eight per wave (its phases), k_emit2 start / end / batches / kind per wave.
constexpr uint32_t kTlCountBlocks = 8192, kTlEmitWgs = 32768;
__device__ unsigned long long g_tl_count[kTlCountBlocks * 4 * 8];
__device__ unsigned long long g_tl_emit[kTlEmitWgs * 4 * 4];
#define TLC(k) do { __builtin_amdgcn_sched_barrier(0); if (lane == 0 && bid < kTlCountBlocks) g_tl_count[((size_t)bid * 4 + wave) * 8 + (k)] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } while (0)
struct TlEmitScope {
    unsigned long long* p; unsigned long long nb, kind;
    __device__ TlEmitScope(uint32_t wg, uint32_t wave, int lane) : p(nullptr), nb(0), kind(0) {
        if (lane == 0 && wg < kTlEmitWgs) { p = &g_tl_emit[((size_t)wg * 4 + wave) * 4]; p[0] = wall_clock64(); }
    }
    __device__ ~TlEmitScope() { if (p) { p[1] = wall_clock64(); p[2] = nb; p[3] = kind; } }
};
#else
#define TLC(k) do { } while (0)
#endif

__device__ __forceinline__ Raster raster_from_setup(const TriSetup& s) {
    Raster r;
    const int x0 = (int)(s.ts.org & 0xFFFu), y0 = (int)(s.ts.org >> 12);
    const long long Px0 = 256ll * x0 + 128, Py0 = 256ll * y0 + 128;
    r.a[0] = s.a0; r.b[0] = s.b0; r.a[1] = s.ts.a1; r.b[1] = s.ts.b1; r.a[2] = s.ts.a2; r.b[2] = s.ts.b2;
    r.c[0] = s.e0 - (long long)s.a0 * Px0 - (long long)s.b0 * Py0;
    r.c[1] = s.ts.e1 - (long long)s.ts.a1 * Px0 - (long long)s.ts.b1 * Py0;
    r.c[2] = s.ts.e2 - (long long)s.ts.a2 * Px0 - (long long)s.ts.b2 * Py0;
    r.area2 = 1; r.ext = 0;
    r.bias = (int)(s.ext >> 24);
    r.x0 = x0; r.y0 = y0; r.x1 = (int)(s.ext & 0xFFFu); r.y1 = (int)((s.ext >> 12) & 0xFFFu);
    return r;
}

// ============================================================================================
// k_count_scan
// ============================================================================================
// k_count_scan in two halves (round 6).  Half A of block `bid`: everything up to the published aggregate and the TriSetup records; it
// hands the triangles' counts and block-local offsets on in registers.  Half B: the look-back, off[] made global, start[].  A workgroup runs A of its own block,
// then A of whatever EXTRA blocks it can take by ticket, then the B halves.  A scene with a few more blocks than the GPU holds
// workgroups (the heterogeneous scene: 1043 for 1024 slots; Sponza has 262 k triangles) used to run them as a second generation that
// could only start when the first workgroup of the first one left — and every workgroup of the first generation waits, in its
// look-back, for the slowest aggregate before it (timeline: 19 blocks alone on the GPU from 23 to 34 us of the kernel).  Now such a
// launch holds exactly the resident workgroups and the extra blocks are counted by the workgroups that finish their own A first, WHILE
// the slow ones are still counting; nobody waits before every aggregate it is going to produce is out, so the look-backs cannot
// deadlock.  (The persistent forms of rounds 5 / 6 ticketed EVERY block and kept the look-back inside the loop:
// profiles/r06/count_scan_fault_analysis.md.)
__device__ __forceinline__ void count_block_a(const SceneDev& sc, uint32_t R, uint32_t* __restrict__ off,
                                              unsigned long long* __restrict__ chain, uint32_t epoch, float4* __restrict__ setup,
                                              const uint32_t bid, uint32_t* wsum, uint32_t* wtall, float4* stash,
                                              uint32_t& c_out, unsigned long long& loc_out, uint32_t& tot_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = bid * kCountBlock;
    const uint32_t t = blockBase + threadIdx.x;
    const bool valid = t < sc.n_tri;
    const uint32_t lastT = min(blockBase + kCountBlock, sc.n_tri) - 1;
    bool uniform_mesh;
    TLC(0);
    const uint32_t m0 = mesh_of_range(sc, blockBase, lastT, uniform_mesh);   // one scalar load (was: a binary search)

    float p[9];
    Geo g;
    Raster rs;
    bool ok = false;
    uint32_t m = m0;
    if (valid) {
        load_positions(sc.tri, t, p);
        const float4 uvb0 = sc.tri.B0[t];
        const float2 uvb1 = sc.tri.B1[t];
        // (a workgroup inside one mesh — the common case — reads the mesh uniforms with scalar loads)
        if (uniform_mesh) geo_setup_mp(p, kConstMesh(sc.meshes + m0), g);
        else { m = find_mesh(sc, sc.tri_first + t); geo_setup_mp(p, sc.meshes + m, g); }
        ok = raster_setup(g, R, rs);
        // Positions and texture coordinates wait in LDS for the second half of the kernel (tri_shade_setup, behind the published
        // aggregate) instead of in fifteen registers across the row loops: with them the kernel spilled 80 bytes per lane once the 32-bit
        // walker joined it (the spill traffic cost the C4 stand-in more than the walker saved), and reading them again from global
        // memory put a 3 us round trip — behind every wave's 112-byte-strided TriSetup stores — where 0.1 us of LDS does.
        if (ok) {
            float4* const mine_s = stash + wave * (7 * 64) + lane;      // [wave][k][lane]: the wave's own 7 KB
            mine_s[0 * 64] = make_float4(p[0], p[1], p[2], p[3]);
            mine_s[1 * 64] = make_float4(p[4], p[5], p[6], p[7]);
            mine_s[2 * 64] = make_float4(p[8], uvb1.x, uvb1.y, 0.0f);
            mine_s[3 * 64] = uvb0;
            // ... and so does the geometry stage's output, which only the second half reads (the raster setup is done with it)
            mine_s[4 * 64] = make_float4(g.xx, g.xy, g.xz, g.nx);
            mine_s[5 * 64] = make_float4(g.ny, g.nz, g.ou[0], g.ou[1]);
            mine_s[6 * 64] = make_float4(g.ou[2], g.ov[0], g.ov[1], g.ov[2]);
        }
    }
    const int rows = ok ? rs.y1 - rs.y0 + 1 : 0;
    uint32_t c = 0, c64 = 0;
    TLC(1);
    // The 32-bit walker (m2s_devfn.h) of EVERY triangle, at its first row that passes the horizontal edges; a triangle outside the walker's
    // range — a nearly horizontal long edge — is counted by the whole wave through the closed form, below.
    const bool mine = ok && rows <= M2S_COUNT_ROWS;
    RowWalker32 rw;
    rw.k0 = 0; rw.k1 = -1;
    bool safe32 = false;
    if (ok) safe32 = row_walker32_init(rs, rs.y0, 1, rows, rw);
    const bool walked = mine && safe32;
    if (walked) {
        RowWalker32 w = rw;
        for (int k = w.k0; k <= w.k1; ++k) {
            int xa, xb;
            row_walker32_next(w, xa, xb);
            if (k == 64) c64 = c;          // (running sum in front of the second 64-row chunk: the tall-triangle table, below)
            c += (uint32_t)max(xb - xa + 1, 0);
        }
        if (w.k1 < 64) c64 = c;            // (a horizontal edge ends the walk above row 64: everything lies in the first chunk)
    }
    uint32_t tall_slot = 0;
    TLC(2);
    {   // Triangles of more rows are counted by the whole wave, in CHUNKS of 64 rows — and leave the running sums in front of every chunk
        // in the tall-triangle table (see kTallCap) for k_emit2.  Triangles of 65 .. M2S_COUNT_ROWS rows, counted by their own lanes
        // above, get a table entry as well (two chunks: 0 and the sum after 64 rows): a slice of k_emit2 that starts in their lower
        // half skips the upper one.
        const bool tall32 = ok && !mine && safe32;
        unsigned long long big = __ballot(ok && !walked && !tall32);       // (outside the 32-bit walker's range: the closed form, one triangle at a time)
        const unsigned long long talls = __ballot(tall32);
        const bool two = walked && rows > 64 && c != 0;
        const unsigned long long twos = __ballot(two);
        uint32_t* const hdr = tall_header(setup, sc.n_tri);
        // the wave's table slots in ONE atomic (one global round trip per wave, not one per tall triangle in front of its count)
        uint32_t slot = 0;
        if ((big | talls | twos) != 0ull && lane == 0) slot = atomicAdd(&hdr[0], (uint32_t)(__popcll(big) + __popcll(talls) + __popcll(twos)));
        slot = __builtin_amdgcn_readfirstlane(slot);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (two) {
            const uint32_t my = slot + (uint32_t)__popcll(big) + (uint32_t)__popcll(talls) + (uint32_t)__popcll(twos & below);
            if (my < kTallCap) {
                uint32_t* const row = hdr + 4 + (size_t)my * kTallChunks;
                row[0] = 0; row[1] = c64;
                tall_slot = my + 1u;
            }
        }
        if (talls) {
            // (triangle, chunk) TASKS, one per lane and round: the lane fetches the triangle's walker from the lane that owns it, jumps it to
            // the chunk's first row and walks the chunk's rows itself.  Until round 6 the wave walked ONE triangle at a time, a row per
            // lane and 64-row step, with a closed-form setup per lane and triangle (~800 instructions) and a wave-wide sum per step: the
            // floor of the heterogeneous scene — two triangles — held its block's aggregate for 12-16 us, and a wave of 64 wall-sized
            // triangles (any low-polygon mesh at a high density) would have walked them one after the other.
            const uint32_t my = slot + (uint32_t)__popcll(big) + (uint32_t)__popcll(talls & below);
            const uint32_t nch = tall32 ? (uint32_t)(rows + 63) >> 6 : 0u;                 // <= 64 (a pixel box has at most 4096 rows)
            const uint32_t incl_c = wave_incl_scan(nch, lane);
            const uint32_t n_chunks = (uint32_t)__builtin_amdgcn_readlane((int)incl_c, 63);
            // few chunks (a floor, a pair of walls): every chunk is cut into 2 / 4 / 8 tasks so that the one round has work for every lane
            uint32_t sub = 0;
            while (sub < 3u && (n_chunks << (sub + 1u)) <= 64u) ++sub;
            const uint32_t rpt = 64u >> sub;                                                // rows per task
            const uint32_t incl_t = incl_c << sub, excl_t = incl_t - (nch << sub), n_tasks = n_chunks << sub;
            uint32_t carry = 0;                                                             // an owner's fragments in the chunks finished so far
            for (uint32_t j0 = 0; j0 < n_tasks; j0 += 64u) {
                const uint32_t j = j0 + (uint32_t)lane;
                const bool task = j < n_tasks;
                // the owner: the first lane whose inclusive count of chunks exceeds j (lanes without chunks repeat their predecessor's)
                int lo = 0, hi = 63;
#pragma unroll
                for (int it = 0; it < 6; ++it) {
                    const int mid = (lo + hi) >> 1;
                    const uint32_t v = (uint32_t)__shfl((int)incl_t, mid);
                    if (v > j) hi = mid; else lo = mid + 1;
                }
                const int owner = task ? min(lo, 63) : lane;
                const uint32_t tidx = j - (uint32_t)__shfl((int)excl_t, owner);
                const uint32_t cidx = tidx >> sub, part = tidx & ((1u << sub) - 1u);
                RowWalker32 w;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    w.q[i] = __shfl(rw.q[i], owner); w.r[i] = (uint32_t)__shfl((int)rw.r[i], owner); w.sq[i] = __shfl(rw.sq[i], owner);
                    w.sr[i] = (uint32_t)__shfl((int)rw.sr[i], owner); w.D[i] = (uint32_t)__shfl((int)rw.D[i], owner);
                }
                w.lower = __shfl(rw.lower, owner); w.x0 = __shfl(rw.x0, owner); w.x1 = __shfl(rw.x1, owner);
                w.k0 = __shfl(rw.k0, owner); w.k1 = __shfl(rw.k1, owner);
                const uint32_t o_slot = (uint32_t)__shfl((int)my, owner);
                const uint32_t o_carry = (uint32_t)__shfl((int)carry, owner);
                const int row0 = (int)(64u * cidx + part * rpt);
                const int first = max(row0, w.k0), last = min(row0 + (int)rpt - 1, w.k1);
                uint32_t sum = 0;
                if (task && first <= last) {
                    row_walker32_jump(w, (uint32_t)(first - w.k0));
                    for (int k = first; k <= last; ++k) {
                        int xa, xb;
                        row_walker32_next(w, xa, xb);
                        sum += (uint32_t)max(xb - xa + 1, 0);
                    }
                }
                // running sums per triangle: a segmented scan over the round's lanes (the tasks of one triangle are neighbours) on top of
                // what the owner carries from the rounds before
                uint32_t seg = sum;
                const uint32_t okey = task ? (uint32_t)owner : 0xFFFFFFFFu;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t v = (uint32_t)__shfl_up((int)seg, d);
                    const uint32_t o = (uint32_t)__shfl_up((int)okey, d);
                    if (lane >= d && o == okey) seg += v;
                }
                const uint32_t before = o_carry + seg - sum;                                // fragments of the triangle in front of this chunk
                if (task && part == 0u && o_slot < kTallCap && cidx < kTallChunks) hdr[4 + (size_t)o_slot * kTallChunks + cidx] = before;
                // every owner takes the total behind ITS last task of this round
                const uint32_t a = max(excl_t, j0), b = min(incl_t, j0 + 64u);
                const uint32_t got = (uint32_t)__shfl((int)(before + sum), b > a ? (int)(b - 1u - j0) : lane);
                if (tall32 && b > a) carry = got;
            }
            if (tall32) { c = carry; tall_slot = my < kTallCap ? my + 1u : 0u; }
        }
        for (; big; ++slot) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const Raster b = shfl_raster(rs, src);
            const bool listed = slot < kTallCap;
            uint32_t* const row = hdr + 4 + (size_t)slot * kTallChunks;
            uint32_t run = 0, ci = 0;
            // the closed form per row (two fp64 reciprocals per edge and row), lane l: rows y0 + l, y0 + l + 64, ...
            for (int yc = b.y0; yc <= b.y1; yc += 64, ++ci) {
                const int y = yc + lane;
                int xa = 0, xb = -1;
                if (y <= b.y1) row_span(b, y, xa, xb);
                if (listed && lane == 0 && ci < kTallChunks) row[ci] = run;
                run += wave_sum((uint32_t)max(xb - xa + 1, 0));
            }
            if (lane == src) { c = run; tall_slot = listed ? slot + 1u : 0u; }
        }
    }
    // ---- counts -> offsets: workgroup scan + decoupled look-back (chain word = one per workgroup) ----
    TLC(3);
    const uint32_t incl = wave_incl_scan(c, lane);
    const bool wave_tall = __ballot(ok && rows > kRowsThread) != 0ull;
    if (lane == 63) { wsum[wave] = incl; wtall[wave] = wave_tall ? 1u : 0u; }
    __syncthreads();
    uint32_t woff = 0, tot = 0, any_tall = 0;
#pragma unroll
    for (int w = 0; w < kCountBlock / 64; ++w) {
        if (w < wave) woff += wsum[w];
        tot += wsum[w];
        any_tall |= wtall[w];
    }
    // the block's class: who emits its fragments (see the file header)
    if (threadIdx.x == 0) block_class(setup, sc.n_tri)[bid] = (tot <= kFineMax && !any_tall) ? 1 : 0;
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    // The workgroup's aggregate is published as soon as it is known — BEFORE the per-triangle setup records are computed and
    // stored: successors can resolve their bases while this workgroup is still busy, and this workgroup's own look-back
    // (below) finds its predecessors' words already in place.  (Setup first, then publish: k_count_scan 0.042 ms on the C4
    // stand-in; without any setup 0.027 ms, without the look-back 0.030 ms: the two used to add up on the critical path.)
    if (wave == 0 && lane == 0)
        chain_store(&chain[bid], (bid == 0 ? kFlagPrefix : kFlagAgg) | etag | ((unsigned long long)tot & kValMask));
    TLC(4);
    {   // the per-triangle half of the fragment stage, once: k_emit2 only reads it
        TriSetup s;
        float4* const wave_s = stash + wave * (7 * 64);
        if (c) {
            // positions, texture coordinates and the geometry stage's output come back from the workgroup's LDS (see `stash` above)
            const float4* const mine_s = wave_s + lane;
            const float4 s0 = mine_s[0 * 64], s1 = mine_s[1 * 64], s2 = mine_s[2 * 64], s3 = mine_s[3 * 64];
            p[0] = s0.x; p[1] = s0.y; p[2] = s0.z; p[3] = s0.w; p[4] = s1.x; p[5] = s1.y; p[6] = s1.z; p[7] = s1.w; p[8] = s2.x;
            const float4 uvb0 = s3;
            const float2 uvb1 = make_float2(s2.y, s2.z);
            {
                const float4 g0 = mine_s[4 * 64], g1 = mine_s[5 * 64], g2 = mine_s[6 * 64];
                g.xx = g0.x; g.xy = g0.y; g.xz = g0.z; g.nx = g0.w; g.ny = g1.x; g.nz = g1.y;
                g.ou[0] = g1.z; g.ou[1] = g1.w; g.ou[2] = g2.x; g.ov[0] = g2.y; g.ov[1] = g2.z; g.ov[2] = g2.w;
            }
            if (uniform_mesh) tri_shade_setup(p, g, rs, kConstMesh(sc.meshes + m0), uvb0, uvb1, s.ts);
            else tri_shade_setup(p, g, rs, sc.meshes + m, uvb0, uvb1, s.ts);
            s.ts.mesh |= m;
            const long long Px0 = 256ll * rs.x0 + 128, Py0 = 256ll * rs.y0 + 128;
            s.a0 = rs.a[0]; s.b0 = rs.b[0];
            s.e0 = (long long)rs.a[0] * Px0 + (long long)rs.b[0] * Py0 + rs.c[0];
            s.ext = (uint32_t)rs.x1 | ((uint32_t)rs.y1 << 12) | ((uint32_t)rs.bias << 24);
            s.tall = tall_slot;
            s.pad[0] = s.pad[1] = 0;
        }
        // The records leave through the wave's 7 KB of LDS as seven contiguous 1 KB runs (one triangle = 112 bytes: a lane storing its
        // own record touches a line per 16 bytes — 448 write requests per wave, and the L2's request rate, not its bandwidth, kept the
        // waves of the C4 stand-in 7 us in these stores: tools/timeline_probe.py).  Pieces of triangles without fragments stay unwritten.
        wave_lds_sync();                                  // every lane has read its stash entries
        if (c) {
            const float4* src4 = reinterpret_cast<const float4*>(&s);
#pragma unroll
            for (int k = 0; k < 7; ++k) wave_s[lane * 7 + k] = src4[k];
        }
        wave_lds_sync();
        const unsigned long long cmask = __ballot(c != 0);
        float4* const dst4 = setup + (size_t)(blockBase + (uint32_t)wave * 64u) * 7;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const uint32_t q = (uint32_t)lane + 64u * j;
            if ((cmask >> (q / 7u)) & 1ull) dst4[q] = wave_s[q];
        }
    }

    // what half B needs: the triangle's count, its offset inside the block, the block's aggregate (registers: both halves are
    // straight-line code of one kernel)
    c_out = c; loc_out = (unsigned long long)woff + (incl - c); tot_out = tot;
    TLC(5);
}

// Half B of block `bid` (after half A of the same block by the same workgroup).
__device__ __forceinline__ void count_block_b(const SceneDev& sc, uint32_t* __restrict__ off, uint32_t* __restrict__ start, uint32_t n_start,
                                              unsigned long long* __restrict__ chain, uint32_t epoch, unsigned long long* __restrict__ total_out,
                                              uint32_t* __restrict__ status, unsigned long long* __restrict__ total_host,
                                              const uint32_t bid, const uint32_t n_tb, unsigned long long* base_sp,
                                              const uint32_t c, const unsigned long long loc, const uint32_t tot) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = bid * kCountBlock;
    const uint32_t t = blockBase + threadIdx.x;
    const bool valid = t < sc.n_tri;
    const unsigned long long etag = (unsigned long long)epoch << kEpochShift;
    if (wave == 0) {
        const uint32_t b = bid;
        const unsigned long long base = b == 0 ? 0ull : lookback(chain, b, lane, epoch, status);
        if (lane == 0) {
            if (b) chain_store(&chain[b], kFlagPrefix | etag | ((base + tot) & kValMask));
            base_sp[0] = base;
            if (b == n_tb - 1) {
                *total_out = base + tot;
                // the counter the host waits for: written by the kernel itself (like the single-pass kernels), no copy behind the pipeline
                if (total_host) __hip_atomic_store(total_host, base + tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    __syncthreads();
    TLC(6);
    const unsigned long long base = base_sp[0];
    const unsigned long long o0 = base + loc;
    if (valid) {
        // offsets are 32-bit (the host rejects totals beyond 2^32 - 1); saturate instead of wrapping
        const unsigned long long o1 = o0 + c;
        off[t] = (uint32_t)min(o0, 0xFFFFFFFFull);
        if (t == sc.n_tri - 1) off[sc.n_tri] = (uint32_t)min(o1, 0xFFFFFFFFull);
    }
    // start[m] = the triangle that owns output record m * kSlice.  A triangle covering many slices (up to 32 768 for a
    // 4096 x 4096 px one) has the whole wave write them.
    const unsigned long long mf = (o0 + kSlice - 1) / kSlice, ml = c ? (o0 + c - 1) / kSlice : 0;
    const bool few = valid && c && ml >= mf && (ml - mf) < 16;
    if (few)
        for (unsigned long long mm = mf; mm <= ml && mm < n_start; ++mm) start[mm] = t;
    unsigned long long many = __ballot(valid && c && ml >= mf && !few);
    while (many) {
        const int src = __ffsll((long long)many) - 1;
        many &= many - 1;
        const unsigned long long f = __shfl(mf, src), l = __shfl(ml, src);
        const uint32_t tt = __shfl(t, src);
        for (unsigned long long mm = f + lane; mm <= l && mm < n_start; mm += 64) start[mm] = tt;
    }
    TLC(7);
}


__global__ void __launch_bounds__(kCountBlock, M2S_COUNT_WAVES) k_count_scan(SceneDev sc, uint32_t R, uint32_t* __restrict__ off,
                                                            uint32_t* __restrict__ start, uint32_t n_start,
                                                            unsigned long long* __restrict__ chain, uint32_t epoch,
                                                            unsigned long long* __restrict__ total_out,
                                                            float4* __restrict__ setup, uint32_t* __restrict__ status,
                                                            unsigned long long* __restrict__ total_host /* pinned, or nullptr */) {
    __shared__ uint32_t wsum[kCountBlock / 64];
    __shared__ uint32_t wtall[kCountBlock / 64];
    __shared__ unsigned long long base_s[1];
    __shared__ uint32_t s_next;
    __shared__ float4 stash[7 * kCountBlock];      // 28 KB: positions, texture coordinates and geometry-stage output of the block's triangles (count_block_a)
    const uint32_t n_tb = (sc.n_tri + (uint32_t)kCountBlock - 1u) / (uint32_t)kCountBlock;
    const uint32_t G = gridDim.x, n_extra = n_tb - G;            // (the launcher: G == n_tb, or the resident workgroups if the rest is at most an eighth of them: one extra block each, at most)
    // tickets: word 2 of the tall-triangle table's header (zero when allocated; k_emit2 / the launcher of a counting-only conversion zero it again)
    uint32_t* const tk = tall_header(setup, sc.n_tri) + 2;
    // (straight-line code, at most ONE extra block per workgroup: as a loop over tickets the kernel keeps its scene pointers live
    // across the back edge and spills 128 bytes per lane — what rounds 5 and 6 ran into with their persistent forms)
    uint32_t c1, tot1, c2 = 0, tot2 = 0;
    unsigned long long loc1, loc2 = 0;
    count_block_a(sc, R, off, chain, epoch, setup, blockIdx.x, wsum, wtall, stash, c1, loc1, tot1);
    uint32_t extra = 0xFFFFFFFFu;
    if (n_extra != 0) {
        __syncthreads();                       // every wave is done with the shared words of the first block
        if (threadIdx.x == 0) {
            const uint32_t e = atomicAdd(tk, 1u);
            s_next = e < n_extra ? G + e : 0xFFFFFFFFu;
        }
        __syncthreads();
        extra = s_next;
        if (extra != 0xFFFFFFFFu) count_block_a(sc, R, off, chain, epoch, setup, extra, wsum, wtall, stash, c2, loc2, tot2);
    }
    count_block_b(sc, off, start, n_start, chain, epoch, total_out, status, total_host, blockIdx.x, n_tb, base_s, c1, loc1, tot1);
    if (extra != 0xFFFFFFFFu) {
        __syncthreads();                       // base_s of the first block has been read by every wave
        count_block_b(sc, off, start, n_start, chain, epoch, total_out, status, total_host, extra, n_tb, base_s, c2, loc2, tot2);
    }
}

// ============================================================================================
// k_emit2
// ============================================================================================
struct Emit2Lds {
    float4 tri[64 * 5];          // TriShade of the current batch of (up to) 64 triangles
    uint32_t entries[kSlice];    // slot << 24 | y << 12 | x, indexed by (record index - slice base)
    float4 stage[32 * 6];        // half-wave record staging; during the expansion its first 64 bytes hold the row-start mask
                                 // of the slice (one bit per record, row_mask below) — the two uses never overlap in time
    __device__ __forceinline__ uint32_t* row_mask() { return reinterpret_cast<uint32_t*>(stage); }
};
static_assert(sizeof(Emit2Lds) == 10240, "10 KB per wave: four workgroups of four waves per CU");
// a fine block's workgroup: the TriShade of its 256 triangles, its whole entry list, one staging area per wave — the same 40 KB
struct FineLds {
    float4 tri[kCountBlock * 5];
    uint32_t entries[kFineMax];  // thread << 24 | y << 12 | x, indexed by (record index - block base)
    float4 stage[kBlock / 64][32 * 6];
};
static_assert(sizeof(FineLds) == sizeof(Emit2Lds) * (kBlock / 64), "both kinds of workgroup of k_emit2 use the same LDS");
static_assert(kCountBlock == kBlock, "a fine block is emitted by one thread per triangle");

// One strip: the fragments whose entries are strip[0 .. n) (slot << 24 | y << 12 | x; slot = index into `tri`, triangle t_base +
// slot) are shaded and written to dst[0 .. n) — staged half a wave at a time, 16 B per lane, non-temporal.
__device__ __forceinline__ void shade_and_store_strip(const SceneDev& sc, const float4* tri, const uint32_t* strip, uint32_t n, uint32_t t_base,
                                                      float4* stage, float4* __restrict__ dst, int lane) {
    const bool have = (uint32_t)lane < n;
    uint32_t en = 0;
    if (have) en = strip[lane];
    const uint32_t tl = en >> 24;
    uint32_t my_mesh = 0;
    if (have) my_mesh = reinterpret_cast<const uint32_t*>(&tri[tl * 5 + 4])[3] & 0xFFFFFFu;
    const uint32_t m_first = __builtin_amdgcn_readfirstlane(my_mesh);   // lane 0 always holds a fragment
    const bool uniform = sc.n_meshes == 1 || __ballot(have && my_mesh != m_first) == 0ull;
    float4 rec[6];
    if (have) {
        const TriShade& ts = *reinterpret_cast<const TriShade*>(&tri[tl * 5]);
        const uint32_t tt = t_base + tl;
        if (uniform) shade_from_tri(sc.tri, tt, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), kConstMesh(sc.meshes + m_first), ts, rec);
        else shade_from_tri(sc.tri, tt, (int)(en & 0xFFFu), (int)((en >> 12) & 0xFFFu), sc.meshes + my_mesh, ts, rec);
    }
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        if (have && (lane >> 5) == half) {
#pragma unroll
            for (int k = 0; k < 6; ++k) stage[(lane & 31) * 6 + k] = rec[k];
        }
        wave_lds_sync();
        float4* __restrict__ dsto = dst + 32u * half * 6u;
        const uint32_t nv = n > 32u * half ? min(32u, n - 32u * half) : 0u;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const uint32_t q = (uint32_t)lane + 64u * j;
            const uint32_t r = q / 6u;
            if (r < nv) nt_store(&dsto[q], stage[q]);
        }
        wave_lds_sync();
    }
}

// A fine block (at most kFineMax fragments from kCountBlock triangles of at most kRowsThread rows each): one thread per triangle
// writes its pixels into the block's entry list, then the four waves take the strips in turn.
__device__ __forceinline__ void emit_fine_block(const SceneDev& sc, const uint32_t* __restrict__ off, unsigned long long nw,
                                                const float4* __restrict__ setup, float4* __restrict__ out, uint32_t blk, FineLds& F) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t T = sc.n_tri, t0 = blk * (uint32_t)kCountBlock, t1 = min(t0 + (uint32_t)kCountBlock, T);
    const uint32_t base = off[t0], end = off[t1];
    if (end <= base || (unsigned long long)base >= nw) return;
    const uint32_t n_store = (uint32_t)min((unsigned long long)(end - base), nw - base);
    const uint32_t t = t0 + threadIdx.x;
    if (t < t1) {
        const uint32_t o0 = off[t], o1 = off[t + 1];
        if (o1 > o0 && o0 - base < n_store) {
            TriSetup s;
            const float4* src4 = setup + (size_t)t * 7;
            float4* dst4 = reinterpret_cast<float4*>(&s);
#pragma unroll
            for (int k = 0; k < 7; ++k) dst4[k] = src4[k];
#pragma unroll
            for (int k = 0; k < 5; ++k) F.tri[threadIdx.x * 5 + k] = dst4[k];
            const Raster rs = raster_from_setup(s);
            const uint32_t tag = (uint32_t)threadIdx.x << 24;
            uint32_t k = o0 - base;
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            for (int y = rs.y0; y <= rs.y1 && k < n_store; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                for (int x = xa; x <= xb && k < n_store; ++x, ++k) F.entries[k] = tag | ((uint32_t)y << 12) | (uint32_t)x;
            }
        }
    }
    __syncthreads();
    for (uint32_t s0 = wave * 64u; s0 < n_store; s0 += (uint32_t)kBlock)
        shade_and_store_strip(sc, F.tri, &F.entries[s0], min(64u, n_store - s0), t0, F.stage[wave], out + ((size_t)base + s0) * 6, lane);
}

__global__ void __launch_bounds__(kBlock, M2S_EMIT2_WAVES) k_emit2(SceneDev sc, uint32_t R, const uint32_t* __restrict__ off,
                                                     const uint32_t* __restrict__ start,
                                                     const unsigned long long* __restrict__ total_p, unsigned long long limit,
                                                     const float4* __restrict__ setup, float4* __restrict__ out,
                                                     uint32_t run /* consecutive workgroups per XCD turn */) {
    __shared__ float4 lds_raw[sizeof(FineLds) / sizeof(float4)];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long total = *total_p;
    const unsigned long long nw = total < limit ? total : limit;  // records actually stored
    const uint32_t T = sc.n_tri;
    const uint8_t* __restrict__ cls = block_class(setup, T);
    // the first workgroups of the launch (one per block of triangles, rounded up to whole groups of eight) take the FINE blocks
    const uint32_t n_tb = (T + (uint32_t)kCountBlock - 1u) / (uint32_t)kCountBlock, n_fine_wg = (n_tb + 7u) & ~7u;
#ifdef M2S_TIMELINE
    TlEmitScope tl(blockIdx.x, wave, lane);
    tl.kind = blockIdx.x < n_fine_wg ? ((blockIdx.x < n_tb && cls[blockIdx.x]) ? 1 : 2) : 3;
#endif
    if (blockIdx.x < n_fine_wg) {
        if (blockIdx.x < n_tb && cls[blockIdx.x]) emit_fine_block(sc, off, nw, setup, out, blockIdx.x, *reinterpret_cast<FineLds*>(lds_raw));
        return;
    }
    const uint32_t bid = blockIdx.x - n_fine_wg;
    Emit2Lds& L = reinterpret_cast<Emit2Lds*>(lds_raw)[wave];
    // XCD-aware mapping (hardware workgroup b runs on XCD b % 8, private L2 each): the XCDs take turns of `run` consecutive
    // workgroups — runs of the output, of the mesh surface, of texture space meet in ONE L2 —, block-cyclically.  Until round 3
    // every XCD had one contiguous EIGHTH of the output: fine for one uniform mesh, but a record does not cost the same
    // everywhere (triangles per slice, magnified or minified maps): on the C4 stand-in the kernel ran 7 % faster with NO mapping
    // at all (plain round-robin).  Turns keep the locality and spread the expensive regions over all XCDs; k_emit2 has no
    // inter-workgroup dependency, so any mapping is correct.
    // the tall-triangle table's slot counter goes back to zero for the next conversion (k_emit2 itself only reads the table)
    if (bid == 0 && threadIdx.x == 0) { tall_header(setup, T)[0] = 0; tall_header(setup, T)[2] = 0; }   // (... and k_count_scan's tickets)
    const uint32_t per_wg = kSlice * (kBlock / 64);
    const uint32_t nblk = (uint32_t)((nw + per_wg - 1) / per_wg);
    const uint32_t xcd = bid & 7u, turn = (bid >> 3) / run, in_run = (bid >> 3) % run;
    const uint32_t lblock = (turn * 8u + xcd) * run + in_run;
    if (lblock >= nblk) return;
    const uint32_t slice = lblock * (kBlock / 64) + wave;
    const unsigned long long wbase64 = (unsigned long long)slice * kSlice;
    if (wbase64 >= nw) return;
    const uint32_t wbase = (uint32_t)wbase64;
    const uint32_t wend = (uint32_t)(nw - wbase64 < (unsigned long long)kSlice ? nw : wbase64 + kSlice);

    uint32_t pos = wbase;                 // next record to produce
    for (uint32_t t_cur = start[slice]; pos < wend && t_cur < T; ) {
        // ---- the batch: up to 64 consecutive triangles, one per lane.  A fine block is stepped over; a batch that would run from a
        // dense block into a fine one ends at the block boundary.  Everything the decision needs is requested at once (the classes of
        // this block and the next, the offsets at both possible ends, the lanes' own offsets): one round trip, as before round 5 ----
#ifdef M2S_TIMELINE
        tl.nb++;
#endif
        const uint32_t blk = t_cur / (uint32_t)kCountBlock;
        const uint32_t blk_end = min((blk + 1u) * (uint32_t)kCountBlock, T);
        const uint32_t t_full = min(t_cur + 64u, T);
        const uint32_t t = t_cur + lane;
        uint32_t o0 = 0xFFFFFFFFu, o1 = 0xFFFFFFFFu;
        if (t < t_full) { o0 = off[t]; o1 = off[t + 1]; }
        const uint32_t c_here = cls[blk], c_next = t_full > blk_end ? cls[blk + 1u] : 0u;
        const uint32_t o_full = off[t_full], o_blk = off[blk_end];
        const bool fine_blk = c_here != 0;
        const uint32_t t_next = (fine_blk || c_next != 0) ? blk_end : t_full;
        const uint32_t o_next = t_next == t_full ? o_full : o_blk;   // first record of the next batch
        const uint32_t bend = min(wend, o_next);             // records of this batch: [pos, bend)
        if (fine_blk) { pos = bend; t_cur = t_next; continue; }   // (those records are emit_fine_block's)
        const bool active = (t < t_next) && (o1 > o0) && (o0 < bend) && (o1 > pos);
        Raster rs;
        rs.x0 = rs.y0 = 0; rs.x1 = rs.y1 = -1; rs.bias = 0; rs.area2 = 1; rs.ext = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) { rs.a[i] = rs.b[i] = 0; rs.c[i] = 0; }
        uint32_t tall_slot = 0;
        if (active) {
            TriSetup s;
            const float4* src4 = setup + (size_t)t * 7;
            float4* dst4 = reinterpret_cast<float4*>(&s);
#pragma unroll
            for (int k = 0; k < 7; ++k) dst4[k] = src4[k];
#pragma unroll
            for (int k = 0; k < 5; ++k) L.tri[lane * 5 + k] = dst4[k];
            rs = raster_from_setup(s);
            tall_slot = s.tall;
        }
        // ---- expansion: (triangle slot, pixel) entries of [pos, bend), in canonical order ----
        // Two steps (round 5).  (1) Row starts: whoever walks a triangle's rows — its own lane, or the whole wave for a tall one —
        // writes ONE entry per covered row, at the row's first record inside [pos, bend), and sets that record's bit in a 512-bit
        // mask.  (2) Fill: all 64 lanes sweep the range; a record's entry is the nearest row start at or before it plus the distance
        // (in x).  Until round 5 the lane that owned a triangle wrote every one of its pixels itself: a slice inside triangles of
        // 100-300 fragments (columns, panels of a Sponza-like scene) kept 2-5 lanes busy for hundreds of iterations while the other
        // sixty waited — as long as the eight strips of shading that followed.
        const uint32_t tag = (uint32_t)lane << 24;
        const int rows = active ? rs.y1 - rs.y0 + 1 : 0;
        uint32_t* const smask = L.row_mask();
        if (lane < kSlice / 32) smask[lane] = 0;
        wave_lds_sync();
        auto mark_row = [&](uint32_t k_row, uint32_t len, uint32_t tag_y, int xa) {   // row = records [k_row, k_row + len), first pixel xa
            if (len != 0 && k_row < bend && k_row + len > pos) {
                const uint32_t st = max(k_row, pos);
                const uint32_t i = st - wbase;
                L.entries[i] = tag_y | (uint32_t)(xa + (int)(st - k_row));
                atomicOr(&smask[i >> 5], 1u << (i & 31u));
            }
        };
        if (active && rows <= kRowsThread) {
            uint32_t k = o0;
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            for (int y = rs.y0; y <= rs.y1 && k < bend; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                const uint32_t len = (uint32_t)max(xb - xa + 1, 0);
                mark_row(k, len, tag | ((uint32_t)y << 12), xa);
                k += len;
            }
        }
        unsigned long long big = __ballot(active && rows > kRowsThread);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const Raster b = shfl_raster(rs, src);
            const uint32_t btag = (uint32_t)src << 24;
            uint32_t acc = __shfl(o0, src);
            int yc0 = b.y0;
            const uint32_t ts_ = (uint32_t)__shfl((int)tall_slot, src);
            if (ts_ != 0 && pos > acc) {
                // start at the 64-row chunk that holds record `pos`: the last chunk whose running sum does not exceed pos - acc
                // (running sums never decrease; a chunk without fragments shares its sum with the next one and is skipped)
                const uint32_t rel = pos - acc;
                const int nch = min((b.y1 - b.y0) / 64 + 1, (int)kTallChunks);
                const uint32_t* row = tall_header(setup, T) + 4 + (size_t)(ts_ - 1u) * kTallChunks;
                const uint32_t pre = lane < nch ? row[lane] : 0xFFFFFFFFu;
                const unsigned long long le = __ballot(pre <= rel);
                const int c0 = le ? 63 - __clzll((long long)le) : 0;
                acc += (uint32_t)__shfl((int)pre, c0);
                yc0 += 64 * c0;
            }
            for (int yc = yc0; yc <= b.y1 && acc < bend; yc += 64) {
                const int y = yc + lane;
                int xa = 0, xb = -1;
                if (y <= b.y1) row_span(b, y, xa, xb);
                const uint32_t len = (uint32_t)max(xb - xa + 1, 0);
                const uint32_t incl = wave_incl_scan(len, lane);
                const uint32_t chunk = __shfl(incl, 63);
                if (acc + chunk > pos) mark_row(acc + (incl - len), len, btag | ((uint32_t)y << 12), xa);
                acc += chunk;
            }
        }
        wave_lds_sync();
        {   // fill: every record of [pos, bend) lies in a marked row (the row that holds `pos` was clipped to start AT pos)
            const uint32_t i0 = pos - wbase, i1 = bend - wbase;
            uint32_t carry = 0;
            for (uint32_t blk = i0 >> 6; blk <= (i1 - 1u) >> 6; ++blk) {
                const uint32_t i = blk * 64u + (uint32_t)lane;
                const uint32_t raw = L.entries[i];
                const unsigned long long m = (unsigned long long)smask[2u * blk] | ((unsigned long long)smask[2u * blk + 1u] << 32);
                const unsigned long long me = m & (~0ull >> (63 - lane));        // row starts at or before my record, in this block
                const int srcl = me ? 63 - __clzll((long long)me) : lane;
                const uint32_t from = (uint32_t)__shfl((int)raw, srcl);
                const uint32_t e = me ? from + (uint32_t)(lane - srcl) : carry + (uint32_t)lane + 1u;
                if (i >= i0 && i < i1) L.entries[i] = e;
                carry = (uint32_t)__builtin_amdgcn_readlane((int)e, 63);
            }
        }
        wave_lds_sync();
        // ---- fragment phase: strips of 64 entries ----
        for (uint32_t s0 = pos; s0 < bend; s0 += 64)
            shade_and_store_strip(sc, L.tri, &L.entries[s0 - wbase], min(64u, bend - s0), t_cur, L.stage, out + (size_t)s0 * 6, lane);
        pos = bend;
        t_cur = t_next;
    }
}

// ---- launchers ---------------------------------------------------------------------------------------------------
uint32_t emit2_slices(uint64_t limit) { return (uint32_t)((limit + kSlice - 1) / kSlice); }
uint32_t count_scan_blocks(uint32_t n_tri) { return (n_tri + kCountBlock - 1) / kCountBlock; }
size_t setup_bytes(uint32_t n_tri) { return setup_tall_offset(n_tri) + 16 + (size_t)kTallCap * kTallChunks * sizeof(uint32_t) + ((count_scan_blocks(n_tri) + 63u) & ~63u); }
size_t setup_tall_offset(uint32_t n_tri) { return (size_t)std::max<uint32_t>(n_tri, 1u) * sizeof(TriSetup); }

void launch_count_scan(const SceneDev& sc, uint32_t R, uint32_t* off, uint32_t* start, uint32_t n_start, unsigned long long* chain,
                       uint32_t epoch, unsigned long long* total, void* setup, uint32_t* status, unsigned long long* total_host, hipStream_t st) {
    if (!sc.n_tri) return;
    uint32_t grid = count_scan_blocks(sc.n_tri);
    // more blocks than resident workgroups, but few enough for tickets: the launch is the resident set (see count_block_a)
    static const uint32_t resident = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return (uint32_t)cus * (uint32_t)M2S_COUNT_WAVES;      // kCountBlock / 64 = 4 waves per workgroup, M2S_COUNT_WAVES per SIMD, 4 SIMDs
    }();
    uint32_t res = resident;
    if (const char* v = debug_env("M2S_COUNT_RESIDENT")) res = (uint32_t)strtoul(v, nullptr, 10);   // debug: 0 = one workgroup per block, always
    // ... and few enough that most workgroups leave after their own block whatever happens: a workgroup with an extra block waits, in that
    // block's look-back, for workgroups dispatched AFTER it — if the GPU is shared (the other lane's kernels, another process) not all of
    // the launch is resident at once, and the workgroups without an extra block are the ones that make room
    if (res && grid > res && grid - res <= res / 8u) grid = res;
    hipLaunchKernelGGL(k_count_scan, dim3(grid), dim3(kCountBlock), 0, st, sc, R, off, start, n_start, chain,
                       epoch & 0xFFFFu, total, (float4*)setup, status, total_host);
}

void launch_emit2(const SceneDev& sc, uint32_t R, const uint32_t* off, const uint32_t* start, const unsigned long long* total,
                  uint64_t limit, const void* setup, float4* out, hipStream_t st) {
    if (!sc.n_tri) return;
    if (!limit) {   // a counting-only conversion: nothing to emit, but the tall-triangle table's slot counter still goes back to zero
        (void)hipMemsetAsync((char*)const_cast<void*>(setup) + setup_tall_offset(sc.n_tri), 0, 16, st);
        return;
    }
    const uint32_t per_wg = kSlice * (kBlock / 64);
    uint32_t n_blocks = (uint32_t)((limit + per_wg - 1) / per_wg);
    uint32_t run = 16;                          // workgroups per XCD turn (profiles/r03/ab_emit2_xcd_turns.log)
    if (const char* v = debug_env("M2S_EMIT2_RUN")) { const unsigned long r = strtoul(v, nullptr, 10); if (r >= 1 && r <= 65536) run = (uint32_t)r; }   // debug
    n_blocks = (n_blocks + 8u * run - 1u) / (8u * run) * (8u * run);   // whole rounds of turns; surplus workgroups leave at once
    n_blocks += (count_scan_blocks(sc.n_tri) + 7u) & ~7u;              // in front of them: one workgroup per block of triangles (the fine blocks)
    hipLaunchKernelGGL(k_emit2, dim3(n_blocks), dim3(kBlock), 0, st, sc, R, off, start, total, (unsigned long long)limit,
                       (const float4*)setup, out, run);
}

// The multi-pass kernels keep a few spilled registers in scratch memory (12-32 bytes per lane), and the runtime sets a queue's scratch
// up inside the FIRST dispatch that needs it (~0.1 ms, seen on k_fused3 when a stack crept into it).  warm_scene pays that at upload for
// scenes AUTO sends here: one workgroup of a kernel that uses 64 bytes per lane.
__global__ void __launch_bounds__(64) k_scratch_warm(uint32_t* sink) {
    volatile uint32_t a[16];
#pragma unroll 1
    for (int i = 0; i < 16; ++i) a[i] = (uint32_t)i + threadIdx.x;
    if (sink && a[threadIdx.x & 15u] == 0xFFFFFFFFu) *sink = 1u;
}
void launch_scratch_warm(hipStream_t st) { hipLaunchKernelGGL(k_scratch_warm, dim3(1), dim3(64), 0, st, (uint32_t*)nullptr); }

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_multipass() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_emit2)); }

}  // namespace m2s

#ifdef M2S_TIMELINE
extern "C" int m2s_debug_timeline(void* count, size_t count_bytes, void* emit, size_t emit_bytes) {
    if (hipMemcpyFromSymbol(count, HIP_SYMBOL(m2s::g_tl_count), std::min(count_bytes, sizeof(m2s::g_tl_count))) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(emit, HIP_SYMBOL(m2s::g_tl_emit), std::min(emit_bytes, sizeof(m2s::g_tl_emit))) != hipSuccess) return 2;
    return 0;
}
extern "C" int m2s_debug_timeline_clear() {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(m2s::g_tl_emit)) != hipSuccess) return 1;
    return hipMemset(p, 0, sizeof(m2s::g_tl_emit)) == hipSuccess ? 0 : 2;
}
#endif
