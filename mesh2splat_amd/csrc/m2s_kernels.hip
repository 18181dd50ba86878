// m2s_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the mesh -> 3DGS
// conversion pass.  Replaces the reference's VS+GS+rasteriser+FS program
//   src/shaders/conversion/converterVS.glsl, converterGS.glsl:326-443, converterFS.glsl:44-104
// driven by ConversionPass::execute (src/renderer/renderPasses/ConversionPass.cpp:9-117).
//
// What lives here: the upload-time kernels (repack, mips, combo textures, mesh table), the exact count the upload takes
// (k_count + k_scan_partials: AUTO's decision, run tables, m2s_download_triangle_counts) with its by-products (k_big_share,
// k_unit_bases, k_run_order), and k_emit_big, the second stage of the single-pass kernels for triangles they defer.  The
// conversion kernels proper are m2s_fused*.hip, m2s_sparse.hip (single pass) and m2s_emit2.hip (multi-pass); the first-
// generation multi-pass emission that started here in round 1 (k_offsets, k_emit) was removed in round 6 (tag r5-final has it).
// Output order is deterministic: (mesh, triangle, pixel row, pixel column) — the reference's order
// is atomic-arrival order (converterFS.glsl:46), i.e. unspecified.
//
// Arithmetic is fp32 with one rounding per operation (compiled with -ffp-contract=off; HIP's
// default correctly-rounded fp32 divide/sqrt), integer rasterisation on a 24.8 grid with int64
// edge functions: see DESIGN.md "Pinned semantics".  No MFMA: nothing here is a contraction.
#include "m2s_devfn.h"

#pragma clang fp contract(off)

namespace m2s {

// ============================================================================================
// upload-time kernels (== SceneManager::setupMeshBuffers / glGenerateMipmap; not in the timed pass)
// ============================================================================================
struct TriPlanesW {
    float4* A0; float4* A1; float* A2; float4* B0; float2* B1; float4* C0; float4* C1; float* C2;
    float4* D0; float4* D1; float4* D2;
};

__global__ void __launch_bounds__(kBlock) k_repack(const float* __restrict__ aos, uint32_t stride, uint32_t src_first,
                                                   uint32_t n, uint32_t dst_first, TriPlanesW d) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float* v0 = aos + (size_t)(src_first + i) * 3 * stride;
    const float* v1 = v0 + stride;
    const float* v2 = v1 + stride;
    uint32_t t = dst_first + i;
    // vertex layout: pos 0-2, normal 3-5, tangent 6-9, uv 10-11 (converterVS.glsl:9-12)
    d.A0[t] = make_float4(v0[0], v0[1], v0[2], v1[0]);
    d.A1[t] = make_float4(v1[1], v1[2], v2[0], v2[1]);
    d.A2[t] = v2[2];
    d.B0[t] = make_float4(v0[10], v0[11], v1[10], v1[11]);
    d.B1[t] = make_float2(v2[10], v2[11]);
    d.C0[t] = make_float4(v0[3], v0[4], v0[5], v1[3]);
    d.C1[t] = make_float4(v1[4], v1[5], v2[3], v2[4]);
    d.C2[t] = v2[5];
    d.D0[t] = make_float4(v0[6], v0[7], v0[8], v0[9]);
    d.D1[t] = make_float4(v1[6], v1[7], v1[8], v1[9]);
    d.D2[t] = make_float4(v2[6], v2[7], v2[8], v2[9]);
}

void launch_repack(const float* d_aos, uint32_t stride_floats, uint32_t /*n_tri_src*/, uint32_t src_first, uint32_t n,
                   uint32_t dst_first, TriPlanes dst, hipStream_t st) {
    if (!n) return;
    TriPlanesW w;
    w.A0 = const_cast<float4*>(dst.A0); w.A1 = const_cast<float4*>(dst.A1); w.A2 = const_cast<float*>(dst.A2);
    w.B0 = const_cast<float4*>(dst.B0); w.B1 = const_cast<float2*>(dst.B1);
    w.C0 = const_cast<float4*>(dst.C0); w.C1 = const_cast<float4*>(dst.C1); w.C2 = const_cast<float*>(dst.C2);
    w.D0 = const_cast<float4*>(dst.D0); w.D1 = const_cast<float4*>(dst.D1); w.D2 = const_cast<float4*>(dst.D2);
    hipLaunchKernelGGL(k_repack, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, d_aos, stride_floats, src_first, n,
                       dst_first, w);
}

// glGenerateMipmap pinned as a 2x2 box filter, round-half-up, floor dimensions (odd tail dropped)
__global__ void __launch_bounds__(kBlock) k_mip(const uint32_t* __restrict__ src, uint32_t sw, uint32_t sh,
                                                uint32_t* __restrict__ dst, uint32_t dw, uint32_t dh) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= dw * dh) return;
    uint32_t x = i % dw, y = i / dw;
    uint32_t x0 = min(2 * x, sw - 1), x1 = min(2 * x + 1, sw - 1);
    uint32_t y0 = min(2 * y, sh - 1), y1 = min(2 * y + 1, sh - 1);
    uint32_t t00 = src[(size_t)y0 * sw + x0], t10 = src[(size_t)y0 * sw + x1];
    uint32_t t01 = src[(size_t)y1 * sw + x0], t11 = src[(size_t)y1 * sw + x1];
    uint32_t o = 0;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        uint32_t s = ((t00 >> (8 * ch)) & 255u) + ((t10 >> (8 * ch)) & 255u) + ((t01 >> (8 * ch)) & 255u) +
                     ((t11 >> (8 * ch)) & 255u);
        o |= ((s + 2u) >> 2) << (8 * ch);
    }
    dst[i] = o;
}

// interleave one mip level of the three maps into the combo layout (see ComboDesc)
__global__ void __launch_bounds__(kBlock) k_combo(const uint32_t* __restrict__ a, const uint32_t* __restrict__ n,
                                                  const uint32_t* __restrict__ m, uint32_t w, uint32_t h,
                                                  uint32_t* __restrict__ dst) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= (w + 1) * h) return;
    const uint32_t x = i % (w + 1), y = i / (w + 1);
    const uint32_t src = y * w + (x == w ? 0u : x);
    dst[3 * i + 0] = a[src];
    dst[3 * i + 1] = n[src];
    dst[3 * i + 2] = m[src];
}

void launch_combo_level(const uint32_t* a, const uint32_t* n, const uint32_t* m, uint32_t w, uint32_t h, uint32_t* dst,
                        hipStream_t st) {
    const uint32_t cnt = (w + 1) * h;
    hipLaunchKernelGGL(k_combo, dim3((cnt + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a, n, m, w, h, dst);
}

// mesh_of8[k] = {mesh of local triangle 8 k, local index one past the last triangle of that mesh (clamped to the shard)}
__global__ void __launch_bounds__(kBlock) k_mesh_table(SceneDev sc, uint2* __restrict__ table) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k * 8u >= sc.n_tri) return;
    const uint32_t m = find_mesh(sc, sc.tri_first + k * 8u);
    const unsigned long long end = (unsigned long long)sc.mesh_first[m + 1] - sc.tri_first;
    table[k] = make_uint2(m, (uint32_t)(end < sc.n_tri ? end : sc.n_tri));
}
void launch_mesh_table(const SceneDev& sc, uint2* table, hipStream_t st) {
    const uint32_t n = (sc.n_tri + 7u) / 8u;
    if (n) hipLaunchKernelGGL(k_mesh_table, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, sc, table);
}

void launch_mip_level(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh,
                      hipStream_t st) {
    uint32_t n = dw * dh;
    hipLaunchKernelGGL(k_mip, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, src, sw, sh, dst, dw, dh);
}

// ============================================================================================
// K1: per-triangle fragment count
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_count(SceneDev sc, uint32_t R, uint32_t* __restrict__ cnt,
                                                  uint32_t* __restrict__ partials) {
    __shared__ uint32_t red[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = blockIdx.x * kTriPerBlock;
    const uint32_t lastT = min(blockBase + kTriPerBlock, sc.n_tri) - 1;
    bool uniform_mesh;
    const uint32_t m0 = mesh_of_range(sc, blockBase, lastT, uniform_mesh);   // one scalar load (was: a binary search)
    uint32_t sum = 0;
    for (int it = 0; it < kTriPerBlock / kBlock; ++it) {
        const uint32_t t = blockBase + it * kBlock + threadIdx.x;
        const bool valid = t < sc.n_tri;
        Raster rs;
        bool ok = false;
        if (valid) ok = setup_raster_for(sc, t, m0, uniform_mesh, R, rs);
        const int rows = ok ? rs.y1 - rs.y0 + 1 : 0;
        uint32_t c = 0;
        if (ok && rows <= kRowsCount) {
            RowWalker rw;
            row_walker_init(rs, rs.y0, rw);
            for (int y = rs.y0; y <= rs.y1; ++y) {
                int xa, xb;
                row_walker_next(rw, xa, xb);
                c += (uint32_t)max(xb - xa + 1, 0);
            }
        }
        // triangles spanning more rows: the whole wave counts one triangle, one row per lane
        unsigned long long big = __ballot(ok && rows > kRowsCount);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const Raster b = shfl_raster(rs, src);
            uint32_t part = 0;
            for (int y = b.y0 + lane; y <= b.y1; y += 64) {
                int xa, xb;
                row_span(b, y, xa, xb);
                part += (uint32_t)max(xb - xa + 1, 0);
            }
            part = wave_sum(part);
            if (lane == src) c = part;
        }
        if (valid) cnt[t] = c;
        sum += c;
    }
    sum = wave_sum(sum);
    if (lane == 0) red[wave] = sum;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

void launch_count(const SceneDev& sc, uint32_t R, uint32_t* cnt, uint32_t* partials, hipStream_t st) {
    if (!sc.n_tri) return;
    hipLaunchKernelGGL(k_count, dim3(n_count_blocks(sc.n_tri)), dim3(kBlock), 0, st, sc, R, cnt, partials);
}

// ============================================================================================
// K_scan: exclusive scan of the partial sums (single workgroup), total -> *total (u64)
// ============================================================================================
__global__ void __launch_bounds__(1024) k_scan_partials(uint32_t* __restrict__ partials, uint32_t n,
                                                        unsigned long long* __restrict__ total) {
    __shared__ uint32_t wsum[16];
    __shared__ unsigned long long carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < n ? partials[i] : 0;
        uint32_t incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w = 0; w < wave; ++w) woff += wsum[w];
        const unsigned long long carry = carry_s;
        if (i < n) partials[i] = (uint32_t)(carry + woff + incl - v);
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

void launch_scan_partials(uint32_t* partials, uint32_t n_partials, unsigned long long* total, hipStream_t st) {
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, st, partials, n_partials, total);
}

// fragments held by the triangles of more than `threshold` fragments: what a single-pass kernel would have to defer (warm_scene)
__global__ void __launch_bounds__(kBlock) k_big_share(const uint32_t* __restrict__ cnt, uint32_t n, uint32_t threshold, unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const uint32_t c = cnt[i];
        if (c > threshold) acc += c;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}
void launch_big_share(const uint32_t* cnt, uint32_t n_tri, uint32_t threshold, unsigned long long* out, hipStream_t st) {
    if (!n_tri) return;
    hipLaunchKernelGGL(k_big_share, dim3(std::min<uint32_t>((n_tri + kBlock - 1) / kBlock, 1024u)), dim3(kBlock), 0, st, cnt, n_tri, threshold, out);
}

// ============================================================================================
// K_unit_bases: where the output of every RUN of units starts (unit = 256 or 512 triangles, run = 1 << shift units), from the
// exact counts and the scanned partial sums — what a launch of the single-pass kernels without runs records as a by-product
// (RunInfo::out), here from the count m2s_upload_scene takes, so that the FIRST conversion of a scene already runs in runs.
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_unit_bases(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ partials,
                                                       uint32_t n_tri, uint32_t unit, uint32_t shift, unsigned long long* __restrict__ run_base) {
    __shared__ uint32_t red[kTriPerBlock / kBlock][kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = blockIdx.x * kTriPerBlock;
#pragma unroll
    for (int it = 0; it < kTriPerBlock / kBlock; ++it) {
        const uint32_t t = blockBase + it * kBlock + threadIdx.x;
        const uint32_t s = wave_sum(t < n_tri ? cnt[t] : 0u);
        if (lane == 0) red[it][wave] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long run = partials[blockIdx.x];
        const uint32_t tri_per_run = unit << shift;                // (a multiple of 256: runs start on 256-triangle boundaries)
        for (int it = 0; it < kTriPerBlock / kBlock; ++it) {
            const uint32_t t = blockBase + it * kBlock;
            if (t < n_tri && t % tri_per_run == 0) run_base[t / tri_per_run] = run;
            run += red[it][0] + red[it][1] + red[it][2] + red[it][3];
        }
    }
}

void launch_unit_bases(const uint32_t* cnt, const uint32_t* partials, uint32_t n_tri, uint32_t unit, uint32_t shift, unsigned long long* run_base, hipStream_t st) {
    if (!n_tri) return;
    hipLaunchKernelGGL(k_unit_bases, dim3(n_count_blocks(n_tri)), dim3(kBlock), 0, st, cnt, partials, n_tri, unit, shift, run_base);
}

// ============================================================================================
// K_run_order: rank sort of the runs by fragment count, descending (ties: lower run first).  n_runs <= a few thousand.
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_run_order(const unsigned long long* __restrict__ run_base, uint32_t n_runs,
                                                      const unsigned long long* __restrict__ total, uint32_t* __restrict__ order, uint32_t n_slots) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_slots) return;
    if (i >= n_runs) { order[i] = i; return; }
    const unsigned long long end = *total;
    auto cost = [&](uint32_t j) { return (j + 1u < n_runs ? run_base[j + 1u] : end) - run_base[j]; };
    const unsigned long long mine = cost(i);
    uint32_t rank = 0;
    for (uint32_t j = 0; j < n_runs; ++j) {
        const unsigned long long cj = cost(j);
        rank += (cj > mine || (cj == mine && j < i)) ? 1u : 0u;
    }
    order[rank] = i;
}
void launch_run_order(const unsigned long long* run_base, uint32_t n_runs, const unsigned long long* total, uint32_t* order, uint32_t n_slots, hipStream_t st) {
    if (!n_slots) return;
    hipLaunchKernelGGL(k_run_order, dim3((n_slots + kBlock - 1) / kBlock), dim3(kBlock), 0, st, run_base, n_runs, total, order, n_slots);
}

// ============================================================================================
// k_emit_big: the second stage of the AUTO pipeline.  One workgroup per (deferred triangle, chunk of
// kEmitF fragments): per-row spans of the whole triangle -> LDS prefix -> the chunk's fragments located
// by binary search -> the same per-fragment code and staged stores as k_emit.  A 4096 x 4096 px triangle
// becomes 8192 equal work items.
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_emit_big(SceneDev sc, uint32_t R, const BigItem* __restrict__ list, uint32_t n_big,
                                                     unsigned long long limit, float4* __restrict__ out) {
    __shared__ uint2 entries[kEmitF];
    __shared__ float4 stage[kBlock * kStageStride];
    __shared__ uint32_t rowoff[4096 + 1];   // exclusive prefix of the row lengths (R <= 4096 rows)
    __shared__ uint16_t rowxa[4096];
    __shared__ uint32_t wsum[kBlock / 64];
    if (blockIdx.y >= n_big) return;
    const BigItem item = list[blockIdx.y];
    const uint32_t f0 = blockIdx.x * kEmitF;
    if (f0 >= item.cnt) return;
    const uint32_t f1 = min(f0 + (uint32_t)kEmitF, item.cnt);
    if (item.off + f0 >= limit) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t m = find_mesh(sc, sc.tri_first + item.t);
    Raster rs;
    if (!setup_raster_for(sc, item.t, m, true, R, rs)) return;   // cannot happen: the fused kernel counted it
    const int rows = rs.y1 - rs.y0 + 1;
    // per-row lengths: thread i owns rows [16 i, 16 i + 16)
    uint32_t local[16], tsum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = threadIdx.x * 16 + k;
        uint32_t len = 0;
        if (r < rows) {
            int xa, xb;
            row_span(rs, rs.y0 + r, xa, xb);
            len = (uint32_t)max(xb - xa + 1, 0);
            rowxa[r] = (uint16_t)xa;
        }
        local[k] = len;
        tsum += len;
    }
    const uint32_t incl = wave_incl_scan(tsum, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t run = incl - tsum;
    for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = threadIdx.x * 16 + k;
        if (r < rows) rowoff[r] = run;
        run += local[k];
    }
    if (threadIdx.x == kBlock - 1) rowoff[rows] = run;   // == item.cnt
    __syncthreads();
    // the chunk's fragments: largest r with rowoff[r] <= f (rows of length 0 share an offset with their successor)
    for (uint32_t f = f0 + threadIdx.x; f < f1; f += kBlock) {
        int lo = 0, hi = rows;   // invariant: rowoff[lo] <= f < rowoff[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (rowoff[mid] <= f) lo = mid; else hi = mid;
        }
        const uint32_t x = (uint32_t)rowxa[lo] + (f - rowoff[lo]);
        entries[f - f0] = make_uint2(item.t, ((uint32_t)(rs.y0 + lo) << 16) | x);
    }
    __syncthreads();
    const uint32_t n_here = f1 - f0;
    const unsigned long long base = item.off + f0;
    // every fragment of this workgroup belongs to ONE triangle: its fragment constants are computed once, not per strip
    const ConstMeshPtr mp = kConstMesh(sc.meshes + m);
    TriShade ts;
    {
        float p[9];
        load_positions(sc.tri, item.t, p);
        const float bmin[3] = { mp->bmin[0], mp->bmin[1], mp->bmin[2] }, bmax[3] = { mp->bmax[0], mp->bmax[1], mp->bmax[2] };
        Geo g;
        geo_setup(p, bmin, bmax, g);
        tri_shade_setup(p, g, rs, mp, sc.tri.B0[item.t], sc.tri.B1[item.t], ts);
    }
    for (uint32_t e0 = 0; e0 < n_here; e0 += kBlock) {
        const uint32_t e = e0 + threadIdx.x;
        if (e < n_here) {
            const uint2 en = entries[e];
            float4 rec[6];
            shade_from_tri(sc.tri, item.t, (int)(en.y & 0xFFFFu), (int)(en.y >> 16), mp, ts, rec);
#pragma unroll
            for (int k = 0; k < 6; k++) stage[threadIdx.x * kStageStride + k] = rec[k];
        }
        __syncthreads();
        uint32_t nrec = min((uint32_t)kBlock, n_here - e0);
        const unsigned long long o0 = base + e0;
        if (o0 >= limit) nrec = 0;
        else if (limit - o0 < nrec) nrec = (uint32_t)(limit - o0);
        float4* __restrict__ dst = out + o0 * 6;
        for (uint32_t q = threadIdx.x; q < nrec * 6; q += kBlock) {
            const uint32_t r = q / 6, k = q - r * 6;
            nt_store(&dst[q], stage[r * kStageStride + k]);
        }
        __syncthreads();
    }
}

void launch_emit_big(const SceneDev& sc, uint32_t R, const BigItem* biglist, uint32_t n_big, uint32_t max_cnt, uint64_t limit,
                     float4* out, hipStream_t st) {
    if (!n_big || !max_cnt) return;
    const uint32_t chunks = (max_cnt + kEmitF - 1) / kEmitF;
    // grid.y is limited to 65535: loop over slabs of triangles
    for (uint32_t y0 = 0; y0 < n_big; y0 += 65535u) {
        const uint32_t ny = min(65535u, n_big - y0);
        hipLaunchKernelGGL(k_emit_big, dim3(chunks, ny), dim3(kBlock), 0, st, sc, R, biglist + y0, ny, (unsigned long long)limit, out);
    }
}

}  // namespace m2s
