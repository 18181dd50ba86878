// m2s_sort.hip — depth sort of the splat buffer (BASELINE config 5 "final radix sort of
// the merged splat buffer").  Mirrors the reference's RadixSortPass (src/renderer/renderPasses/
// RadixSortPass.cpp:8-90): key = floatBitsToUint(view-space z) (radixSortPrepass.glsl:23-33 on the depths
// written by gaussianSplattingPrepassCS.glsl:203), value = index, 32-bit LSD radix sort ascending on the raw
// bits (so negative view-space z sorts front to back), then a gather of the 6 x vec4 records
// (radixSortGather.glsl:30-49).  Key build and gather are hand-written; the sort itself is rocPRIM's
// device radix sort (a plain library primitive, like the reference's third-party glu::RadixSort).
#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

#include "m2s_device.h"
#include "m2s_viewmath.h"

#pragma clang fp contract(off)

namespace m2s {

// Key = raw bits of view-space z = row 2 of worldToView * (P, 1), pinned association.  The positions sit at a 96-byte stride
// inside the records: reading them touches EVERY 128-byte line of the record buffer (2.3 GB for BASELINE config 5's 24.3 M
// records, to use 16 bytes of each 96).  The reference sorts every frame (RadixSortPass runs per frame while the camera moves), so
// the first sort after the records have changed also leaves the positions behind as a compact plane (16 B per record) and every
// later sort of the same records builds its keys from that: 389 MB instead of 2.3 GB.
__device__ __forceinline__ uint32_t depth_key(float4 p, float v02, float v12, float v22, float v32) {
    const float z = ((v02 * p.x + v12 * p.y) + v22 * p.z) + v32;
    return __float_as_uint(z);
}
// The key the viewer's own sort uses (RadixSortPass.cpp:16-45): the raw bits of gaussianDepthPostFiltering = the prepass's view-space z — taken
// from m2s_viewmath.h's view_project, the very function k_prepass calls, so that sorting BEFORE the prepass (sort_prepass_permutation) orders by
// exactly the bits the prepass will store.  cull: records the prepass's frustum test rejects get the key kCulledKey and sort behind every
// survivor (they are counted; min / max are taken over the survivors only): the prepass then runs over the first `visible` positions alone.
struct DepthMVP { float M[16], V[16], P[16]; uint32_t cull; };
constexpr uint32_t kCulledKey = 0xFFFFFFFFu;
// Both key kernels also leave the smallest and the largest key of every WAVE behind (wave_mm[wave] = {min, max}: one 8-byte store per 64
// records) and k_reduce_minmax folds those into minmax[0] / [1]: view-space depths of a bounded scene share their sign and most of their
// exponent, so the keys differ only in their low 20-25 bits — sort_by_depth sorts `key - min` over exactly the bits that differ and saves a
// radix pass.  (Atomics from the key kernels themselves — 95 k waves on two addresses, even behind a plain read that skips most of them —
// or a grid-stride loop over few blocks each cost more than the pass they were to save: 0.095 -> 0.27-0.46 ms for the key stage.)
__device__ __forceinline__ void wave_minmax(uint32_t& kmin, uint32_t& kmax) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, d));
        kmax = max(kmax, (uint32_t)__shfl_xor((int)kmax, d));
    }
}
__device__ __forceinline__ void reduce_minmax(uint32_t kmin, uint32_t kmax, uint2* __restrict__ wave_mm) {
    wave_minmax(kmin, kmax);
    if ((threadIdx.x & 63) == 0) wave_mm[(blockIdx.x * kBlock + threadIdx.x) >> 6] = make_uint2(kmin, kmax);
}
__global__ void __launch_bounds__(kBlock) k_reduce_minmax(const uint2* __restrict__ wave_mm, uint32_t n_waves, uint32_t* __restrict__ minmax) {
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_waves; i += gridDim.x * kBlock) {
        const uint2 v = wave_mm[i];
        kmin = min(kmin, v.x); kmax = max(kmax, v.y);
    }
    wave_minmax(kmin, kmax);
    if ((threadIdx.x & 63) == 0) { atomicMin(&minmax[0], kmin); atomicMax(&minmax[1], kmax); }
}
__global__ void __launch_bounds__(kBlock) k_depth_keys(const float4* __restrict__ rec, uint32_t n, float v02, float v12,
                                                       float v22, float v32, uint32_t* __restrict__ key, float4* __restrict__ plane,
                                                       uint2* __restrict__ minmax) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    if (i < n) {
        const float4 p = rec[(size_t)i * 6];  // position (xyz, 1)
        const uint32_t k = depth_key(p, v02, v12, v22, v32);
        key[i] = k;
        if (plane) plane[i] = p;
        kmin = kmax = k;
    }
    reduce_minmax(kmin, kmax, minmax);
}
__global__ void __launch_bounds__(kBlock) k_depth_keys_from_plane(const float4* __restrict__ plane, uint32_t n, float v02, float v12,
                                                                  float v22, float v32, uint32_t* __restrict__ key, uint2* __restrict__ minmax) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    if (i < n) {
        const uint32_t k = depth_key(plane[i], v02, v12, v22, v32);
        key[i] = k;
        kmin = kmax = k;
    }
    reduce_minmax(kmin, kmax, minmax);
}

// wave_cnt (cull mode): survivors of every wave; bit 31 set if one of them has the key kCulledKey itself (a NaN depth with that payload:
// the caller then falls back to the prepass's own compaction)
__global__ void __launch_bounds__(kBlock) k_depth_keys_mv(const float4* __restrict__ rec, const float4* __restrict__ plane_in, uint32_t n, DepthMVP x,
                                                          uint32_t* __restrict__ key, float4* __restrict__ plane_out, uint2* __restrict__ minmax,
                                                          uint32_t* __restrict__ wave_cnt) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    uint32_t kmin = 0xFFFFFFFFu, kmax = 0u;
    bool vis = false, clash = false;
    if (i < n) {
        const float4 p = plane_in ? plane_in[i] : rec[(size_t)i * 6];
        float4 ws, vs, pos2d;
        vis = view_project(x.M, x.V, x.P, p.x, p.y, p.z, ws, vs, pos2d) || !x.cull;
        uint32_t k = __float_as_uint(vs.z);
        clash = vis && x.cull && k == kCulledKey;
        if (!vis) k = kCulledKey;
        key[i] = k;
        if (plane_out) plane_out[i] = p;
        if (vis) kmin = kmax = k;
    }
    if (x.cull) {
        const unsigned long long m = __ballot(vis);
        const unsigned long long c = __ballot(clash);
        if ((threadIdx.x & 63) == 0) wave_cnt[(blockIdx.x * kBlock + threadIdx.x) >> 6] = (uint32_t)__popcll(m) | (c ? 0x80000000u : 0u);
    }
    reduce_minmax(kmin, kmax, minmax);
}
// minmax[2] += survivors, minmax[3] |= clash flag
__global__ void __launch_bounds__(kBlock) k_reduce_counts(const uint32_t* __restrict__ wave_cnt, uint32_t n_waves, uint32_t* __restrict__ minmax) {
    uint32_t sum = 0, flag = 0;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n_waves; i += gridDim.x * kBlock) {
        const uint32_t v = wave_cnt[i];
        sum += v & 0x7FFFFFFFu; flag |= v >> 31;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, d); flag |= (uint32_t)__shfl_xor((int)flag, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&minmax[2], sum); if (flag) atomicOr(&minmax[3], 1u); }
}

__global__ void __launch_bounds__(kBlock) k_gather_records(const float4* __restrict__ src, const uint32_t* __restrict__ val,
                                                           uint32_t n, float4* __restrict__ dst) {
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= (size_t)n * 6) return;
    const uint32_t r = (uint32_t)(q / 6), k = (uint32_t)(q - (size_t)r * 6);
    // (non-temporal store: the sorted copy is read by another launch, not by this one — the one-shot copy of tools/copy_probe.hip
    // gains 6 % from it on the same boxes)
    const float4 v = src[(size_t)val[r] * 6 + k];
    float4* const d = &dst[q];
    __builtin_nontemporal_store(v.x, &d->x); __builtin_nontemporal_store(v.y, &d->y);
    __builtin_nontemporal_store(v.z, &d->z); __builtin_nontemporal_store(v.w, &d->w);
}
// the sorted keys of a reduced-range sort are `key - offset`: put the offset back (only when somebody asks for the keys)
__global__ void __launch_bounds__(kBlock) k_add_to_keys(uint32_t* __restrict__ key, uint32_t n, uint32_t offset) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) key[i] += offset;
}
void launch_add_to_keys(uint32_t* keys, uint32_t n, uint32_t offset, hipStream_t st) {
    if (n && offset) hipLaunchKernelGGL(k_add_to_keys, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, keys, n, offset);
}

// waves of the key kernels' grid = entries of the per-wave {min, max} table
static inline uint32_t sort_grid_waves(uint32_t n) { return (n + kBlock - 1) / kBlock * (kBlock / 64); }

struct SubtractKey {
    uint32_t m;
    __host__ __device__ uint32_t operator()(uint32_t k) const { return k - m; }
};
// cull mode: survivors' keys minus the smallest one; the culled ones (kCulledKey) all become `behind` = one more than the largest survivor
struct SubtractOrBehind {
    uint32_t m, behind;
    __host__ __device__ uint32_t operator()(uint32_t k) const { return k == kCulledKey ? behind : k - m; }
};

size_t sort_temp_bytes(uint32_t n) {
    size_t bytes = 0, bytes_t = 0, bytes_c = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, n, 0, 32,
                                    (hipStream_t)0);
    (void)rocprim::radix_sort_pairs(nullptr, bytes_t, rocprim::make_transform_iterator((uint32_t*)nullptr, SubtractKey{ 0u }), (uint32_t*)nullptr,
                                    rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, n, 0, 24, (hipStream_t)0);
    // + the two words of the key range and the per-wave {min, max} table of the key kernels (at the end of the buffer): one entry for
    // EVERY wave of the key kernels' grid, the idle waves of the last workgroup included ((n + 63) / 64 entries were up to
    // three short of what the grid writes)
    (void)rocprim::radix_sort_pairs(nullptr, bytes_c, rocprim::make_transform_iterator((uint32_t*)nullptr, SubtractOrBehind{ 0u, 0u }), (uint32_t*)nullptr,
                                    rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, n, 0, 24, (hipStream_t)0);
    const size_t lib = std::max(bytes, std::max(bytes_t, bytes_c));
    // (... and one survivor count per wave behind that table: m2s_prepass_sorted's cull mode)
    return ((lib + 15) & ~(size_t)15) + 16 + (size_t)sort_grid_waves(n) * 12;
}

// Keys of the n records + their stable radix sort: vals_out = the permutation (record index of every sorted position), keys_out = the
// sorted keys (minus *key_offset when the sort ran over the bits in which the keys differ).  mv == nullptr: key = view-space z by the
// row view[2], view[6], view[10], view[14] (m2s_sort_by_depth); else the prepass's own depth bits (depth_key_mv).
// plane: room for n float4 or nullptr; plane_valid: it already holds the positions of these records.  ev (or nullptr): events around the
// key stage and the radix sort (ev[0], ev[1], ev[2]).  The values are the record indices: they come from a counting iterator, not from memory.
static hipError_t keys_and_sort(const float4* rec, uint32_t n, const float view[16], const DepthMVP* mv, uint32_t* keys_in, uint32_t* keys_out,
                                uint32_t* vals_out, void* temp, size_t temp_bytes, float4* plane, bool plane_valid, hipEvent_t* ev, hipStream_t st,
                                uint32_t* key_offset, uint32_t* pinned_mm /* mv && mv->cull: FOUR pinned words */, uint32_t* n_visible, bool* clash) {
    *key_offset = 0;
    if (n_visible) *n_visible = n;
    if (clash) *clash = false;
    const bool cull = mv && mv->cull;
    const uint32_t n_waves = sort_grid_waves(n);       // (idle waves of the last workgroup store {0xFFFFFFFF, 0}: neutral for the fold)
    const size_t tail = 16 + (size_t)n_waves * 12;
    if (temp_bytes < tail + 16) return hipErrorInvalidValue;
    temp_bytes = (temp_bytes - tail) & ~(size_t)15;
    uint32_t* minmax = reinterpret_cast<uint32_t*>(static_cast<char*>(temp) + temp_bytes);     // {min, max, survivors, clash}
    uint2* wave_mm = reinterpret_cast<uint2*>(minmax + 4);
    uint32_t* wave_cnt = reinterpret_cast<uint32_t*>(wave_mm + n_waves);
    const uint32_t init[4] = { 0xFFFFFFFFu, 0u, 0u, 0u };
    hipError_t e = hipMemcpyAsync(minmax, init, sizeof init, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    const dim3 grid((n + kBlock - 1) / kBlock);
    if (ev) (void)hipEventRecord(ev[0], st);
    const bool from_plane = plane && plane_valid;
    if (mv) hipLaunchKernelGGL(k_depth_keys_mv, grid, dim3(kBlock), 0, st, rec, from_plane ? (const float4*)plane : (const float4*)nullptr, n, *mv, keys_in,
                               from_plane ? (float4*)nullptr : plane, wave_mm, wave_cnt);
    else if (from_plane) hipLaunchKernelGGL(k_depth_keys_from_plane, grid, dim3(kBlock), 0, st, plane, n, view[2], view[6], view[10], view[14], keys_in, wave_mm);
    else hipLaunchKernelGGL(k_depth_keys, grid, dim3(kBlock), 0, st, rec, n, view[2], view[6], view[10], view[14], keys_in, plane, wave_mm);
    const dim3 fold(std::min<uint32_t>((n_waves + kBlock - 1) / kBlock, 256u));
    hipLaunchKernelGGL(k_reduce_minmax, fold, dim3(kBlock), 0, st, wave_mm, n_waves, minmax);
    if (cull) hipLaunchKernelGGL(k_reduce_counts, fold, dim3(kBlock), 0, st, wave_cnt, n_waves, minmax);
    if (ev) (void)hipEventRecord(ev[1], st);
    // the keys' range decides how many radix passes the library makes: 8 (16) bytes come back to the host (one sync, ~15 us of a ~1.7 ms call)
    uint32_t mm_local[4] = { 0u, 0xFFFFFFFFu, n, 0u };
    uint32_t* mm = pinned_mm ? pinned_mm : mm_local;          // (pinned: the copy is a DMA the sync waits for, not a staged pageable copy)
    e = hipMemcpyAsync(mm, minmax, (cull ? 4 : 2) * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return e;
    if (cull) {
        if (n_visible) *n_visible = mm[2];
        if (clash) *clash = mm[3] != 0u;
        if (mm[2] == 0u) { if (ev) (void)hipEventRecord(ev[2], st); return hipSuccess; }      // nothing survives: nothing to sort
    }
    const uint32_t range = mm[1] >= mm[0] ? mm[1] - mm[0] : 0xFFFFFFFFu;
    if (cull && range != 0xFFFFFFFFu) {
        // survivors: key - min in [0, range]; the culled records: range + 1 — behind all of them, in input order among themselves
        int bits = 1;
        while (bits < 32 && ((range + 1u) >> bits) != 0u) ++bits;
        *key_offset = mm[0];
        e = rocprim::radix_sort_pairs(temp, temp_bytes, rocprim::make_transform_iterator(keys_in, SubtractOrBehind{ mm[0], range + 1u }), keys_out,
                                      rocprim::counting_iterator<uint32_t>(0), vals_out, n, 0, bits, st);
    } else {
        int bits = 1;
        while (bits < 32 && (range >> bits) != 0u) ++bits;
        if ((bits + 7) / 8 < 4) {   // fewer 8-bit passes than the full 32-bit sort: sort key - min over `bits` bits (same order, same stability)
            *key_offset = mm[0];
            e = rocprim::radix_sort_pairs(temp, temp_bytes, rocprim::make_transform_iterator(keys_in, SubtractKey{ *key_offset }), keys_out,
                                          rocprim::counting_iterator<uint32_t>(0), vals_out, n, 0, bits, st);
        } else {
            e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), vals_out, n, 0, 32, st);
        }
    }
    if (e != hipSuccess) return e;
    if (ev) (void)hipEventRecord(ev[2], st);
    return hipSuccess;
}

// stage_ev (or nullptr): four events recorded around the three stages (keys | radix sort | gather).
hipError_t sort_by_depth(const float4* rec, uint32_t n, const float view[16], uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out,
                         void* temp, size_t temp_bytes, float4* sorted, float4* plane, bool plane_valid, hipEvent_t* stage_ev, hipStream_t st,
                         uint32_t* key_offset_out, uint32_t* pinned_mm) {
    if (key_offset_out) *key_offset_out = 0;
    if (!n) return hipSuccess;
    uint32_t key_offset = 0;
    hipError_t e = keys_and_sort(rec, n, view, nullptr, keys_in, keys_out, vals_out, temp, temp_bytes, plane, plane_valid, stage_ev, st, &key_offset, pinned_mm,
                                 nullptr, nullptr);
    if (e != hipSuccess) return e;
    const size_t nq = (size_t)n * 6;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((nq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rec, vals_out, n, sorted);
    if (stage_ev) (void)hipEventRecord(stage_ev[3], st);
    if (key_offset_out) *key_offset_out = key_offset;            // keys_out holds key - offset (m2s_device_sorted_keys puts it back on demand)
    else launch_add_to_keys(keys_out, n, key_offset, st);
    return hipGetLastError();
}

// The viewer's depth sort taken BEFORE its prepass (m2s_prepass_sorted, m2s_viewer.cpp): the permutation that orders the records by the
// depth the prepass is going to store (model = u_modelToWorld, view = u_worldToView, both column-major), no gather — the prepass reads the
// records through it and appends its survivors in that order, so the 96-byte gather of RadixSortPass::gatherPost (radixSortGather.glsl)
// and the prepass's own read of the records become ONE pass over them.
// cull: the prepass's frustum test is applied here (it depends on the position alone; NOT to be used when the prepass also tests against a
// depth image, which needs the record's alpha): *n_visible = survivors, and they occupy the first *n_visible positions of the permutation.
// *clash: a survivor's own key equals the marker of the culled ones (a NaN depth with an all-ones payload): call again without cull.
hipError_t sort_prepass_permutation(const float4* rec, uint32_t n, const float model[16], const float view[16], const float proj[16], bool cull,
                                    uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out, void* temp, size_t temp_bytes, float4* plane, bool plane_valid,
                                    hipEvent_t* ev, hipStream_t st, uint32_t* pinned_mm4, uint32_t* n_visible, bool* clash) {
    if (n_visible) *n_visible = n;
    if (clash) *clash = false;
    if (!n) return hipSuccess;
    DepthMVP mv;
    memcpy(mv.M, model, sizeof mv.M);
    memcpy(mv.V, view, sizeof mv.V);
    memcpy(mv.P, proj, sizeof mv.P);
    mv.cull = cull ? 1u : 0u;
    uint32_t key_offset = 0;
    const hipError_t e = keys_and_sort(rec, n, view, &mv, keys_in, keys_out, vals_out, temp, temp_bytes, plane, plane_valid, ev, st, &key_offset, pinned_mm4,
                                       n_visible, clash);
    return e != hipSuccess ? e : hipGetLastError();
}

// RadixSortPass::execute proper (RadixSortPass.cpp:8-90), on the prepass output: keys are the raw bits of
// gaussianDepthPostFiltering (radixSortPrepass.glsl:23-33 copies them and writes val = gid: here the key buffer IS the
// depth buffer and the values come from a counting iterator, so that kernel disappears), then the gather of the
// six-vec4 QuadNdcTransformations (radixSortGather.glsl:30-49).
size_t sort_prepass_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, rocprim::counting_iterator<uint32_t>(0),
                                    (uint32_t*)nullptr, n, 0, 32, (hipStream_t)0);
    return bytes;
}

hipError_t sort_prepass(const float* depths, const float4* quads, uint32_t n, uint32_t* keys_out, uint32_t* vals_out, void* temp,
                        size_t temp_bytes, float4* sorted, hipStream_t st) {
    if (!n) return hipSuccess;
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, reinterpret_cast<const uint32_t*>(depths), keys_out,
                                             rocprim::counting_iterator<uint32_t>(0), vals_out, n, 0, 32, st);
    if (e != hipSuccess) return e;
    const size_t nq = (size_t)n * 6;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((nq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, quads, vals_out, n, sorted);
    return hipGetLastError();
}

// ---- pieces of the sample sort across ranks (m2s_dist_sort_by_depth, m2s_dist.cpp) ----
// out[i] = the key at position ((i + 1) * n) / (take + 1) of the rank's SORTED keys, i < take = min(s, n); the remaining
// slots hold the "no sample" value 2^64 - 1; out[s] = take.
__global__ void k_pick_samples(const uint32_t* __restrict__ keys, unsigned long long n, uint32_t s, unsigned long long* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long take = n < s ? n : s;
    if (i < s) {
        unsigned long long v = ~0ull;
        if (i < take) {
            unsigned long long pos = ((unsigned long long)(i + 1) * n) / (take + 1);
            if (pos > n - 1) pos = n - 1;
            v = keys[pos];
        }
        out[i] = v;
    } else if (i == s) {
        out[s] = take;
    }
}

// out[j] = number of keys < splitter[j] (keys ascending): where the block that goes to rank j + 1 starts
__global__ void k_lower_bounds(const uint32_t* __restrict__ keys, unsigned long long n, const unsigned long long* __restrict__ splitters, uint32_t m,
                               unsigned long long* __restrict__ out) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const unsigned long long sp = splitters[j];
    unsigned long long lo = 0, hi = n;
    while (lo < hi) {
        const unsigned long long mid = lo + (hi - lo) / 2;
        if ((unsigned long long)keys[mid] < sp) lo = mid + 1; else hi = mid;
    }
    out[j] = lo;
}

void launch_pick_samples(const uint32_t* keys, uint64_t n, uint32_t s, unsigned long long* out, hipStream_t st) {
    hipLaunchKernelGGL(k_pick_samples, dim3((s + 1 + 255) / 256), dim3(256), 0, st, keys, (unsigned long long)n, s, out);
}

void launch_lower_bounds(const uint32_t* keys, uint64_t n, const unsigned long long* splitters, uint32_t m, unsigned long long* out, hipStream_t st) {
    if (!m) return;
    hipLaunchKernelGGL(k_lower_bounds, dim3((m + 63) / 64), dim3(64), 0, st, keys, (unsigned long long)n, splitters, m, out);
}

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_sort() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_gather_records)); }

}  // namespace m2s
