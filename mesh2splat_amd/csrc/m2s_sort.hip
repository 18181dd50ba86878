// m2s_sort.hip — depth sort of the splat buffer (BASELINE config 5 "final radix sort of
// the merged splat buffer").  Mirrors the reference's RadixSortPass (src/renderer/renderPasses/
// RadixSortPass.cpp:8-90): key = floatBitsToUint(view-space z) (radixSortPrepass.glsl:23-33 on the depths
// written by gaussianSplattingPrepassCS.glsl:203), value = index, 32-bit LSD radix sort ascending on the raw
// bits (so negative view-space z sorts front to back), then a gather of the 6 x vec4 records
// (radixSortGather.glsl:30-49).  Key build and gather are hand-written; the sort itself is rocPRIM's
// device radix sort (a plain library primitive, like the reference's third-party glu::RadixSort).
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "m2s_device.h"

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
__global__ void __launch_bounds__(kBlock) k_depth_keys(const float4* __restrict__ rec, uint32_t n, float v02, float v12,
                                                       float v22, float v32, uint32_t* __restrict__ key, float4* __restrict__ plane) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4 p = rec[(size_t)i * 6];  // position (xyz, 1)
    key[i] = depth_key(p, v02, v12, v22, v32);
    if (plane) plane[i] = p;
}
__global__ void __launch_bounds__(kBlock) k_depth_keys_from_plane(const float4* __restrict__ plane, uint32_t n, float v02, float v12,
                                                                  float v22, float v32, uint32_t* __restrict__ key) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    key[i] = depth_key(plane[i], v02, v12, v22, v32);
}

// one thread per float4: consecutive lanes write consecutive 16 B of the sorted buffer (fully coalesced
// 1 KiB stores); the 96 B source records are gathered (6 lanes share one record)
__global__ void __launch_bounds__(kBlock) k_gather_records(const float4* __restrict__ src, const uint32_t* __restrict__ val,
                                                           uint32_t n, float4* __restrict__ dst) {
    const size_t q = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (q >= (size_t)n * 6) return;
    const uint32_t r = (uint32_t)(q / 6), k = (uint32_t)(q - (size_t)r * 6);
    dst[q] = src[(size_t)val[r] * 6 + k];
}

size_t sort_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint32_t*)nullptr, (uint32_t*)nullptr, rocprim::counting_iterator<uint32_t>(0), (uint32_t*)nullptr, n, 0, 32,
                                    (hipStream_t)0);
    return bytes;
}

// plane: room for n float4 or nullptr; plane_valid: it already holds the positions of these records.  stage_ev (or nullptr): four
// events recorded around the three stages (keys | radix sort | gather).  The values are the record indices: they come from a
// counting iterator, not from memory.
hipError_t sort_by_depth(const float4* rec, uint32_t n, const float view[16], uint32_t* keys_in, uint32_t* keys_out, uint32_t* vals_out,
                         void* temp, size_t temp_bytes, float4* sorted, float4* plane, bool plane_valid, hipEvent_t* stage_ev, hipStream_t st) {
    if (!n) return hipSuccess;
    const dim3 grid((n + kBlock - 1) / kBlock);
    if (stage_ev) (void)hipEventRecord(stage_ev[0], st);
    if (plane && plane_valid) hipLaunchKernelGGL(k_depth_keys_from_plane, grid, dim3(kBlock), 0, st, plane, n, view[2], view[6], view[10], view[14], keys_in);
    else hipLaunchKernelGGL(k_depth_keys, grid, dim3(kBlock), 0, st, rec, n, view[2], view[6], view[10], view[14], keys_in, plane);
    if (stage_ev) (void)hipEventRecord(stage_ev[1], st);
    hipError_t e = rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), vals_out, n, 0, 32, st);
    if (e != hipSuccess) return e;
    if (stage_ev) (void)hipEventRecord(stage_ev[2], st);
    const size_t nq = (size_t)n * 6;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((nq + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, rec, vals_out, n, sorted);
    if (stage_ev) (void)hipEventRecord(stage_ev[3], st);
    return hipGetLastError();
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
