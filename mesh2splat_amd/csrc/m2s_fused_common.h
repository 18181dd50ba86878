// m2s_fused_common.h — pieces shared by the single-pass kernels (m2s_fused2.hip, m2s_fused3.hip, m2s_sparse.hip): triangle classes,
// the look-back chain (word format, loads/stores, the look-back itself) and the rare medium-triangle expansion.
#pragma once
#include "m2s_devfn.h"

#pragma clang fp contract(off)

namespace m2s {

constexpr uint32_t kBigCount = 96;         // triangles with more fragments, or more than kFusedRows pixel rows, are
constexpr int kFusedRows = 16;             // only counted here and emitted by the multi-pass pipeline
constexpr uint32_t kSpinLimit = 1u << 22;  // look-back polls before giving up (~ seconds)

// chain word = flag(2) | epoch(16) | value(46).  The epoch changes with every launch, so words left over
// from the previous launch read as "not published" and the chain needs no per-launch memset.
constexpr unsigned long long kFlagAgg = 1ull << 62, kFlagPrefix = 2ull << 62, kValMask = (1ull << 46) - 1;
constexpr int kEpochShift = 46;
__device__ __forceinline__ unsigned chain_flag(unsigned long long v, uint32_t epoch) {
    return (((v >> kEpochShift) & 0xFFFFu) == epoch) ? (unsigned)(v >> 62) : 0u;
}

__device__ __forceinline__ unsigned long long chain_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ unsigned long long wave_incl_scan64(unsigned long long v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned long long n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}

enum : int { kNone = 0, kSmall = 1, kMedium = 2, kBig = 3 };


// Rare path, deliberately NOT inlined: its closed-form span code (fp64 divisions) would otherwise add ~30
// VGPRs of pressure to the fragment loop of every wave.  Re-derives the raster setup from global memory.
[[maybe_unused]] static __device__ __noinline__ void expand_medium(const float4* A0, const float4* A1, const float* A2, const MeshParams* mp,
                                           uint32_t t, uint32_t R, uint32_t lane, uint32_t cto, uint32_t win, uint32_t wend,
                                           uint32_t* entries) {
    const float4 a0 = A0[t], a1 = A1[t];
    const float p[9] = { a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, A2[t] };
    Geo g;
    geo_setup(p, mp->bmin, mp->bmax, g);
    Raster r2;
    if (!raster_setup(g, R, r2)) return;
    uint32_t ci = cto;
    RowWalker rw;
    row_walker_init(r2, r2.y0, rw);
    for (int y = r2.y0; y <= r2.y1 && ci < wend; ++y) {
        int xa, xb;
        row_walker_next(rw, xa, xb);
        for (int x = xa; x <= xb; ++x, ++ci)
            if (ci >= win && ci < wend) entries[ci - win] = (lane << 24) | ((uint32_t)y << 12) | (uint32_t)x;
    }
}

// Decoupled look-back: sum of the totals of all waves before `wid`.  Each poll inspects kLbWindows
// windows of 64 consecutive chain words (lane l of window j reads word first-64j-l: coalesced 512-byte
// reads, all issued before the first is consumed = one memory round trip for 512 predecessors).  All
// resident waves reach this point at a similar age, so the nearest published PREFIX is typically several
// hundred entries back; a 64- or 256-entry reach cost 3-4 sequential round trips per wave.
// Not inlined: runs once per wave.
constexpr int kLbWindows = 8;
[[maybe_unused]] static __device__ __noinline__ unsigned long long lookback(const unsigned long long* chain, uint32_t wid, int lane, uint32_t epoch,
                                                    uint32_t* status) {
    const unsigned long long virt_prefix = kFlagPrefix | ((unsigned long long)epoch << kEpochShift);
    unsigned long long acc = 0;             // per-lane partial sum; reduced across the wave once, at the end
    long long first = (long long)wid - 1;   // nearest predecessor not yet accounted for
    uint32_t spins = 0;
    for (;;) {
        unsigned long long v[kLbWindows];
#pragma unroll
        for (int j = 0; j < kLbWindows; ++j) {
            const long long ij = first - 64 * j - lane;
            v[j] = ij >= 0 ? chain_load(&chain[ij]) : virt_prefix;
        }
        bool done = false, stalled = false;
#pragma unroll
        for (int j = 0; j < kLbWindows; ++j) {
            if (done || stalled) continue;
            const unsigned flag = chain_flag(v[j], epoch);
            const unsigned long long pm = __ballot(flag == 2), im = __ballot(flag == 0);
            if (pm) {
                const int pl = __ffsll((long long)pm) - 1;        // nearest inclusive prefix in this window
                if ((im & ((1ull << pl) - 1ull)) == 0) {           // every nearer entry is published
                    if (lane <= pl) acc += v[j] & kValMask;
                    done = true;
                } else stalled = true;
            } else if (im == 0) {                                  // 64 aggregates: take them, go further back
                acc += v[j] & kValMask;
                first -= 64;
            } else stalled = true;
        }
        if (done) break;
        if (stalled) {
            if (++spins > kSpinLimit) {
                if (lane == 0) __hip_atomic_store(&status[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const unsigned long long base = wave_sum64(acc);
    return ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
           __builtin_amdgcn_readfirstlane((uint32_t)base);
}


}  // namespace m2s
