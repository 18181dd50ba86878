// m2s_exact.h — the correctly rounded fp32 reciprocal, square root and division of the DECISION arithmetic (m2s_devfn.h: geo_setup,
// geo_flat, tri_shade_*), in 3 / 5 / 3 instructions instead of the compiler's 11 / 16 / 11 (v_div_scale x 2 + v_rcp + 6 FMA +
// v_div_fmas + v_div_fixup; range scaling + v_sqrt + two neighbour tests + class fix-up).  Round 6, VERDICT r5 item 6.
//
// The pinned semantics are IEEE: `u = rel / range`, `1.0f / length(e)`, `sqrtf(dot(e, e))` as the reference's shader writes them
// (converterGS.glsl:326-399) and as the oracle evaluates them.  These sequences return THE SAME BITS — not an approximation:
//
//   rcp_rn(x)       = RN(1 / x)      v_rcp_f32 (<= 1 ulp) + one Newton step in two FMAs
//   sqrt_rn(x)      = RN(sqrt(x))    v_rsq_f32 (<= 1 ulp), s = x y, h = y / 2, one residual step  s + (x - s s) h  in two FMAs
//   div_rn(a, b, y) = RN(a / b)      given y = RN(1 / b):  q = a y,  r = a - q b (exact in one FMA),  q + r y
//
// Proof = exhaustion on the hardware itself (tests/exact_math/exact_math_check.hip, run by tests/test_gpu_exact_math.py; the
// complete run is profiles/r06/exact_math_exhaustive.jsonl): none of the sequences reads the exponent, and inside the guarded
// ranges below no intermediate overflows, underflows or leaves the normal range, so every intermediate scales exactly with powers
// of two and the operand SIGNIFICANDS are the whole domain — 2^23 for 1/x (checked at every exponent of the guarded range all the
// same: 129 x 2^23 operands), every float of [2^-96, 2^100] for sqrt, and all 2^23 x 2^23 = 7.0e13 (dividend, divisor) pairs for
// a / b (44 s on one MI355X), each compared with the compiler's IEEE expansion: zero mismatches.  The division identity involves no
// hardware approximation (y is the correctly rounded reciprocal, the rest is FMA arithmetic) and is re-checked on the CPU over a
// sample of divisors x all 2^23 dividends by tests/test_round6_math.py.
//
// Outside the guarded ranges (operands near the denormals or the overflow threshold, zero, infinities) the callers fall back to
// the compiler's IEEE expansion, wave-uniformly (`wave_all`): a scalar branch that real meshes take for the few triangles with
// a zero-length edge or a vertex exactly on its mesh's bounding-box plane.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace m2s {

constexpr float kSqrtLo = 0x1p-96f, kSqrtHi = 0x1p100f;    // sqrt_rn: x in [kSqrtLo, kSqrtHi] (its root: [2^-48, 2^50], inside rcp_rn's range)
constexpr float kRcpLo = 0x1p-64f, kRcpHi = 0x1p64f;       // rcp_rn: |x| in [kRcpLo, kRcpHi]
constexpr float kDivLo = 0x1p-60f, kDivHi = 0x1p60f;       // div_rn: |a|, b in [kDivLo, kDivHi]  (|a / b| in [2^-120, 2^120]: normal; a - q b exact)

__device__ __forceinline__ float rcp_rn(float x) {
    const float y = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ float sqrt_rn(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
__device__ __forceinline__ float div_rn(float a, float b, float y) {
    const float q = a * y;
    const float r = __builtin_fmaf(-q, b, a);
    return __builtin_fmaf(r, y, q);
}

// true if `c` holds on every active lane of the wave (wave-uniform: the callers branch on it with a scalar branch)
__device__ __forceinline__ bool wave_all(bool c) { return __ballot(!c) == 0ull; }

__device__ __forceinline__ bool in_sqrt_range(float x) { return x >= kSqrtLo && x <= kSqrtHi; }                       // false for NaN
__device__ __forceinline__ bool in_sqrt_range3(float a, float b, float c) {
    return fminf(fminf(a, b), c) >= kSqrtLo && fmaxf(fmaxf(a, b), c) <= kSqrtHi;     // (a NaN hidden by min / max takes the fast path: NaN either way)
}

}  // namespace m2s
