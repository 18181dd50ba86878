// tests/exact_math/exact_math_check.hip — TEST INFRASTRUCTURE (built by tests/exact_math/Makefile, run by tests/test_gpu_exact_math.py).
// Exhaustive comparison, on the GPU itself, of the short correctly rounded sequences the conversion kernels ship
// (mesh2splat_amd/csrc/m2s_exact.h: rcp_rn, sqrt_rn, div_rn — included here, not restated) with the compiler's IEEE expansions of
// `1.0f / x`, `sqrtf(x)`, `a / b` (v_div_scale / v_div_fmas / v_div_fixup; correctly rounded by specification).
//
//   exact_math_check rcp                      every |x| in [2^-64, 2^64], both signs                (2 x 129 x 2^23 operands)
//   exact_math_check sqrt                     every x in [2^-96, 2^100]                             (197 x 2^23 operands)
//   exact_math_check div <first> <count>      divisor significands [first, first + count) x all 2^23 dividend significands
//   exact_math_check divall [seconds]         all 2^23 divisor significands: 7.0e13 pairs, 44 s on one MI355X
// (div / divall also draw 2^24 random pairs with exponents over the whole guarded range [2^-60, 2^60] and both signs.)
// One JSON line per sequence: {"probe", "candidate", "mismatches", "first"}.
//
// Scale invariance: none of the sequences reads the exponent, and inside the guarded ranges no intermediate overflows, underflows
// or leaves the normal range (m2s_exact.h), so every intermediate scales exactly with a power of two and the operand SIGNIFICANDS are
// the whole domain of a / b: a, b in [1, 2) gives the quotient significand ma / mb or 2 ma / mb — both alignments.
// Also reported, for the record: what the hardware seeds alone (v_rcp_f32, v_sqrt_f32) and a second correction step would give.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

#include "../../mesh2splat_amd/csrc/m2s_exact.h"
using namespace m2s;

#pragma clang fp contract(off)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

// ---- for the record only: the seeds alone, and one more correction step than the shipped sequences take ---------------------
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float rcp_nr2(float x) { float y = rcp_rn(x); float e = fma_(-x, y, 1.0f); return fma_(e, y, y); }
__device__ __forceinline__ float sqrt_nr2(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y, h = 0.5f * y;
    const float e = fma_(-h, s, 0.5f);
    s = fma_(s, e, s); h = fma_(h, e, h);
    const float r = fma_(-s, s, x);
    return fma_(r, h, s);
}
__device__ __forceinline__ float div_c2(float a, float b, float y) { float q = div_rn(a, b, y); float r = fma_(-q, b, a); return fma_(r, y, q); }

struct Bad { unsigned long long n[4]; uint32_t first[4][2]; };

__device__ void note(Bad* bad, int k, uint32_t a, uint32_t b) {
    if (atomicAdd(&bad->n[k], 1ull) == 0ull) { bad->first[k][0] = a; bad->first[k][1] = b; }
}

__global__ void k_rcp(Bad* bad, uint32_t lo) {
    const uint32_t bits = lo + blockIdx.x * blockDim.x + threadIdx.x;   // every float of [2^-64, 2^64] ...
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {                                    // ... and its negative
        const float x = __uint_as_float(bits | ((uint32_t)sg << 31));
        const float ref = 1.0f / x;
        if (__float_as_uint(rcp_rn(x)) != __float_as_uint(ref)) note(bad, 0, __float_as_uint(x), 0);
        if (__float_as_uint(rcp_nr2(x)) != __float_as_uint(ref)) note(bad, 1, __float_as_uint(x), 0);
        if (__float_as_uint(__builtin_amdgcn_rcpf(x)) != __float_as_uint(ref)) note(bad, 2, __float_as_uint(x), 0);
    }
}

__global__ void k_sqrt(Bad* bad, uint32_t lo) {
    const uint32_t bits = lo + blockIdx.x * blockDim.x + threadIdx.x;   // every float of [2^-96, 2^100]
    const float x = __uint_as_float(bits);
    const float ref = sqrtf(x);
    if (__float_as_uint(sqrt_rn(x)) != __float_as_uint(ref)) note(bad, 0, bits, 0);
    if (__float_as_uint(sqrt_nr2(x)) != __float_as_uint(ref)) note(bad, 1, bits, 0);
    if (__float_as_uint(__builtin_amdgcn_sqrtf(x)) != __float_as_uint(ref)) note(bad, 2, bits, 0);
    // the composition the kernels use for 1.0f / length: root in [2^-48, 2^50], inside rcp_rn's range
    if (__float_as_uint(rcp_rn(sqrt_rn(x))) != __float_as_uint(1.0f / ref)) note(bad, 3, bits, 0);
}

// one workgroup per divisor mantissa; its threads walk all 2^23 dividend mantissas (a in [1, 2), b in [1, 2): the quotient's
// significand is ma / mb or 2 ma / mb — both alignments)
__global__ void __launch_bounds__(256) k_div(Bad* bad, uint32_t mb0) {
    const uint32_t mb = mb0 + blockIdx.x;
    const float b = __uint_as_float((127u << 23) | (mb & 0x7FFFFFu));
    const float y = rcp_rn(b);                                   // (= 1.0f / b: k_rcp)
    unsigned long long bad1 = 0, bad2 = 0;
    uint32_t f1 = 0, f2 = 0;
    for (uint32_t i = threadIdx.x; i < (1u << 23); i += 256u) {
        const uint32_t abits = (127u << 23) + i;
        const float a = __uint_as_float(abits);
        const float ref = a / b;
        const float q1 = div_rn(a, b, y), q2 = div_c2(a, b, y);
        if (__float_as_uint(q1) != __float_as_uint(ref)) { if (!bad1) f1 = abits; ++bad1; }
        if (__float_as_uint(q2) != __float_as_uint(ref)) { if (!bad2) f2 = abits; ++bad2; }
    }
    if (bad1) { if (atomicAdd(&bad->n[0], bad1) == 0ull) { bad->first[0][0] = f1; bad->first[0][1] = __float_as_uint(b); } }
    if (bad2) { if (atomicAdd(&bad->n[1], bad2) == 0ull) { bad->first[1][0] = f2; bad->first[1][1] = __float_as_uint(b); } }
}

// random pairs over the whole guarded range of exponents: |a|, b in [2^-60, 2^60], both signs of a
__global__ void __launch_bounds__(256) k_div_edges(Bad* bad, uint32_t seed) {
    uint32_t s = seed * 2654435761u + blockIdx.x * 0x9E3779B9u + threadIdx.x * 0x85EBCA6Bu;
    unsigned long long bad1 = 0, bad2 = 0;
    uint32_t fa = 0, fb = 0;
    for (int it = 0; it < 4096; ++it) {
        s = s * 1664525u + 1013904223u; const uint32_t ma = s >> 9;
        s = s * 1664525u + 1013904223u; const uint32_t mbb = s >> 9;
        s = s * 1664525u + 1013904223u;
        const uint32_t ea = 67u + (s >> 8) % 120u, eb = 67u + (s >> 20) % 120u;     // biased exponents 67 .. 186 = 2^-60 .. 2^59
        const uint32_t sign = (s & 1u) << 31;
        const float a = __uint_as_float(sign | (ea << 23) | ma), b = __uint_as_float((eb << 23) | mbb);
        const float y = rcp_rn(b);
        const float ref = a / b;
        if (__float_as_uint(div_rn(a, b, y)) != __float_as_uint(ref)) { if (!bad1) { fa = __float_as_uint(a); fb = __float_as_uint(b); } ++bad1; }
        if (__float_as_uint(div_c2(a, b, y)) != __float_as_uint(ref)) ++bad2;
    }
    if (bad1) { if (atomicAdd(&bad->n[2], bad1) == 0ull) { bad->first[2][0] = fa; bad->first[2][1] = fb; } }
    if (bad2) atomicAdd(&bad->n[3], bad2);
}

static void report(const char* what, const Bad& b, const char* const names[4], int n) {
    for (int k = 0; k < n; ++k)
        printf("{\"probe\": \"%s\", \"candidate\": \"%s\", \"mismatches\": %llu, \"first\": [\"0x%08x\", \"0x%08x\"]}\n", what, names[k], b.n[k],
               b.first[k][0], b.first[k][1]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "rcp";
    Bad* d; Bad h;
    CK(hipMalloc(&d, sizeof(Bad)));
    CK(hipMemset(d, 0, sizeof(Bad)));
    if (!strcmp(mode, "rcp")) {
        const uint32_t lo = (127u - 64u) << 23, hi = ((127u + 64u) << 23) + 1u;        // 2^-64 ... 2^64 inclusive
        for (uint32_t at = lo; at < hi; at += 1u << 26) {
            const uint32_t n = hi - at < (1u << 26) ? hi - at : (1u << 26);
            hipLaunchKernelGGL(k_rcp, dim3((n + 255) / 256), dim3(256), 0, 0, d, at);   // (the last launch runs up to 255 floats past 2^64: inside the proven range all the same)
        }
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
        const char* nm[4] = { "rcp_rn (shipped): v_rcp_f32 + 1 Newton step", "v_rcp_f32 + 2 Newton steps", "v_rcp_f32 alone", "" };
        report("rcp: every |x| in [2^-64, 2^64], both signs, vs 1.0f / x", h, nm, 3);
    } else if (!strcmp(mode, "sqrt")) {
        const uint32_t lo = (127u - 96u) << 23, hi = ((127u + 100u) << 23) + 1u;       // 2^-96 ... 2^100 inclusive
        for (uint32_t at = lo; at < hi; at += 1u << 26) {
            const uint32_t n = hi - at < (1u << 26) ? hi - at : (1u << 26);
            hipLaunchKernelGGL(k_sqrt, dim3((n + 255) / 256), dim3(256), 0, 0, d, at);
        }
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
        const char* nm[4] = { "sqrt_rn (shipped): v_rsq_f32, s = x y, one residual step", "v_rsq_f32, coupled step + residual step", "v_sqrt_f32 alone",
                              "rcp_rn(sqrt_rn(x)) vs 1.0f / sqrtf(x)" };
        report("sqrt: every x in [2^-96, 2^100] vs sqrtf(x)", h, nm, 4);
    } else if (!strcmp(mode, "div") || !strcmp(mode, "divall")) {
        uint32_t m0 = 0, cnt = 1u << 23;
        double budget = 1e30;
        if (!strcmp(mode, "div")) { m0 = argc > 2 ? strtoul(argv[2], 0, 0) : 0; cnt = argc > 3 ? strtoul(argv[3], 0, 0) : 4096; }
        else budget = argc > 2 ? atof(argv[2]) : 120.0;
        hipLaunchKernelGGL(k_div_edges, dim3(4096), dim3(256), 0, 0, d, 12345u);
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t done = 0;
        const uint32_t chunk = 1u << 14;
        while (done < cnt) {
            const uint32_t n = cnt - done < chunk ? cnt - done : chunk;
            hipLaunchKernelGGL(k_div, dim3(n), dim3(256), 0, 0, d, m0 + done);
            CK(hipDeviceSynchronize());
            done += n;
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (el > budget) break;
        }
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CK(hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost));
        const char* nm[4] = { "div_rn (shipped): q = a y, one residual correction", "two residual corrections",
                              "div_rn (shipped), 2^24 random pairs with exponents over [2^-60, 2^60], both signs", "two corrections, the same random pairs" };
        char what[256];
        snprintf(what, sizeof what, "div: divisor mantissas [%u, %u) x 2^23 dividends = %.4g pairs in %.1f s, y = rcp_rn(b), vs a / b", m0, m0 + done,
                 (double)done * 8388608.0, el);
        report(what, h, nm, 4);
        printf("{\"divisors_done\": %u, \"of\": %u, \"complete\": %s}\n", done, cnt, done == cnt ? "true" : "false");
    } else { fprintf(stderr, "mode?\n"); return 1; }
    return 0;
}
