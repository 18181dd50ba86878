/* tests/exact_math/div_identity.c — TEST INFRASTRUCTURE (compiled and run by tests/test_round6_math.py; CPU only).
 * The arithmetic identities behind mesh2splat_amd/csrc/m2s_exact.h, in IEEE fp32 with a fused multiply-add — the same operations
 * the GPU executes (v_mul_f32, v_fma_f32), so what holds here holds there:
 *   div:   y = RN(1/b);  q = RN(a y);  r = a - q b (one FMA: exact);  RN(q + r y) == RN(a / b)
 *          for every dividend significand (a in [1, 2)) against a list of divisor significands (argv: count, seed).
 *   seeds: the reciprocal / square-root sequences started from the CORRECTLY ROUNDED seed are exact for every significand; started
 *          one ulp off they are not (a few significands): they rest on the hardware's actual seeds, which is why their proof is the
 *          exhaustive run ON the GPU (tests/test_gpu_exact_math.py), not this file.
 * Output: one line of JSON. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(int argc, char** argv) {
    const int nd = argc > 1 ? atoi(argv[1]) : 64;
    uint32_t s = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 0) : 12345u;
    unsigned long long bad_div = 0, pairs = 0;
    uint32_t first_a = 0, first_b = 0;
    for (int d = 0; d < nd; d++) {
        uint32_t mb;
        if (d == 0) mb = 0; else if (d == 1) mb = 0x7FFFFFu; else if (d == 2) mb = 0x7FFFFEu; else if (d == 3) mb = 1; else if (d == 4) mb = 0x400000u;
        else { s = s * 1664525u + 1013904223u; mb = s >> 9; }
        const float b = u2f((127u << 23) | mb), y = 1.0f / b;
        unsigned long long l = 0;
        uint32_t fa = 0;
#pragma omp parallel for reduction(+ : l) reduction(max : fa)
        for (uint32_t i = 0; i < (1u << 23); i++) {
            const float a = u2f((127u << 23) + i), ref = a / b;
            const float q = a * y, r = __builtin_fmaf(-q, b, a), q1 = __builtin_fmaf(r, y, q);
            if (f2u(q1) != f2u(ref)) { l++; if (f2u(a) > fa) fa = f2u(a); }
        }
        if (l && !bad_div) { first_a = fa; first_b = f2u(b); }
        bad_div += l; pairs += 1u << 23;
    }
    unsigned long long bad_rcp[3] = { 0, 0, 0 }, bad_sqrt[3] = { 0, 0, 0 };
#pragma omp parallel for reduction(+ : bad_rcp[:3], bad_sqrt[:3])
    for (uint32_t m = 0; m < (1u << 24); m++) {
        if (m < (1u << 23)) {                      /* 1/x, x in [1, 2) */
            const float x = u2f((127u << 23) | m), ref = 1.0f / x;
            for (int k = -1; k <= 1; k++) {
                const float y = u2f(f2u(ref) + k), e = __builtin_fmaf(-x, y, 1.0f), r = __builtin_fmaf(e, y, y);
                if (f2u(r) != f2u(ref)) bad_rcp[k + 1]++;
            }
        }
        const float x = u2f((127u << 23) + m), ref = sqrtf(x), yref = (float)(1.0 / sqrt((double)x));   /* sqrt, x in [1, 4) */
        for (int k = -1; k <= 1; k++) {
            const float y = u2f(f2u(yref) + k), sx = x * y, h = 0.5f * y, r = __builtin_fmaf(-sx, sx, x), s1 = __builtin_fmaf(r, h, sx);
            if (f2u(s1) != f2u(ref)) bad_sqrt[k + 1]++;
        }
    }
    printf("{\"divisors\": %d, \"pairs\": %llu, \"div_mismatches\": %llu, \"first\": [\"0x%08x\", \"0x%08x\"], "
           "\"rcp_mismatches_by_seed_offset\": [%llu, %llu, %llu], \"sqrt_mismatches_by_seed_offset\": [%llu, %llu, %llu]}\n",
           nd, pairs, bad_div, first_a, first_b, bad_rcp[0], bad_rcp[1], bad_rcp[2], bad_sqrt[0], bad_sqrt[1], bad_sqrt[2]);
    return 0;
}
