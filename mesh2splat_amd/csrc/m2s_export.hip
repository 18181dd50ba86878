// m2s_export.hip — device-side .ply row encoder for formats 1 (PBR, 76 B / row) and 2 (compressed PBR, 48 B / row):
// SURVEY.md §8 f-1.  Restates, per Gaussian, what the reference's writers compute on one CPU thread with one
// ofstream::write per field (src/parsers/parsers.cpp:232-316 PBR, :339-428 compressed PBR, EncodeOcta :320-337,
// convertRgbToSh utils.cpp:45-49, invSigmoid utils.hpp:270, scale multiplier SceneManager.cpp:668), so that what crosses
// PCIe is the file's own bytes (76 or 48 instead of 96 per Gaussian) and the host only copies them into the file.
//
// The rows must be BYTE-identical to the reference's (and to m2s_write_ply's).  Everything except the logarithm is IEEE
// arithmetic in the reference's operation order (this file is compiled without contraction; `/` is HIP's correctly rounded
// division).  The logarithm is std::log(float) = the C library's logf on the machine that runs the reference; glibc's logf
// (since 2.27: Szabolcs Nagy's algorithm from Arm's optimized-routines, MIT licence) is NOT correctly rounded, so neither
// a correctly rounded log nor the device's __logf would reproduce its bytes.  logf_glibc() below restates that algorithm —
// 16-entry table of (1/c, log c), degree-3 polynomial, all in fp64 — with the constants read out of libm.so.6's
// __logf_data; an exhaustive comparison over all 2 139 095 039 positive finite floats against the host's logf shows zero
// mismatches (the logf_check program of the test infrastructure; with or without fused multiply-adds: the fp64 result is 29 bits wider than the fp32 it
// is rounded to).
#include "m2s_device.h"

#pragma clang fp contract(off)

namespace m2s {

__device__ __constant__ double kLogTab[16][2] = {
    { 0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2 }, { 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2 },
    { 0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2 }, { 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3 },
    { 0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3 }, { 0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3 },
    { 0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4 }, { 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4 },
    { 0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5 }, { 0x1p+0, 0x0p+0 },
    { 0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5 },  { 0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4 },
    { 0x1.b2036576afce6p-1, 0x1.526e57720db08p-3 },  { 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3 },
    { 0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2 },  { 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 },
};

__device__ __forceinline__ float logf_glibc(float x) {
    constexpr double kLn2 = 0x1.62e42fefa39efp-1, kA0 = -0x1.00ea348b88334p-2, kA1 = 0x1.5575b0be00b6ap-2, kA2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {   // zero, subnormal, negative, inf, NaN
        if (ix * 2u == 0u) return __uint_as_float(0xff800000u);             // log(+-0) = -inf
        if (ix == 0x7f800000u) return x;                                     // log(inf) = inf
        if (ix * 2u > 0xff000000u) return __uint_as_float(ix | 0x00400000u); // NaN in, quiet NaN out
        if (ix & 0x80000000u) return __uint_as_float(0xffc00000u);           // log(negative): x86's default NaN
        ix = __float_as_uint(x * 0x1p23f);                                   // subnormal: normalise
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const uint32_t i = (tmp >> 19) & 15u;
    const int k = (int)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0xff800000u);
    const double invc = kLogTab[i][0], logc = kLogTab[i][1];
    const double z = (double)__uint_as_float(iz);
    const double r = z * invc - 1.0;
    const double y0 = logc + (double)k * kLn2;
    const double r2 = r * r;
    double y = kA1 * r + kA2;
    y = kA0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

constexpr float kShC0 = 0.28209479177387814f;   // params.hpp:17 SH_COEFF0

// glm::clamp == min(max(x, lo), hi) with glm's comparison order
__device__ __forceinline__ float glm_clamp(float x, float lo, float hi) {
    const float mx = (x < lo) ? lo : x;
    return (hi < mx) ? hi : mx;
}
// parsers.cpp:370-375
__device__ __forceinline__ uint32_t to_byte(float v) { return (uint32_t)(uint8_t)(int)roundf(glm_clamp(v, 0.0f, 1.0f) * 255.0f); }
// utils.hpp:270 invSigmoid (std::clamp keeps NaN)
__device__ __forceinline__ float inv_sigmoid(float alpha) {
    alpha = (alpha < 0.0f) ? 0.0f : (1.0f < alpha) ? 1.0f : alpha;
    return -logf_glibc((1.0f / (alpha + 1e-8f)) - 1.0f);
}

__global__ void __launch_bounds__(256) k_encode_rows(const float4* __restrict__ rec, unsigned long long n, uint32_t format, float sm,
                                                     uint32_t* __restrict__ out) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4* g = rec + i * 6;
    const float4 pos = g[0], col = g[1], scl = g[2], nrm = g[3], rot = g[4], pbr = g[5];
    if (format == 2) {   // parsers.cpp:377-426
        uint32_t w[12];
        w[0] = __float_as_uint(pos.x); w[1] = __float_as_uint(pos.y); w[2] = __float_as_uint(pos.z);
        w[3] = to_byte(col.x) | (to_byte(col.y) << 8) | (to_byte(col.z) << 16) | (to_byte(col.w) << 24);
        w[4] = __float_as_uint(rot.x); w[5] = __float_as_uint(rot.y); w[6] = __float_as_uint(rot.z); w[7] = __float_as_uint(rot.w);
        const float min_xy = (scl.y < scl.x) ? scl.y : scl.x;                 // std::min(sx, sy), parsers.cpp:403
        w[8] = __float_as_uint(logf_glibc(scl.x * sm));
        w[9] = __float_as_uint(logf_glibc(scl.y * sm));
        w[10] = __float_as_uint(logf_glibc(min_xy * sm));
        // EncodeOcta, parsers.cpp:320-337 (OctWrap flips both components on a joint sign test)
        const float d = fabsf(nrm.x) + fabsf(nrm.y) + fabsf(nrm.z) + 1e-8f;
        const float nx = nrm.x / d, ny = nrm.y / d, nz = nrm.z / d;
        float ex = nx, ey = ny;
        if (!(nz >= 0.0f)) {
            const float s = (nx >= 0.0f && ny >= 0.0f) ? 1.0f : -1.0f;
            ex = (1.0f - fabsf(ny)) * s;
            ey = (1.0f - fabsf(nx)) * s;
        }
        ex = ex * 0.5f + 0.5f;
        ey = ey * 0.5f + 0.5f;
        const uint32_t ox = (uint32_t)(uint8_t)(int)glm_clamp(roundf(ex * 255.0f), 0.0f, 255.0f);
        const uint32_t oy = (uint32_t)(uint8_t)(int)glm_clamp(roundf(ey * 255.0f), 0.0f, 255.0f);
        w[11] = ox | (oy << 8) | (to_byte(pbr.y) << 16) | (to_byte(pbr.x) << 24);   // roughness, then metallic
        uint4* o = reinterpret_cast<uint4*>(out + i * 12);                           // 48 B rows: 16-byte aligned
        o[0] = make_uint4(w[0], w[1], w[2], w[3]);
        o[1] = make_uint4(w[4], w[5], w[6], w[7]);
        o[2] = make_uint4(w[8], w[9], w[10], w[11]);
        return;
    }
    // format 1, parsers.cpp:268-314: x y z | nx ny nz | f_dc | metallic roughness | opacity | scale | rot (w,x,y,z)
    uint32_t* o = out + i * 19;
    o[0] = __float_as_uint(pos.x); o[1] = __float_as_uint(pos.y); o[2] = __float_as_uint(pos.z);
    o[3] = __float_as_uint(nrm.x); o[4] = __float_as_uint(nrm.y); o[5] = __float_as_uint(nrm.z);
    o[6] = __float_as_uint((col.x - 0.5f) / kShC0); o[7] = __float_as_uint((col.y - 0.5f) / kShC0); o[8] = __float_as_uint((col.z - 0.5f) / kShC0);
    o[9] = __float_as_uint(pbr.x); o[10] = __float_as_uint(pbr.y);
    o[11] = __float_as_uint(inv_sigmoid(col.w));
    o[12] = __float_as_uint(logf_glibc(scl.x * sm)); o[13] = __float_as_uint(logf_glibc(scl.y * sm)); o[14] = __float_as_uint(logf_glibc(scl.z * sm));
    o[15] = __float_as_uint(rot.x); o[16] = __float_as_uint(rot.y); o[17] = __float_as_uint(rot.z); o[18] = __float_as_uint(rot.w);
}

void launch_encode_rows(const float4* rec, uint64_t n, uint32_t format, float sm, uint8_t* out, hipStream_t st) {
    if (!n) return;
    hipLaunchKernelGGL(k_encode_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rec, (unsigned long long)n, format, sm,
                       reinterpret_cast<uint32_t*>(out));
}

// (m2s_device.h: preload_*) makes the runtime load this file's code object now instead of inside the first launch
hipError_t preload_export() { hipFuncAttributes a; return hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k_encode_rows)); }

}  // namespace m2s
