// m2s_devfn.h — device functions shared by the conversion kernels (gfx950).  Restates
// converterGS.glsl:326-443 (geo_setup / geo_flat), the pinned rasteriser (raster_setup / row_span)
// and converterFS.glsl:44-104 with software trilinear sampling (shade_from_tri).
#pragma once
#include "m2s_device.h"
#include "m2s_exact.h"

#pragma clang fp contract(off)

namespace m2s {

// ============================================================================================
// small helpers
// ============================================================================================
__device__ __forceinline__ float len3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
__device__ __forceinline__ float len3sq(float x, float y, float z) { return (x * x + y * y) + z * z; }     // len3 = sqrt of this, same order
// 1.0f / len3(x, y, z), both operations correctly rounded (m2s_exact.h: the short sequences inside their proven range, the
// compiler's IEEE expansion elsewhere; the choice is wave-uniform)
__device__ __forceinline__ float inv_len3(float x, float y, float z) {
    const float q = len3sq(x, y, z);
    if (wave_all(in_sqrt_range(q))) return rcp_rn(sqrt_rn(q));
    return 1.0f / sqrtf(q);
}

// Inclusive prefix sum across the 64 lanes of a wave in six DPP adds (no LDS round trips: the shuffle-based version
// costs six ds_bpermute latencies): Hillis-Steele inside each row of 16 lanes (row_shr 1, 2, 4, 8; lanes without a
// source add 0), then lane 15 of rows 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2 and 3 (row_bcast:31).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int /*lane*/) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true);
    return v;
}
// (round 6: through the DPP scan above — six dependent ds_bpermute round trips took 0.5 us per call inside k_count_scan's tall-triangle
// loop, one call per 64 rows: the floor of the heterogeneous scene held its block's aggregate, and with it every block's look-back, for 18 us)
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v, 0), 63);
}
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the point and drains outstanding LDS traffic.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// exact floor(num/den), den > 0, |num| < 2^52, |num / den| < 2^44: an APPROXIMATE fp64 quotient + one integer correction step.
// Round 6: the quotient is num * (1/den) with the reciprocal from v_rcp_f64 and two Newton steps (relative error <= 2^-50 whatever
// the seed's accuracy from 2^-14 up), not an IEEE division (v_div_scale / v_div_fmas / v_div_fixup: ~60 issue slots of half-rate fp64
// each, six to twelve per triangle in the row walkers): |num y - num/den| <= 2^44 2^-50 = 2^-6 < 1, so floor() of it is the true
// floor or one off, and the remainder test below — exact integer arithmetic — settles which (tests/test_round6_math.py enumerates
// the argument with reciprocals perturbed far beyond that).  Divisions that share their divisor share the reciprocal (floordivmod_by).
__device__ __forceinline__ double rcp_f64(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, y, 1.0);
    y = __builtin_fma(e, y, y);
    e = __builtin_fma(-d, y, 1.0);
    return __builtin_fma(e, y, y);
}
// floor(num / den) and the remainder num - q den in [0, den), y = rcp_f64((double)den)
__device__ __forceinline__ long long floordivmod_by(long long num, long long den, double y, long long& rem) {
    long long q = (long long)floor((double)num * y);
    long long r = num - q * den;
    if (r < 0) { q -= 1; r += den; }
    else if (r >= den) { q += 1; r -= den; }
    rem = r;
    return q;
}
__device__ __forceinline__ long long floordiv_pos(long long num, long long den) {
    long long r;
    return floordivmod_by(num, den, rcp_f64((double)den), r);
}

// ============================================================================================
// geometry-shader restatement (converterGS.glsl:326-443)
// ============================================================================================
// The mesh table is written at upload and never while a conversion runs: a wave-uniform entry may be read through the constant
// address space (scalar loads, even after the kernel's own stores — see shade_from_tri).
typedef const __attribute__((address_space(4))) MeshParams* ConstMeshPtr;
__device__ __forceinline__ ConstMeshPtr kConstMesh(const MeshParams* p) { return (ConstMeshPtr)p; }

struct Geo {
    float xx, xy, xz;  // xAxis = normalised longest edge  (GS:345, 401)
    float nx, ny, nz;  // face normal                      (GS:347)
    float ou[3], ov[3];  // bbox-normalised orthogonal UVs  (GS:353-399)
};

__device__ __forceinline__ void geo_setup(const float p[9], const float* __restrict__ bmin,
                                          const float* __restrict__ bmax, Geo& g) {
    float e1x = p[3] - p[0], e1y = p[4] - p[1], e1z = p[5] - p[2];
    float e2x = p[6] - p[0], e2y = p[7] - p[1], e2z = p[8] - p[2];
    float e3x = p[6] - p[3], e3y = p[7] - p[4], e3z = p[8] - p[5];
    // the three edge lengths; the length of the edge the swap below selects is one of them (same operands, same value): the
    // shader's fourth square root (normalize(edge1), GS:345) is not recomputed
    const float q1 = len3sq(e1x, e1y, e1z), q2 = len3sq(e2x, e2y, e2z), q3 = len3sq(e3x, e3y, e3z);
    const bool fast_len = wave_all(in_sqrt_range3(q1, q2, q3));
    float l1, l2, l3;
    if (fast_len) { l1 = sqrt_rn(q1); l2 = sqrt_rn(q2); l3 = sqrt_rn(q3); }
    else { l1 = sqrtf(q1); l2 = sqrtf(q2); l3 = sqrtf(q3); }
    float ls = l1;
    // GS:333-342: strict >, else-if; second branch leaves edge2 untouched
    if (l2 > l1 && l2 > l3) {
        float tx = e1x, ty = e1y, tz = e1z;
        e1x = e2x; e1y = e2y; e1z = e2z;
        e2x = tx; e2y = ty; e2z = tz;
        ls = l2;
    } else if (l3 > l1 && l3 > l2) {
        e1x = e3x; e1y = e3y; e1z = e3z;
        ls = l3;
    }
    float inv;
    if (fast_len) inv = rcp_rn(ls);     // ls in [2^-48, 2^50]
    else inv = 1.0f / ls;
    g.xx = e1x * inv; g.xy = e1y * inv; g.xz = e1z * inv;
    float cx = g.xy * e2z - g.xz * e2y, cy = g.xz * e2x - g.xx * e2z, cz = g.xx * e2y - g.xy * e2x;
    inv = inv_len3(cx, cy, cz);
    g.nx = cx * inv; g.ny = cy * inv; g.nz = cz * inv;
    float ax = fabsf(g.nx), ay = fabsf(g.ny), az = fabsf(g.nz);
    // GS:360-396: (y,z) | (x,z) | (x,y) with strict compares and fall-through on ties
    const bool first = (ax > ay) && (ax > az);
    const bool second = !first && (ay > az);
    const bool useY_asA = first;             // A = 1 (y) only in the first branch, else 0 (x)
    const bool useY_asB = !first && !second; // B = 1 (y) only in the third branch, else 2 (z)
    float bminA = useY_asA ? bmin[1] : bmin[0], bmaxA = useY_asA ? bmax[1] : bmax[0];
    float bminB = useY_asB ? bmin[1] : bmin[2], bmaxB = useY_asB ? bmax[1] : bmax[2];
    float range = fmaxf(bmaxA - bminA, bmaxB - bminB);
    // u = rel / range: correctly rounded IEEE divisions, as the shader writes them (GS:362-363,375-376,388-389)
    // and as the oracle, which is checked bit for bit against the reference's GLSL run through glm, evaluates
    // them.  (rel * (1/range) differs in the last place for ~15 % of the vertices, and the UV->3D Jacobian of a
    // thin triangle amplifies that to 1e-3 in Scale.)  The six quotients share their divisor: one correctly rounded
    // reciprocal, then RN(rel / range) in three instructions each (div_rn, m2s_exact.h) where every operand lies in
    // the proven range — else, for the whole wave, the compiler's IEEE division.
    float ra[3], rb[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float pa = useY_asA ? p[3 * i + 1] : p[3 * i + 0];
        const float pb = useY_asB ? p[3 * i + 1] : p[3 * i + 2];
        ra[i] = pa - bminA;
        rb[i] = pb - bminB;
    }
    const float lo = fminf(fminf(fminf(fabsf(ra[0]), fabsf(ra[1])), fabsf(ra[2])), fminf(fminf(fabsf(rb[0]), fabsf(rb[1])), fminf(fabsf(rb[2]), range)));
    const float hi = fmaxf(fmaxf(fmaxf(fabsf(ra[0]), fabsf(ra[1])), fabsf(ra[2])), fmaxf(fmaxf(fabsf(rb[0]), fabsf(rb[1])), fmaxf(fabsf(rb[2]), range)));
    if (wave_all(lo >= kDivLo && hi <= kDivHi)) {
        const float y = rcp_rn(range);
#pragma unroll
        for (int i = 0; i < 3; i++) { g.ou[i] = div_rn(ra[i], range, y); g.ov[i] = div_rn(rb[i], range, y); }
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) { g.ou[i] = ra[i] / range; g.ov[i] = rb[i] / range; }
    }
}

// geo_setup with the mesh uniforms taken from a mesh-table entry.  MP is `const MeshParams*` (per-lane pointer: six vector
// loads per lane) or the same pointer in the CONSTANT address space (kConstMesh; wave-uniform mesh: scalar loads, once).
// Round 3: the triangle phase used the per-lane form everywhere — on BASELINE config 5 its descriptor loads (here and in
// tri_shade_setup) were 8 k of a 18 k-cycle round (tools/sparse_timing.py).
template <class MP>
__device__ __forceinline__ void geo_setup_mp(const float p[9], MP mp, Geo& g) {
    const float bmin[3] = { mp->bmin[0], mp->bmin[1], mp->bmin[2] }, bmax[3] = { mp->bmax[0], mp->bmax[1], mp->bmax[2] };
    geo_setup(p, bmin, bmax, g);
}

// ---- value arithmetic helpers -------------------------------------------------------------------
// Two classes of arithmetic (DESIGN.md "Pinned semantics"):
//  * DECISION arithmetic — everything that selects triangles, axes, pixels (geo_setup, raster_setup,
//    coverage): exact IEEE fp32, one rounding per operation, bit-identical to the CPU oracle.
//  * VALUE arithmetic — continuous outputs only (Scale, Quaternion magnitude, barycentrics, attribute
//    interpolation, texture filtering, normal mapping): FMA contraction and the hardware
//    reciprocal / rsqrt / sqrt / log2 (<= 1 ulp each); agrees with the oracle to ~1e-6 relative,
//    tolerance 1e-4.
// (float)int64 with a single rounding, for |v| < 2^53: the two halves are exact in fp64, their sum is exact,
// the final fp64 -> fp32 conversion rounds once (== the oracle's (float)E).  5 instructions instead of the
// compiler's ~12-instruction integer normalisation sequence.
__device__ __forceinline__ float i64_to_f32(long long v) {
    const double d = __builtin_fma((double)(int)(v >> 32), 4294967296.0, (double)(unsigned)v);
    return (float)d;
}

// Explicit fused multiply-add: VALUE arithmetic spells out every FMA instead of leaving contraction to the
// compiler, so that the fused and the multi-pass pipelines (two inlining contexts of the same functions)
// execute the same operations and produce bit-identical records.
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot3_(float ax, float ay, float az, float bx, float by, float bz) { return fma_(az, bz, fma_(ay, by, ax * bx)); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// flat outputs of the GS: Scale (GS:409-430) and Quaternion (GS:401-407, quat_cast GS:131-183)
__device__ __forceinline__ void geo_flat(const float p[9], const Geo& g, float& sx, float& sy, float4& rot) {
    // yAxis = normalize(cross(normal, xAxis)): exact, because the quat_cast branch below is a decision
    float cx = g.ny * g.xz - g.nz * g.xy, cy = g.nz * g.xx - g.nx * g.xz, cz = g.nx * g.xy - g.ny * g.xx;
    const float inv = inv_len3(cx, cy, cz);
    float yx = cx * inv, yy = cy * inv, yz = cz * inv;
    // m[c][r]: columns x, y, n
    const float m00 = g.xx, m01 = g.xy, m02 = g.xz;
    const float m10 = yx, m11 = yy, m12 = yz;
    const float m20 = g.nx, m21 = g.ny, m22 = g.nz;
    float fx = m00 - m11 - m22, fy = m11 - m00 - m22, fz = m22 - m00 - m11, fw = m00 + m11 + m22;
    int bi = 0;
    float fb = fw;
    if (fx > fb) { fb = fx; bi = 1; }
    if (fy > fb) { fb = fy; bi = 2; }
    if (fz > fb) { fb = fz; bi = 3; }
    {
        const float bv = fast_sqrt(fb + 1.0f) * 0.5f;
        const float mult = 0.25f * fast_rcp(bv);
        // off-diagonal sums/differences selected without divergent branches
        const float d0 = m12 - m21, d1 = m20 - m02, d2 = m01 - m10;
        const float s0 = m01 + m10, s1 = m20 + m02, s2 = m12 + m21;
        float qw, qx, qy, qz;
        if (bi == 0) { qw = bv; qx = d0 * mult; qy = d1 * mult; qz = d2 * mult; }
        else if (bi == 1) { qw = d0 * mult; qx = bv; qy = s0 * mult; qz = s1 * mult; }
        else if (bi == 2) { qw = d1 * mult; qx = s0 * mult; qy = bv; qz = s2 * mult; }
        else { qw = d2 * mult; qx = s1 * mult; qy = s2 * mult; qz = bv; }
        rot = make_float4(qw, qx, qy, qz);  // GS:407 stores (w,x,y,z)
        // Jacobian: UVMatrix[col][row], inverse2x2 (GS:206-220), multiplyMat2x3WithMat2x2 (GS:222-235)
        const float U00 = g.ou[1] - g.ou[0], U10 = g.ou[2] - g.ou[0];
        const float U01 = g.ov[1] - g.ov[0], U11 = g.ov[2] - g.ov[0];
        const float det = fma_(U00, U11, -(U01 * U10));
        const float invDet = det != 0.0f ? fast_rcp(det) : 0.0f;   // inverse2x2 returns mat2(0) for det == 0
        const float I00 = U11 * invDet, I10 = -U10 * invDet, I01 = -U01 * invDet, I11 = U00 * invDet;
        const float v0x = p[3] - p[0], v0y = p[4] - p[1], v0z = p[5] - p[2];
        const float v1x = p[6] - p[0], v1y = p[7] - p[1], v1z = p[8] - p[2];
        const float jux = fma_(v1x, I01, v0x * I00), juy = fma_(v1y, I01, v0y * I00), juz = fma_(v1z, I01, v0z * I00);
        const float jvx = fma_(v1x, I11, v0x * I10), jvy = fma_(v1y, I11, v0y * I10), jvz = fma_(v1z, I11, v0z * I10);
        sx = fast_sqrt(dot3_(jux, juy, juz, jux, juy, juz));
        sy = fast_sqrt(dot3_(jvx, jvy, jvz, jvx, jvy, jvz));
    }
}

// ============================================================================================
// pinned rasteriser: viewport transform, 24.8 snap (RNE), int64 edge functions, top-left rule
// ============================================================================================
struct Raster {
    int a[3], b[3];      // E_i(Px,Py) = a*Px + b*Py + c, interior positive; edge i opposite vertex i
    long long c[3];
    long long area2;
    int bias;            // bit i: boundary of edge i is inside
    int x0, x1, y0, y1;  // inclusive pixel bbox, clamped to the viewport
    int ext;             // max sub-pixel extent of the (unclamped) triangle bbox
};

constexpr float kGuardPx = 16384.0f;

// First half of the raster setup: viewport transform, 24.8 snap, pixel box.  false: a coordinate beyond the guard band (or NaN),
// or no pixel centre inside the box (the zero-area test is the caller's: raster_setup / raster_small).
struct RasterHead {
    int X[3], Y[3];      // snapped window coordinates (24.8)
    int x0, x1, y0, y1;  // inclusive pixel box, clamped to the viewport
    int ext;             // max sub-pixel extent of the (unclamped) triangle box
};
__device__ __forceinline__ bool raster_head(const Geo& g, uint32_t R, RasterHead& h) {
    const float half = (float)R * 0.5f;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float ndx = g.ou[i] * 2.0f - 1.0f, ndy = g.ov[i] * 2.0f - 1.0f;  // GS:439
        float xw = half * ndx + half, yw = half * ndy + half;            // glViewport(0,0,R,R)
        ok = ok && (fabsf(xw) < kGuardPx) && (fabsf(yw) < kGuardPx);     // false for NaN
        h.X[i] = (int)rintf(xw * 256.0f);
        h.Y[i] = (int)rintf(yw * 256.0f);
    }
    if (!ok) return false;
    const int xmin = min(h.X[0], min(h.X[1], h.X[2])), xmax = max(h.X[0], max(h.X[1], h.X[2]));
    const int ymin = min(h.Y[0], min(h.Y[1], h.Y[2])), ymax = max(h.Y[0], max(h.Y[1], h.Y[2]));
    h.ext = max(xmax - xmin, ymax - ymin);
    h.x0 = max((xmin - 128 + 255) >> 8, 0);
    h.x1 = min((xmax - 128) >> 8, (int)R - 1);
    h.y0 = max((ymin - 128 + 255) >> 8, 0);
    h.y1 = min((ymax - 128) >> 8, (int)R - 1);
    return h.x0 <= h.x1 && h.y0 <= h.y1;
}

__device__ __forceinline__ bool raster_setup(const Geo& g, uint32_t R, Raster& s) {
    RasterHead h;
    const bool box = raster_head(g, R, h);
    const int* X = h.X;
    const int* Y = h.Y;
    if (!box) return false;
    long long area2 = (long long)(X[1] - X[0]) * (Y[2] - Y[0]) - (long long)(Y[1] - Y[0]) * (X[2] - X[0]);
    if (area2 == 0) return false;
    const int sgn = area2 < 0 ? -1 : 1;  // no culling (ConversionPass.cpp:48)
    s.area2 = area2 < 0 ? -area2 : area2;
    s.bias = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int ia = (i + 1) % 3, ib = (i + 2) % 3;
        int dy = Y[ib] - Y[ia], dx = X[ib] - X[ia];
        s.a[i] = -dy * sgn;
        s.b[i] = dx * sgn;
        s.c[i] = ((long long)dy * X[ia] - (long long)dx * Y[ia]) * sgn;
        if (s.a[i] > 0 || (s.a[i] == 0 && s.b[i] > 0)) s.bias |= 1 << i;
    }
    s.ext = h.ext;
    s.x0 = h.x0; s.x1 = h.x1; s.y0 = h.y0; s.y1 = h.y1;
    return true;
}

// The same setup for a triangle whose sub-pixel extent is at most 2304 (so |a|, |b| <= 2304) in a pixel box of at most 8 x 8
// (k_fused3): every quantity fits 32 bits and every product 24 x 24 bits.  E_i(P) = a_i (Px - X_ia) + b_i (Py - Y_ia) — the edge
// function written from the edge's first vertex, equal to a_i Px + b_i Py + c_i of raster_setup — and the centre of the box-origin
// pixel lies inside the triangle's own box (x0 is the first centre >= xmin and x0 <= x1 puts it <= xmax), so both factors are at
// most 2304 and each product below 2^23.  No 64-bit and no full-rate-quarter 32 x 32 multiplies (v_mul_lo_u32, v_mad_i64_i32:
// 16 cycles each; raster_setup has twenty of them), same integers.
struct RasterSmall {
    int a[3], b[3];
    int e[3];            // E_i at the centre of pixel (x0, y0), exact
    int area2;           // > 0
    int bias;
};
__device__ __forceinline__ bool raster_small(const RasterHead& h, RasterSmall& s) {
    const int area = __mul24(h.X[1] - h.X[0], h.Y[2] - h.Y[0]) - __mul24(h.Y[1] - h.Y[0], h.X[2] - h.X[0]);
    if (area == 0) return false;
    const bool neg = area < 0;
    s.area2 = neg ? -area : area;
    const int Px0 = 256 * h.x0 + 128, Py0 = 256 * h.y0 + 128;
    s.bias = 0;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int ia = (i + 1) % 3, ib = (i + 2) % 3;
        const int dy = h.Y[ib] - h.Y[ia], dx = h.X[ib] - h.X[ia];
        s.a[i] = neg ? dy : -dy;
        s.b[i] = neg ? -dx : dx;
        s.e[i] = __mul24(s.a[i], Px0 - h.X[ia]) + __mul24(s.b[i], Py0 - h.Y[ia]);
        if (s.a[i] > 0 || (s.a[i] == 0 && s.b[i] > 0)) s.bias |= 1 << i;
    }
    return true;
}

// Coverage of the wave's SMALL triangles (one per lane; lanes without one pass small = false): bit 8 dy + dx of mhi:mlo = pixel
// (x0 + dx, y0 + dy) is covered (top-left rule through the bias, as everywhere).  The wave walks rows and columns TOGETHER
// (wave-uniform trip counts: the tallest and the widest box among its triangles), every lane shifting the sign of its own three
// edge functions into a row of bits: v_or3 + v_alignbit + three adds per pixel, loop control in scalar registers.  (The per-lane
// loop with a 64-bit mask this replaces took thirteen vector instructions per pixel and ran as long as the largest box as well.)
__device__ __forceinline__ void small_coverage(bool small, int w, int rows, const RasterSmall& rs, uint32_t& mlo, uint32_t& mhi) {
    const int ws = small ? w : 0, rws = small ? rows : 0;
    int wmax = 0, rmax = 0;
    while (__ballot(ws > wmax) != 0ull) ++wmax;
    while (__ballot(rws > rmax) != 0ull) ++rmax;
    mlo = 0; mhi = 0;
    if (!wmax) return;
    int e0 = 0, e1 = 0, e2 = 0, ax0 = 0, ax1 = 0, ax2 = 0, by0 = 0, by1 = 0, by2 = 0;
    if (small) {
        e0 = rs.e[0] + ((rs.bias >> 0) & 1) - 1; e1 = rs.e[1] + ((rs.bias >> 1) & 1) - 1; e2 = rs.e[2] + ((rs.bias >> 2) & 1) - 1;
        ax0 = rs.a[0] * 256; ax1 = rs.a[1] * 256; ax2 = rs.a[2] * 256;
        by0 = rs.b[0] * 256; by1 = rs.b[1] * 256; by2 = rs.b[2] * 256;
    }
    const uint32_t wmask = (1u << ws) - 1u;     // (0 for a lane without a small triangle)
    for (int dy = 0; dy < rmax; ++dy) {
        int r0 = e0, r1 = e1, r2 = e2;
        uint32_t outside = 0;                    // column dx at bit wmax - 1 - dx; 1 = some edge function negative
        for (int dx = 0; dx < wmax; ++dx) {
            outside = __builtin_amdgcn_alignbit(outside, (uint32_t)(r0 | r1 | r2), 31);   // (outside << 1) | sign bit
            r0 += ax0; r1 += ax1; r2 += ax2;
        }
        uint32_t row = (__brev(~outside) >> (32 - wmax)) & wmask;   // column dx at bit dx; 1 = covered
        if (dy >= rws) row = 0;
        if (dy < 4) mlo |= row << (8 * dy);
        else mhi |= row << (8 * dy - 32);
        e0 += by0; e1 += by1; e2 += by2;
    }
}

// covered pixels of row y form one interval [xa, xb] (empty if xa > xb): exact closed form
__device__ __forceinline__ void row_span(const Raster& s, int y, int& xa, int& xb) {
    const long long Py = 256ll * y + 128;
    long long lo = s.x0, hi = s.x1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const long long alpha = 256ll * s.a[i];
        // pixel x is inside edge i  <=>  alpha*x + beta >= 1   (E > 0, or E >= 0 on an owned boundary)
        const long long beta = 128ll * s.a[i] + (long long)s.b[i] * Py + s.c[i] + ((s.bias >> i) & 1);
        if (alpha > 0) {
            long long q = floordiv_pos(alpha - beta, alpha);  // ceil((1-beta)/alpha)
            lo = q > lo ? q : lo;
        } else if (alpha < 0) {
            long long q = floordiv_pos(beta - 1, -alpha);
            hi = q < hi ? q : hi;
        } else if (beta < 1) {
            hi = lo - 1;
        }
    }
    if (hi < lo) { xa = 0; xb = -1; }
    else { xa = (int)lo; xb = (int)hi; }
}

// Division-free row stepping for loops that visit a triangle's rows IN ORDER (row_span is the closed form for an
// arbitrary row: ~250 instructions, two fp64 divisions per edge; the wave-cooperative paths, which evaluate one row
// per lane, keep using it).  Per edge the bound of row y is floor(n(y) / D) with D = 256 |a| and a numerator that
// advances by a constant per row, so quotient and remainder are stepped Bresenham-style:
//     r += sr;  q += sq;  if (r >= D) { r -= D; ++q; }
// Exact: identical spans to row_span for every row (tests/test_gpu_parity.py::test_row_walker_*, and every count test).
struct RowWalker {
    long long q[3];      // floor(n/D) of the current row
    uint32_t r[3];       // n - q*D, in [0, D)
    int sq[3];           // floor(step / D):  |step| = 256 |b| < 2^31 and D >= 256  ->  |sq| < 2^23
    uint32_t sr[3];      // step - sq*D, in [0, D)
    uint32_t D[3];       // 256 |a|; 0 marks a horizontal edge
    long long beta[3];   // horizontal edges only: beta(y), stepped by 256 b
    int bstep[3];
    int lower;           // bit i: edge i bounds x from below (a > 0)
    int x0, x1;
};
// One edge of a walker: quotient / remainder of the row's numerator and of the per-row step, both by D = 256 |a| — ONE reciprocal,
// one code path for lower and upper bounds (the two used to be the sides of a divergent branch with two fp64 divisions each).
__device__ __forceinline__ void walker_edge(long long alpha, long long beta, long long bs, long long& q, uint32_t& r, int& sq, uint32_t& sr, uint32_t& D) {
    const bool pos = alpha > 0;
    const long long d = pos ? alpha : -alpha;             // 256 |a| < 2^31; 0: horizontal edge (handled through beta by the walkers)
    D = (uint32_t)d;
    const long long dd = d ? d : 1;
    const double y = rcp_f64((double)(uint32_t)dd);
    const long long n = pos ? alpha - beta : beta - 1;    // x >= floor((alpha - beta) / alpha)  |  x <= floor((beta - 1) / -alpha)
    const long long st = pos ? -bs : bs;
    long long r1, r2;
    const long long q1 = floordivmod_by(n, dd, y, r1), q2 = floordivmod_by(st, dd, y, r2);
    q = d ? q1 : 0; r = d ? (uint32_t)r1 : 0u; sq = d ? (int)q2 : 0; sr = d ? (uint32_t)r2 : 0u;
}
__device__ __forceinline__ void row_walker_init(const Raster& s, int y, RowWalker& w) {
    const long long Py = 256ll * y + 128;
    w.lower = 0;
    w.x0 = s.x0; w.x1 = s.x1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const long long alpha = 256ll * s.a[i];
        const long long beta = 128ll * s.a[i] + (long long)s.b[i] * Py + s.c[i] + ((s.bias >> i) & 1);
        const int bs = 256 * s.b[i];
        w.beta[i] = beta;
        w.bstep[i] = bs;
        if (alpha > 0) w.lower |= 1 << i;
        walker_edge(alpha, beta, (long long)bs, w.q[i], w.r[i], w.sq[i], w.sr[i], w.D[i]);
    }
}
// span of the current row, then advance to the next one
template <class W>
__device__ __forceinline__ void row_walker_next(W& w, int& xa, int& xb) {
    long long lo = w.x0, hi = w.x1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        if (w.D[i]) {
            if ((w.lower >> i) & 1) lo = w.q[i] > lo ? w.q[i] : lo;
            else hi = w.q[i] < hi ? w.q[i] : hi;
            uint32_t r = w.r[i] + w.sr[i];     // < 2^32: both < D < 2^31
            long long q = w.q[i] + w.sq[i];
            if (r >= w.D[i]) { r -= w.D[i]; q += 1; }
            w.r[i] = r; w.q[i] = q;
        } else {
            if (w.beta[i] < 1) hi = lo - 1;
            w.beta[i] += w.bstep[i];
        }
    }
    if (hi < lo) { xa = 0; xb = -1; }
    else { xa = (int)lo; xb = (int)hi; }
}

// ---- the walker in 32 bits (round 6) -----------------------------------------------------------------------------------------
// The per-wave timeline of k_count_scan (tools/timeline_probe.py) showed what its blocks wait for: ONE wave walking rows — 38 rows
// per lane in a column mesh 8-13 us, the two floor triangles of the heterogeneous scene 15 us — at ~100 instructions per row, most
// of them 64-bit compares / selects / adds on quotients that the closed form allows to reach 2^44, plus a second, divergent branch
// per edge for horizontal edges.  Here the same Bresenham steps on 32-bit quotients (an edge's x at a row of the triangle stays
// within a few thousand pixels unless the edge is nearly horizontal AND the triangle tall: the init checks every quotient over the
// rows it is going to visit and says so — the caller then takes the 64-bit walker, wave-uniformly) and WITHOUT the horizontal
// branch: an edge with a == 0 does not bound x at all, it passes or fails whole rows, beta(k) = beta(0) + k * 256 b * stride >= 1,
// i.e. it only narrows the range of rows [k0, k1] — one closed-form division at the init; in the loop such an edge is neutral
// (q = -2^30 as a lower bound, D = 1: never a carry).  ~33 full-rate instructions per row.  Spans identical to row_span / RowWalker
// for every visited row; rows outside [k0, k1] are empty (tests: every count test, tests/test_round6_math.py::test_walker32_*).
struct RowWalker32 {
    int q[3];
    uint32_t r[3];
    int sq[3];
    uint32_t sr[3];
    uint32_t D[3];       // 256 |a|, or 1 for an edge without an x bound
    int lower;           // bit i: edge i bounds x from below
    int x0, x1;
    int k0, k1;          // rows y + k0 * stride .. y + k1 * stride pass the horizontal edges (k0 > k1: none); the walker stands at k0
};
constexpr long long kWalk32 = 1ll << 29;
// rows y, y + stride, ..., y + (n - 1) * stride of triangle s.  false: a quotient may leave the 32-bit walker's range.
__device__ __forceinline__ bool row_walker32_init(const Raster& s, int y, int stride, int n, RowWalker32& w) {
    long long k0 = 0, k1 = (long long)n - 1;
    {   // (a triangle with two horizontal edges has no area and never gets here: ONE division)
        const bool h0 = s.a[0] == 0, h1 = s.a[1] == 0, h2 = s.a[2] == 0;
        if (h0 || h1 || h2) {
            // (selected by masks, not by an index: an indexed pick among s.b[] / s.c[] sends the whole Raster to scratch memory)
            const int bh = (h0 ? s.b[0] : 0) | (h1 ? s.b[1] : 0) | (h2 ? s.b[2] : 0);
            const long long ch = (h0 ? s.c[0] : 0ll) | (h1 ? s.c[1] : 0ll) | (h2 ? s.c[2] : 0ll);
            const int bit = (h0 ? s.bias : 0) | (h1 ? s.bias >> 1 : 0) | (h2 ? s.bias >> 2 : 0);
            const long long beta = (long long)bh * (256ll * y + 128) + ch + (bit & 1);
            const long long bs = 256ll * bh * stride;
            // beta(k) is linear: if the first and the last row pass, every row does — the usual case (the edge is the triangle's top
            // or bottom; only a row whose centres lie ON it, or a box clamped by the viewport, can fail): no division then
            if (beta < 1 || beta + k1 * bs < 1) {
                if (bs != 0) {
                    const long long f = floordiv_pos(bs > 0 ? bs - beta : beta - 1, bs > 0 ? bs : -bs);
                    if (bs > 0) k0 = f > k0 ? f : k0;       // k >= ceil((1 - beta) / bs)
                    else k1 = f < k1 ? f : k1;              // k <= floor((beta - 1) / -bs)
                } else k1 = -1;
            }
        }
    }
    w.x0 = s.x0; w.x1 = s.x1;
    w.lower = 0;
    if (k0 > k1) { k0 = 0; k1 = -1; }
    w.k0 = (int)k0; w.k1 = (int)k1;
    const int yy = y + (int)k0 * stride;
    const long long steps = k1 - k0;                      // -1: no row at all
    const long long Py = 256ll * yy + 128;
    bool safe = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const long long alpha = 256ll * s.a[i];
        const long long beta = 128ll * s.a[i] + (long long)s.b[i] * Py + s.c[i] + ((s.bias >> i) & 1);
        const long long bs = 256ll * s.b[i] * stride;
        long long q;
        walker_edge(alpha, beta, bs, q, w.r[i], w.sq[i], w.sr[i], w.D[i]);
        if (alpha > 0 || alpha == 0) w.lower |= 1 << i;
        if (alpha == 0) { q = -2 * kWalk32; w.D[i] = 1u; }                  // (walker_edge left r = sq = sr = 0)
        else if (steps >= 0) {
            // the last row's quotient lies in [qe, qe + steps], qe = q + steps * sq: |sq| < 2^29 and steps < 2^12, so the fp32 estimate
            // of |qe| is off by less than 2^19 — tested against 2^28, the exact q against 2^29 by its sign extension
            const float qe = fabsf((float)q + (float)(int)steps * (float)w.sq[i]);
            safe = safe && (q >> 29) == (q >> 63) && qe < 268435456.0f;
        }
        w.q[i] = (int)q;
    }
    return safe || steps < 0;
}
// floor(x / D) and the remainder for 0 <= x < 4096 D + D, D < 2^31 (a walker jumping up to 4095 rows ahead): the quotient stays below 2^13, so an
// fp32 estimate (three roundings of 2^-24 relative each) is off by at most one and the exact remainder settles it — no fp64.
__device__ __forceinline__ uint32_t jump_quot(unsigned long long x, uint32_t D, uint32_t& rem) {
    uint32_t e = (uint32_t)((float)x * __builtin_amdgcn_rcpf((float)D));
    long long r = (long long)x - (long long)((unsigned long long)e * D);
    if (r < 0) { e -= 1u; r += D; }
    else if (r >= (long long)D) { e += 1u; r -= D; }
    rem = (uint32_t)r;
    return e;
}
// the walker d rows further down (0 <= d < 4096; the init has vouched for every quotient on the way: 32-bit wrap-around arithmetic is exact)
__device__ __forceinline__ void row_walker32_jump(RowWalker32& w, uint32_t d) {
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint32_t rem;
        const uint32_t e = jump_quot((unsigned long long)w.r[i] + (unsigned long long)d * w.sr[i], w.D[i], rem);
        w.q[i] = (int)((uint32_t)w.q[i] + d * (uint32_t)w.sq[i] + e);
        w.r[i] = rem;
    }
}
// span of the current row (empty iff xb < xa; xb - xa + 1 >= -2^31 / 2), then advance by one step
__device__ __forceinline__ void row_walker32_next(RowWalker32& w, int& xa, int& xb) {
    int lo = w.x0, hi = w.x1;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const bool l = (w.lower >> i) & 1;
        const int ql = l ? w.q[i] : (int)0x80000000, qh = l ? 0x7FFFFFFF : w.q[i];
        lo = ql > lo ? ql : lo;
        hi = qh < hi ? qh : hi;
        const uint32_t t = w.r[i] + w.sr[i];                                  // < 2^32: both < D < 2^31
        const bool c = t >= w.D[i];
        w.r[i] = c ? t - w.D[i] : t;
        w.q[i] += w.sq[i] + (c ? 1 : 0);
    }
    xa = lo; xb = hi;
}

__device__ __forceinline__ Raster shfl_raster(const Raster& s, int src) {
    Raster r;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        r.a[i] = __shfl(s.a[i], src);
        r.b[i] = __shfl(s.b[i], src);
        r.c[i] = __shfl(s.c[i], src);
    }
    r.area2 = 0;
    r.bias = __shfl(s.bias, src);
    r.ext = 0;
    r.x0 = __shfl(s.x0, src); r.x1 = __shfl(s.x1, src);
    r.y0 = __shfl(s.y0, src); r.y1 = __shfl(s.y1, src);
    return r;
}

// ============================================================================================
// scene access
// ============================================================================================
__device__ __forceinline__ void load_positions(const TriPlanes& tp, uint32_t t, float p[9]) {
    float4 a0 = tp.A0[t], a1 = tp.A1[t];
    float a2 = tp.A2[t];
    p[0] = a0.x; p[1] = a0.y; p[2] = a0.z; p[3] = a0.w;
    p[4] = a1.x; p[5] = a1.y; p[6] = a1.z; p[7] = a1.w;
    p[8] = a2;
}

// mesh of GLOBAL triangle gt: last m with mesh_first[m] <= gt
__device__ __forceinline__ uint32_t find_mesh(const SceneDev& sc, uint32_t gt) {
    uint32_t lo = 0, hi = sc.n_meshes;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (sc.mesh_first[mid] <= gt) lo = mid; else hi = mid;
    }
    return lo;
}

// Mesh of the range of local triangles [t0, t_last] (t0 a multiple of 8, wave-uniform) and whether the whole range lies in it:
// one 8-byte load through the constant address space (the table is written at upload, never during a conversion).
__device__ __forceinline__ uint32_t mesh_of_range(const SceneDev& sc, uint32_t t0, uint32_t t_last, bool& uniform) {
    const __attribute__((address_space(4))) uint32_t* q = (const __attribute__((address_space(4))) uint32_t*)sc.mesh_of8;
    const uint32_t m = q[2u * (t0 >> 3)], end = q[2u * (t0 >> 3) + 1u];
    uniform = end > t_last;
    return m;
}

__device__ __forceinline__ bool setup_raster_for(const SceneDev& sc, uint32_t t, uint32_t mesh_hint, bool uniform_mesh,
                                                 uint32_t R, Raster& rs) {
    float p[9];
    load_positions(sc.tri, t, p);
    uint32_t m = uniform_mesh ? mesh_hint : find_mesh(sc, sc.tri_first + t);
    const MeshParams* mp = sc.meshes + m;
    Geo g;
    geo_setup(p, mp->bmin, mp->bmax, g);
    return raster_setup(g, R, rs);
}

// ============================================================================================
// fragment-shader restatement (converterFS.glsl:44-104) with software trilinear sampling
// (sampler state glUtils.cpp:292-313: REPEAT, LINEAR_MIPMAP_LINEAR / LINEAR, levels 0..4).
// VALUE arithmetic throughout (see geo_flat).
// ============================================================================================
constexpr float kUnorm8 = 0.003921568859368563f;  // fp32 nearest to 1/255

struct __attribute__((packed, aligned(4))) ComboPair { uint32_t v[6]; };  // {A,N,M} of texel i and of texel i+1

// Texel reads name the GLOBAL address space explicitly.  The texel base pointers come out of the mesh table (a load), so
// to the compiler they are generic pointers and every texel fetch became a FLAT load (flat_load_dwordx4 + 64-bit address
// arithmetic per lane; a flat load ties up the LDS counter as well: round 1's ISA had 32 of them per kernel).  With the
// address space spelled out they are global loads with the (wave-uniform) base in scalar registers and a 32-bit offset per lane.
typedef const __attribute__((address_space(1))) char* GlobalBytes;
__device__ __forceinline__ GlobalBytes as_global(const void* p) { return (GlobalBytes)p; }
__device__ __forceinline__ ComboPair ld_combo(GlobalBytes base, uint32_t dword_off) {
    const __attribute__((address_space(1))) uint32_t* p = reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(base + (size_t)(dword_off * 4u));
    ComboPair r;
#pragma unroll
    for (int i = 0; i < 6; ++i) r.v[i] = p[i];     // merged into one dwordx4 + one dwordx2 load
    return r;
}
__device__ __forceinline__ uint32_t ld_texel(GlobalBytes base, uint32_t texel) {
    return *reinterpret_cast<const __attribute__((address_space(1))) uint32_t*>(base + (size_t)(texel * 4u));
}


// Per-triangle constants of the fragment stage: computed once per triangle (fused kernel: in the
// triangle phase, kept in LDS; multi-pass emit: read from the TriSetup record k_count_scan wrote).  80 bytes = five float4.
struct TriShade {
    int a1, b1, a2, b2;      // edge functions opposite vertex 1 / 2 ...
    long long e1, e2;        // ... and their exact values at the bbox origin pixel (x0,y0)
    float inva, sx, sy, lod0;  // 1/area2 (IEEE), Scale.xy (GS:423-430), LOD lambda of map 0
    float4 rot;              // Quaternion (w,x,y,z)
    float lod1, lod2;        // LOD lambda of maps 1, 2 (constant per triangle: UV is affine in the window)
                             // -- meshes with a combo texture: lod0 = level blend weight, lod1/lod2 = bits of the two
                             //    level offsets, mesh[31:24] = l0 | l1 << 4 (see tri_shade_setup)
    uint32_t org;            // y0 << 12 | x0
    uint32_t mesh;           // [23:0] mesh index (read by waves that straddle a mesh boundary)
};
static_assert(sizeof(TriShade) == 80, "TriShade must be five float4");

// The same record for triangles whose pixel box is at most 8 x 8 and whose sub-pixel extent is at most 2304 (the only ones the
// sparse kernel shades itself, m2s_sparse.hip): edge coefficients are differences of snapped coordinates (|.| <= 2304: int16)
// and the edge values at the box origin stay below 2^24 (int32).  Same field names, same values: shade_from_tri is a template.
struct TriShadeS {
    short a1, b1, a2, b2;
    int e1, e2;
    float inva, sx, sy, lod0;
    float4 rot;
    float lod1, lod2;
    uint32_t org;
    uint32_t mesh;
};
static_assert(sizeof(TriShadeS) == 64, "TriShadeS must be four float4");

__device__ __forceinline__ float lod_from_grad(float fw, float fh, float dudx, float dvdx, float dudy, float dvdy) {
    const float sx = dudx * fw, tx = dvdx * fh, sy = dudy * fw, ty = dvdy * fh;
    const float r2 = fmaxf(fma_(tx, tx, sx * sx), fma_(ty, ty, sy * sy));
    return 0.5f * fast_log2(r2);   // log2(sqrt(r2))
}

// Everything of the per-triangle fragment constants that does not depend on how the edge functions are stored: Scale, Quaternion,
// the levels of detail.  Needs ts.inva; a1 .. b2 = the coefficients of edges 1 and 2.
template <class MP, class TS>   // MP: const MeshParams* in the generic or the constant address space (kConstMesh); TS: TriShade / TriShadeS
__device__ __forceinline__ void tri_shade_rest(const float p[9], const Geo& g, int a1, int b1, int a2, int b2, MP mp,
                                               float4 b0, float2 b1uv, TS& ts) {
    ts.mesh = 0;
    const float A1 = (float)a1 * 256.0f * ts.inva, B1 = (float)b1 * 256.0f * ts.inva;
    const float A2 = (float)a2 * 256.0f * ts.inva, B2 = (float)b2 * 256.0f * ts.inva;
    geo_flat(p, g, ts.sx, ts.sy, ts.rot);
    ts.lod0 = ts.lod1 = ts.lod2 = 0.0f;
    const auto ta = &mp->tex[0];
    const auto tn = &mp->tex[1];
    const auto tm = &mp->tex[2];
    const bool hasA = ta->texels != nullptr, hasN = tn->texels != nullptr, hasM = tm->texels != nullptr;
    if (hasA || hasN || hasM) {
        // UV is affine in window space (all w = 1, GS:439): d(lambda_i)/dx = A_i, d/dy = B_i per pixel
        const float du1 = b0.z - b0.x, du2 = b1uv.x - b0.x, dv1 = b0.w - b0.y, dv2 = b1uv.y - b0.y;
        const float dudx = fma_(A2, du2, A1 * du1), dvdx = fma_(A2, dv2, A1 * dv1);
        const float dudy = fma_(B2, du2, B1 * du1), dvdy = fma_(B2, dv2, B1 * dv1);
        if (hasA) ts.lod0 = lod_from_grad((float)ta->w, (float)ta->h, dudx, dvdx, dudy, dvdy);
        if (hasN) ts.lod1 = (hasA && tn->w == ta->w && tn->h == ta->h) ? ts.lod0
                          : lod_from_grad((float)tn->w, (float)tn->h, dudx, dvdx, dudy, dvdy);
        if (hasM) ts.lod2 = (hasA && tm->w == ta->w && tm->h == ta->h) ? ts.lod0
                          : (hasN && tm->w == tn->w && tm->h == tn->h) ? ts.lod1
                          : lod_from_grad((float)tm->w, (float)tm->h, dudx, dvdx, dudy, dvdy);
        if (mp->combo.texels != nullptr) {
            // Interleaved maps (one size, one LOD): everything the sampler derives from the level of detail is a
            // per-TRIANGLE constant, so it is derived here once instead of once per fragment: the two mip levels, the
            // blend weight and the dword offsets of the levels inside the combo chain.  TriShade then carries
            //   lod0 = blend weight f (0: single level), lod1 / lod2 = bit patterns of the level offsets,
            //   mesh[31:24] = l0 | l1 << 4.
            const uint32_t nl = ta->n_levels;
            const float q = (float)(nl - 1), lambda = ts.lod0;
            float d = 0.0f, f = 0.0f;
            if (lambda > 0.0f) {   // false for NaN: magnification
                if (lambda >= q) d = q;
                else { d = floorf(lambda); f = lambda - d; }
            }
            const uint32_t l0 = (uint32_t)d, l1 = min(l0 + 1u, nl - 1u);
            const uint32_t c1 = mp->combo.coff[1], c2 = mp->combo.coff[2], c3 = mp->combo.coff[3], c4 = mp->combo.coff[4];
            const uint32_t off0 = l0 == 0 ? 0u : l0 == 1 ? c1 : l0 == 2 ? c2 : l0 == 3 ? c3 : c4;
            const uint32_t off1 = l1 == 0 ? 0u : l1 == 1 ? c1 : l1 == 2 ? c2 : l1 == 3 ? c3 : c4;
            ts.lod0 = f;
            ts.lod1 = __uint_as_float(off0);
            ts.lod2 = __uint_as_float(off1);
            ts.mesh = (l0 | (l1 << 4)) << 24;
        }
    }
}

// p: positions; b0/b1: uv planes of the triangle.  Needs a valid Raster.
template <class MP>   // const MeshParams* in the generic or the constant address space (kConstMesh)
__device__ __forceinline__ void tri_shade_setup(const float p[9], const Geo& g, const Raster& rs, MP mp,
                                                float4 b0, float2 b1, TriShade& ts) {
    // The barycentrics feed the texture coordinates, where any rounding difference is amplified by the
    // texture size and contrast: they follow the oracle's exact operation sequence
    // lambda_i = float(E_i) * (1 / float(area2)) with exact integer E_i (DECISION-class arithmetic).
    {
#pragma clang fp contract(off)
        ts.inva = rcp_rn((float)rs.area2);     // = 1.0f / (float)area2 (m2s_exact.h): 1 <= area2 < 2^63, inside rcp_rn's range
    }
    const long long Px0 = 256ll * rs.x0 + 128, Py0 = 256ll * rs.y0 + 128;
    ts.a1 = rs.a[1]; ts.b1 = rs.b[1]; ts.a2 = rs.a[2]; ts.b2 = rs.b[2];
    ts.e1 = (long long)rs.a[1] * Px0 + (long long)rs.b[1] * Py0 + rs.c[1];
    ts.e2 = (long long)rs.a[2] * Px0 + (long long)rs.b[2] * Py0 + rs.c[2];
    ts.org = ((uint32_t)rs.y0 << 12) | (uint32_t)rs.x0;
    tri_shade_rest(p, g, rs.a[1], rs.b[1], rs.a[2], rs.b[2], mp, b0, b1, ts);
}

// TriShadeS of one small triangle, straight from its 32-bit raster setup (raster_small; the same values tri_shade_setup computes
// in 64 bits and the sparse kernel narrows: m2s_devfn.h, TriShadeS)
template <class MP>
__device__ __forceinline__ void tri_shade_small(const float p[9], const Geo& g, const RasterHead& h, const RasterSmall& rs, MP mp, float4 b0, float2 b1,
                                                uint32_t m, TriShadeS& c) {
    {
#pragma clang fp contract(off)
        c.inva = rcp_rn((float)rs.area2);    // = 1.0f / (float)area2 (m2s_exact.h); 1 <= area2 <= 2304^2 < 2^24: the conversion is exact, like (float)(long long) of the same value
    }
    c.a1 = (short)rs.a[1]; c.b1 = (short)rs.b[1]; c.a2 = (short)rs.a[2]; c.b2 = (short)rs.b[2];
    c.e1 = rs.e[1]; c.e2 = rs.e[2];
    c.org = ((uint32_t)h.y0 << 12) | (uint32_t)h.x0;
    tri_shade_rest(p, g, rs.a[1], rs.b[1], rs.a[2], rs.b[2], mp, b0, b1, c);
    c.mesh |= m;
}

// Texel addresses (in texels from the start of the mip chain) and bilinear weights of one level.
struct TexTap {
    uint32_t o00, o10, o01, o11;
    float w00, w10, w01, w11;
};

// uf, vf in [0,1] (REPEAT already applied).  W,H,level_off describe the level.
__device__ __forceinline__ void tex_tap(uint32_t W, uint32_t H, uint32_t level_off, float uf, float vf, TexTap& t) {
    float up, vp;
    {   // texel coordinates and fractional weights: exact oracle sequence (no FMA)
#pragma clang fp contract(off)
        up = uf * (float)W - 0.5f;
        vp = vf * (float)H - 0.5f;
    }
    const float fi = floorf(up), fj = floorf(vp);
    const float a = up - fi, b = vp - fj;
    int i0 = (int)fi, j0 = (int)fj;  // in [-1, W-1]
    int i1 = i0 + 1, j1 = j0 + 1;
    i0 = i0 < 0 ? i0 + (int)W : i0;
    j0 = j0 < 0 ? j0 + (int)H : j0;
    i1 = i1 >= (int)W ? i1 - (int)W : i1;
    j1 = j1 >= (int)H ? j1 - (int)H : j1;
    // (rows and widths are below 2^24: the full-rate 24-bit multiply, not the quarter-rate v_mul_lo_u32)
    const uint32_t r0 = level_off + __umul24((uint32_t)j0, W), r1 = level_off + __umul24((uint32_t)j1, W);
    t.o00 = r0 + (uint32_t)i0; t.o10 = r0 + (uint32_t)i1;
    t.o01 = r1 + (uint32_t)i0; t.o11 = r1 + (uint32_t)i1;
    const float na = 1.0f - a, nb = 1.0f - b;
    t.w00 = na * nb; t.w10 = a * nb; t.w01 = na * b; t.w11 = a * b;
}

// The LOD-dependent sampling state shared by every map of the same size: one or two levels.
struct TexState {
    TexTap lo, hi;
    float f;     // weight of `hi` (0 = single level)
};

template <class TD>   // const TexDesc* in the generic or in the constant address space (see shade_from_tri)
__device__ __forceinline__ void tex_state(TD t, float uf, float vf, float lambda, TexState& st) {
    const uint32_t nl = t->n_levels, w = t->w, h = t->h;
    const float q = (float)(nl - 1);
    float d = 0.0f, f = 0.0f;
    if (lambda > 0.0f) {   // false for NaN: magnification
        if (lambda >= q) d = q;
        else { d = floorf(lambda); f = lambda - d; }
    }
    const uint32_t l0 = (uint32_t)d, l1 = min(l0 + 1u, nl - 1u);
    // level offsets are wave-uniform scalars; pick per lane
    const uint32_t o1 = t->off[1], o2 = t->off[2], o3 = t->off[3], o4 = t->off[4];
    const uint32_t off0 = l0 == 0 ? 0u : l0 == 1 ? o1 : l0 == 2 ? o2 : l0 == 3 ? o3 : o4;
    const uint32_t off1 = l1 == 0 ? 0u : l1 == 1 ? o1 : l1 == 2 ? o2 : l1 == 3 ? o3 : o4;
    tex_tap(max(1u, w >> l0), max(1u, h >> l0), off0, uf, vf, st.lo);
    st.f = f;
    if (__ballot(f != 0.0f) != 0ull) tex_tap(max(1u, w >> l1), max(1u, h >> l1), off1, uf, vf, st.hi);
    else st.hi = st.lo;
}

// The eight texels of one map's trilinear footprint (two levels x 2 x 2), requested together; filtered later.
struct TexFetch { uint32_t lo[4], hi[4]; };
__device__ __forceinline__ void tex_issue(const uint32_t* __restrict__ texels, const TexState& st, TexFetch& tf) {
    const GlobalBytes gb = as_global(texels);
    tf.lo[0] = ld_texel(gb, st.lo.o00); tf.lo[1] = ld_texel(gb, st.lo.o10); tf.lo[2] = ld_texel(gb, st.lo.o01); tf.lo[3] = ld_texel(gb, st.lo.o11);
    tf.hi[0] = ld_texel(gb, st.hi.o00); tf.hi[1] = ld_texel(gb, st.hi.o10); tf.hi[2] = ld_texel(gb, st.hi.o01); tf.hi[3] = ld_texel(gb, st.hi.o11);
}
template <int NCH>  // number of leading channels wanted (RGBA byte order)
__device__ __forceinline__ void tap_filter(const uint32_t t[4], const TexTap& tp, float out[NCH]) {
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) {
        const float c00 = (float)((t[0] >> (8 * ch)) & 255u), c10 = (float)((t[1] >> (8 * ch)) & 255u);
        const float c01 = (float)((t[2] >> (8 * ch)) & 255u), c11 = (float)((t[3] >> (8 * ch)) & 255u);
        out[ch] = fma_(tp.w11, c11, fma_(tp.w01, c01, fma_(tp.w10, c10, tp.w00 * c00)));
    }
}
// (1 - f) lo + f hi, always in the two-level form: a lane that blends nothing has f = 0 and st.hi == st.lo, and
// fma(0, hi, 1 * lo) is lo exactly — the bits of the one-level form, without a branch between the loads and their use
// (round 3: the second level used to be fetched inside `if (any lane blends)`, after the first level had been consumed: two
// dependent memory round trips per map, see combo_issue)
template <int NCH>
__device__ __forceinline__ void tex_finish(const TexFetch& tf, const TexState& st, float out[NCH]) {
    float lo[NCH], hi[NCH];
    tap_filter<NCH>(tf.lo, st.lo, lo);
    tap_filter<NCH>(tf.hi, st.hi, hi);
    const float f = st.f, nf = 1.0f - st.f;
#pragma unroll
    for (int ch = 0; ch < NCH; ch++) out[ch] = fma_(f, hi[ch], nf * lo[ch]) * kUnorm8;
}

// ---- combo path -----------------------------------------------------------------------------------

// Address + weights of one level's footprint in the combo layout.
struct ComboTap {
    uint32_t o0, o1;   // dword offsets of the two row pairs (texel i0 and its right neighbour, rows j0 and j1)
    float w00, w10, w01, w11;
};

__device__ __forceinline__ void combo_tap(uint32_t level_off, uint32_t W, uint32_t H, float uf, float vf, ComboTap& t) {
    float up, vp;
    {   // exact oracle sequence (no FMA), as in tex_tap
#pragma clang fp contract(off)
        up = uf * (float)W - 0.5f;
        vp = vf * (float)H - 0.5f;
    }
    const float fi = floorf(up), fj = floorf(vp);
    const float a = up - fi, b = vp - fj;
    int i0 = (int)fi, j0 = (int)fj;  // in [-1, W-1]
    int j1 = j0 + 1;
    i0 = i0 < 0 ? i0 + (int)W : i0;  // i0 == W-1 reads the wrapped extra column as its right neighbour
    j0 = j0 < 0 ? j0 + (int)H : j0;
    j1 = j1 >= (int)H ? j1 - (int)H : j1;
    const uint32_t stride = W + 1;
    // (rows and strides are below 2^24: the full-rate 24-bit multiply, not the quarter-rate v_mul_lo_u32)
    t.o0 = level_off + (__umul24((uint32_t)j0, stride) + (uint32_t)i0) * 3u;
    t.o1 = level_off + (__umul24((uint32_t)j1, stride) + (uint32_t)i0) * 3u;
    const float na = 1.0f - a, nb = 1.0f - b;
    t.w00 = na * nb; t.w10 = a * nb; t.w01 = na * b; t.w11 = a * b;
}

// un-normalised bilinear sums for the nine channels we need:
// albedo rgba (0-3), normal rgb (4-6), roughness = MR.g (7), metallic = MR.b (8)
__device__ __forceinline__ void combo_filter(const ComboPair& r0, const ComboPair& r1, const ComboTap& t, float out[9]) {
#define M2S_CH(word, sh) fma_(t.w11, (float)((r1.v[(word) + 3] >> (sh)) & 255u), fma_(t.w01, (float)((r1.v[word] >> (sh)) & 255u), \
                          fma_(t.w10, (float)((r0.v[(word) + 3] >> (sh)) & 255u), t.w00 * (float)((r0.v[word] >> (sh)) & 255u))))
    out[0] = M2S_CH(0, 0); out[1] = M2S_CH(0, 8); out[2] = M2S_CH(0, 16); out[3] = M2S_CH(0, 24);
    out[4] = M2S_CH(1, 0); out[5] = M2S_CH(1, 8); out[6] = M2S_CH(1, 16);
    out[7] = M2S_CH(2, 8); out[8] = M2S_CH(2, 16);
#undef M2S_CH
}

__device__ __forceinline__ float frac_repeat(float u) {
    float f = u - floorf(u);
    f = (f >= 0.0f) ? f : 0.0f;   // NaN -> 0
    return fminf(f, 1.0f);
}

// the requests of one fragment's texel footprint (both mip levels) and what the filter needs afterwards
struct ComboFetch {
    ComboTap tlo, thi;
    ComboPair a0, a1, b0, b1;
    float f;
};
template <class MP, class TD, class TS>
__device__ __forceinline__ void combo_issue(MP mp, TD t, float uf, float vf, const TS& ts, ComboFetch& cf) {
    // level selection was done per triangle (tri_shade_setup)
    const uint32_t w = t->w, h = t->h;
    const float f = ts.lod0;
    const uint32_t off0 = __float_as_uint(ts.lod1), off1 = __float_as_uint(ts.lod2);
    const uint32_t l0 = (ts.mesh >> 24) & 15u, l1 = ts.mesh >> 28;
    const uint32_t* __restrict__ base = mp->combo.texels;
    ComboTap& tlo = cf.tlo;
    ComboTap& thi = cf.thi;
    cf.f = f;
    combo_tap(off0, max(1u, w >> l0), max(1u, h >> l0), uf, vf, tlo);
    combo_tap(off1, max(1u, w >> l1), max(1u, h >> l1), uf, vf, thi);
    // ALL row reads of both levels are requested before the first one is consumed: one memory round trip instead of two.
    // Round 3: that was the intent since round 1, but NOT what ran.  The second level used to sit inside `if (any lane of the wave
    // blends two levels)`; the compiler hoisted the byte -> float conversions of the FIRST level — common to both branches —
    // above that branch, and with them an s_waitcnt vmcnt(0): the second level's reads were issued only after the first level
    // had arrived (k_fused2 ISA: 36 v_cvt_f32_ubyte and a full wait between the two groups of loads) = two dependent round
    // trips in the longest latency chain of the kernel.  There is no branch any more: both levels are always read and
    // filtered; a lane that blends nothing has f = 0, its second level gets weight 0 and contributes exactly +0 (same bits
    // as the one-level form: lo + 0), at the price of ~100 vector instructions in waves that only magnify.
    // 32-bit byte offsets from the (scalar) base: saddr + voffset addressing, no 64-bit address arithmetic per lane
    // (m2s_upload_scene only builds a combo texture whose size fits)
    const GlobalBytes gb = as_global(base);
    cf.a0 = ld_combo(gb, tlo.o0);
    cf.a1 = ld_combo(gb, tlo.o1);
    cf.b0 = ld_combo(gb, thi.o0);
    cf.b1 = ld_combo(gb, thi.o1);
#ifdef M2S_FUSED3_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}
__device__ __forceinline__ void combo_finish(ComboFetch& cf, float out[9]) {
    ComboTap& tlo = cf.tlo;
    ComboTap& thi = cf.thi;
    const float f = cf.f;
    // fold the level blend (1-f, f) and the UNORM8 scale into the eight bilinear weights: 8 FMAs per
    // channel and nothing else (VALUE arithmetic: same quantity as (1-f)*tau_lo + f*tau_hi, other rounding)
    const float klo = (1.0f - f) * kUnorm8, khi = f * kUnorm8;
    tlo.w00 *= klo; tlo.w10 *= klo; tlo.w01 *= klo; tlo.w11 *= klo;
    thi.w00 *= khi; thi.w10 *= khi; thi.w01 *= khi; thi.w11 *= khi;
    float lo[9], hi[9];
    combo_filter(cf.a0, cf.a1, tlo, lo);
    combo_filter(cf.b0, cf.b1, thi, hi);
#pragma unroll
    for (int ch = 0; ch < 9; ch++) out[ch] = lo[ch] + hi[ch];
}
template <class MP, class TD, class TS>
__device__ __forceinline__ void combo_sample(MP mp, TD t, float uf, float vf, const TS& ts, float out[9]) {
    ComboFetch cf;
    combo_issue(mp, t, uf, vf, ts, cf);
    combo_finish(cf, out);
}


template <typename T>
__device__ __forceinline__ T ld_plane(const T* base, uint32_t t) {
    // 32-bit byte offset from a wave-uniform base: lets the backend use the saddr + voffset form
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + (size_t)(t * (uint32_t)sizeof(T)));
}

// The per-fragment part of rasteriser + FS (converterFS.glsl:44-104) for pixel (x,y) of triangle t.
// `mp` should be wave-uniform (scalar) for speed; correctness does not depend on it.
// `stamps` (debug timing builds only, else nullptr and folded away): three s_memtime slots.
// MP is `const MeshParams*` or the same pointer in the CONSTANT address space (kConstMesh below).  The mesh table is
// written at upload and never while a conversion runs; saying so lets the compiler use scalar loads (s_load) for the
// descriptor fields when the pointer is wave-uniform, even after the kernel's own record stores (which otherwise make
// every later global read a vector load: alias analysis cannot tell the records from the table).

// kComboOnly (k_fused3): the caller guarantees that the mesh samples through its combo texture or has no map at all; the
// separate-maps sampler is then not compiled into the caller (it is what sets the register count of the fragment stage).
template <class MP, class TS = TriShade, bool kComboOnly = false>   // TS: TriShade, or the 64-byte TriShadeS (same field names)
__device__ __forceinline__ void shade_from_tri(const TriPlanes& tp, uint32_t t, int x, int y, MP mp,
                                               const TS& ts, float4 rec[6], unsigned long long* stamps = nullptr,
                                               const float2* uvl = nullptr /* (u0,v0), (u1-u0,v1-v0), (u2-u0,v2-v0) kept by the caller */) {
    // screen-linear barycentrics from the exact integer edge functions, evaluated relative to the
    // triangle's bbox origin pixel: E_i(x,y) = E_i(x0,y0) + a_i*256*(x-x0) + b_i*256*(y-y0)
    const int dx256 = (x - (int)(ts.org & 0xFFFu)) * 256, dy256 = (y - (int)(ts.org >> 12)) * 256;
    float l1, l2;
    if constexpr (sizeof(ts.e1) == 4) {
        // TriShadeS: |a|, |b| <= 2304 and the box is at most 8 x 8 pixels, so every edge value inside it stays below 2^31 (at a
        // covered pixel: 0 <= E <= area2 <= 2304^2): 32-bit integers, and (float)int32 rounds once like i64_to_f32
        const int E1 = ts.e1 + (int)ts.a1 * dx256 + (int)ts.b1 * dy256;
        const int E2 = ts.e2 + (int)ts.a2 * dx256 + (int)ts.b2 * dy256;
        l1 = (float)E1 * ts.inva;
        l2 = (float)E2 * ts.inva;
    } else {
        const long long E1 = ts.e1 + (long long)ts.a1 * dx256 + (long long)ts.b1 * dy256;
        const long long E2 = ts.e2 + (long long)ts.a2 * dx256 + (long long)ts.b2 * dy256;
        l1 = i64_to_f32(E1) * ts.inva;
        l2 = i64_to_f32(E2) * ts.inva;
    }

    // smooth varyings (converterGS.glsl:432-441): Position, Normal, Tangent, UV.
    // The UV planes are requested FIRST: vector-memory results return in issue order, so the texture
    // coordinates (and with them the dependent texel fetches) need to wait only for these two loads while
    // the other nine attribute loads are still in flight.
    float4 b0 = make_float4(0, 0, 0, 0);
    float2 b1 = make_float2(0, 0);
    if (uvl == nullptr) {
        b0 = ld_plane(tp.B0, t);
        b1 = ld_plane(tp.B1, t);
    }
    const float4 a0 = ld_plane(tp.A0, t), a1 = ld_plane(tp.A1, t);
    const float a2 = ld_plane(tp.A2, t);
    const float4 c0 = ld_plane(tp.C0, t), c1 = ld_plane(tp.C1, t);
    const float c2 = ld_plane(tp.C2, t);
    const float4 d0 = ld_plane(tp.D0, t), d1 = ld_plane(tp.D1, t), d2 = ld_plane(tp.D2, t);
    float U, V;
    {   // texture coordinates: exact oracle sequence (no FMA), see tri_shade_setup
#pragma clang fp contract(off)
        if (uvl != nullptr) {   // the same three roundings per coordinate: the differences were formed by the same subtraction
            const float2 q0 = uvl[0], q1 = uvl[1], q2 = uvl[2];
            U = (q0.x + l1 * q1.x) + l2 * q2.x;
            V = (q0.y + l1 * q1.y) + l2 * q2.y;
        } else {
            U = (b0.x + l1 * (b0.z - b0.x)) + l2 * (b1.x - b0.x);
            V = (b0.y + l1 * (b0.w - b0.y)) + l2 * (b1.y - b0.y);
        }
    }
#define M2S_LERP(f0, f1, f2) fma_(l2, (f2) - (f0), fma_(l1, (f1) - (f0), (f0)))

    const auto ta = &mp->tex[0];
    const auto tn = &mp->tex[1];
    const auto tm = &mp->tex[2];
    const uint32_t* xa = ta->texels;   // (non-const only for the debug ablation switch below)
    const uint32_t* xn = tn->texels;
    const uint32_t* xm = tm->texels;
    if (stamps) stamps[0] = __builtin_amdgcn_s_memtime();   // uv planes arrived, U/V computed
    const float uf = frac_repeat(U), vf = frac_repeat(V);
    float col[4] = { 1.0f, 1.0f, 1.0f, 1.0f };   // FS:53-62
    float nrm[3] = { 0.0f, 0.0f, 1.0f };
    float metal = 0.1f, rough = 0.5f;            // FS:87-95 defaults
    const uint32_t* __restrict__ cmb = mp->combo.texels;
    if (cmb != nullptr) {
        // all three maps, same size: one LOD state, interleaved texels, 2 x 24-byte row reads per level
        float acc[9];
        combo_sample(mp, ta, uf, vf, ts, acc);
        col[0] = acc[0]; col[1] = acc[1]; col[2] = acc[2]; col[3] = acc[3];
        nrm[0] = acc[4]; nrm[1] = acc[5]; nrm[2] = acc[6];
        rough = acc[7]; metal = acc[8];
    } else if constexpr (!kComboOnly) {
        // separate maps (sizes differ, or a map is missing): the sampling state of every present map first, then ALL texel
        // reads (up to 24), then the filters — one memory round trip for the whole texture stage (round 3; was up to six)
        TexState sa, sn, sm;
        TexFetch fa, fn, fm;
        const bool n_like_a = xa != nullptr && xn != nullptr && tn->w == ta->w && tn->h == ta->h;
        const bool m_like_a = xa != nullptr && xm != nullptr && tm->w == ta->w && tm->h == ta->h;
        const bool m_like_n = xn != nullptr && xm != nullptr && tm->w == tn->w && tm->h == tn->h;
        if (xa != nullptr) tex_state(ta, uf, vf, ts.lod0, sa);
        if (xn != nullptr) { if (n_like_a) sn = sa; else tex_state(tn, uf, vf, ts.lod1, sn); }
        if (xm != nullptr) {
            // (the state of the normal map when it has the size of this map, else the albedo map's: as before)
            if (xn != nullptr ? m_like_n : m_like_a) sm = (xn != nullptr) ? sn : sa;
            else tex_state(tm, uf, vf, ts.lod2, sm);
        }
        if (xa != nullptr) tex_issue(xa, sa, fa);
        if (xn != nullptr) tex_issue(xn, sn, fn);
        if (xm != nullptr) tex_issue(xm, sm, fm);
        if (xa != nullptr) tex_finish<4>(fa, sa, col);
        if (xn != nullptr) tex_finish<3>(fn, sn, nrm);
        if (xm != nullptr) {
            float sv[3];
            tex_finish<3>(fm, sm, sv);
            metal = sv[2]; rough = sv[1];
        }
    }
    if (stamps) stamps[1] = __builtin_amdgcn_s_memtime();   // texels arrived and filtered
    // the remaining varyings: by now their loads have had the whole texture fetch to arrive
    const float Pxw = M2S_LERP(a0.x, a0.w, a1.z), Pyw = M2S_LERP(a0.y, a1.x, a1.w), Pzw = M2S_LERP(a0.z, a1.y, a2);
    const float Nx = M2S_LERP(c0.x, c0.w, c1.z), Ny = M2S_LERP(c0.y, c1.x, c1.w), Nz = M2S_LERP(c0.z, c1.y, c2);
    // FS:66-81
    float ox = Nx, oy = Ny, oz = Nz;
    if (xn != nullptr) {
        const float Tx = M2S_LERP(d0.x, d1.x, d2.x), Ty = M2S_LERP(d0.y, d1.y, d2.y), Tz = M2S_LERP(d0.z, d1.z, d2.z);
        const float Tw = M2S_LERP(d0.w, d1.w, d2.w);
        float rx = fma_(nrm[0], 2.0f, -1.0f), ry = fma_(nrm[1], 2.0f, -1.0f), rz = fma_(nrm[2], 2.0f, -1.0f);
        float inv = fast_rsq(dot3_(rx, ry, rz, rx, ry, rz));
        rx *= inv; ry *= inv; rz *= inv;
        // cross(Normal, Tangent.xyz)
        float bx = fma_(Ny, Tz, -(Nz * Ty)), by = fma_(Nz, Tx, -(Nx * Tz)), bz = fma_(Nx, Ty, -(Ny * Tx));
        inv = fast_rsq(dot3_(bx, by, bz, bx, by, bz)) * Tw;
        bx *= inv; by *= inv; bz *= inv;
        inv = fast_rsq(dot3_(Nx, Ny, Nz, Nx, Ny, Nz));
        const float nnx = Nx * inv, nny = Ny * inv, nnz = Nz * inv;
        const float wx = fma_(nnx, rz, fma_(bx, ry, Tx * rx)), wy = fma_(nny, rz, fma_(by, ry, Ty * rx)), wz = fma_(nnz, rz, fma_(bz, ry, Tz * rx));
        inv = fast_rsq(dot3_(wx, wy, wz, wx, wy, wz));
        ox = wx * inv; oy = wy * inv; oz = wz * inv;
    }
#undef M2S_LERP
    if (stamps) stamps[2] = __builtin_amdgcn_s_memtime();   // interpolation + TBN done
    // FS:98-103
    rec[0] = make_float4(Pxw, Pyw, Pzw, 1.0f);
    rec[1] = make_float4(col[0] * mp->color[0], col[1] * mp->color[1], col[2] * mp->color[2], col[3] * mp->color[3]);
    rec[2] = make_float4(ts.sx, ts.sy, 1e-7f, 0.0f);
    rec[3] = make_float4(ox, oy, oz, 0.0f);
    rec[4] = ts.rot;
    rec[5] = make_float4(metal, rough, 0.0f, 1.0f);
}

// Record stores are non-temporal: the records are never re-read by the conversion; keeping them out of the 4 MiB L2
// leaves it to the texture / vertex lines (measured on the first single-pass kernel, round 1: 0.236 -> 0.200 ms).
__device__ __forceinline__ void nt_store(float4* p, float4 v) {
    __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y);
    __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);   // merged into one dwordx4 nt
}

}  // namespace m2s
