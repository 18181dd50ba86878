// m2s_kernels.hip — hand-written HIP kernels (gfx950 / CDNA4, wave64) for the mesh -> 3DGS
// conversion pass.  Replaces the reference's VS+GS+rasteriser+FS program
//   src/shaders/conversion/converterVS.glsl, converterGS.glsl:326-443, converterFS.glsl:44-104
// driven by ConversionPass::execute (src/renderer/renderPasses/ConversionPass.cpp:9-117).
//
// Pipeline (one stream, no host round trip until the final counter read-back):
//   k_count    1 thread / triangle : GS setup + exact fragment count (closed-form row spans)
//   k_scan     1 workgroup         : exclusive scan of the per-1024-triangle partial sums
//   k_offsets  1 thread / triangle : per-triangle output offsets + emit-block start table
//   k_emit     1 thread / Gaussian : load-balanced expansion (triangle -> pixels) through LDS, full
//                                    GS+FS math per fragment, 96 B records staged in LDS and
//                                    written with fully coalesced 16 B/lane stores.
// Output order is deterministic: (mesh, triangle, pixel row, pixel column) — the reference's order
// is atomic-arrival order (converterFS.glsl:46), i.e. unspecified.
//
// Arithmetic is fp32 with one rounding per operation (compiled with -ffp-contract=off; HIP's
// default correctly-rounded fp32 divide/sqrt), integer rasterisation on a 24.8 grid with int64
// edge functions: see DESIGN.md "Pinned semantics".  No MFMA: nothing here is a contraction.
#include "m2s_device.h"

#pragma clang fp contract(off)

namespace m2s {

// ============================================================================================
// small helpers
// ============================================================================================
__device__ __forceinline__ float len3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t n = __shfl_up(v, d);
        if (lane >= d) v += n;
    }
    return v;
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ void wave_lds_sync() {
    // LDS operations of one wave execute in order; this only stops the compiler from moving
    // LDS accesses across the point and drains outstanding LDS traffic.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// exact floor(num/den), den > 0, |num| < 2^52: fp64 quotient + one integer correction step
__device__ __forceinline__ long long floordiv_pos(long long num, long long den) {
    long long q = (long long)floor((double)num / (double)den);
    long long r = num - q * den;
    if (r < 0) q -= 1;
    else if (r >= den) q += 1;
    return q;
}

// ============================================================================================
// geometry-shader restatement (converterGS.glsl:326-443)
// ============================================================================================
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
    float l1 = len3(e1x, e1y, e1z), l2 = len3(e2x, e2y, e2z), l3 = len3(e3x, e3y, e3z);
    // GS:333-342: strict >, else-if; second branch leaves edge2 untouched
    if (l2 > l1 && l2 > l3) {
        float tx = e1x, ty = e1y, tz = e1z;
        e1x = e2x; e1y = e2y; e1z = e2z;
        e2x = tx; e2y = ty; e2z = tz;
    } else if (l3 > l1 && l3 > l2) {
        e1x = e3x; e1y = e3y; e1z = e3z;
    }
    float inv = 1.0f / len3(e1x, e1y, e1z);
    g.xx = e1x * inv; g.xy = e1y * inv; g.xz = e1z * inv;
    float cx = g.xy * e2z - g.xz * e2y, cy = g.xz * e2x - g.xx * e2z, cz = g.xx * e2y - g.xy * e2x;
    inv = 1.0f / len3(cx, cy, cz);
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
    float invRange = 1.0f / range;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float pa = useY_asA ? p[3 * i + 1] : p[3 * i + 0];
        float pb = useY_asB ? p[3 * i + 1] : p[3 * i + 2];
        g.ou[i] = (pa - bminA) * invRange;
        g.ov[i] = (pb - bminB) * invRange;
    }
}

// flat outputs of the GS: Scale (GS:409-430) and Quaternion (GS:401-407, quat_cast GS:131-183)
__device__ __forceinline__ void geo_flat(const float p[9], const Geo& g, float& sx, float& sy, float4& rot) {
    // yAxis = normalize(cross(normal, xAxis))
    float cx = g.ny * g.xz - g.nz * g.xy, cy = g.nz * g.xx - g.nx * g.xz, cz = g.nx * g.xy - g.ny * g.xx;
    float inv = 1.0f / len3(cx, cy, cz);
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
    float bv = sqrtf(fb + 1.0f) * 0.5f;
    float mult = 0.25f / bv;
    float qx, qy, qz, qw;
    if (bi == 0) {
        qw = bv; qx = (m12 - m21) * mult; qy = (m20 - m02) * mult; qz = (m01 - m10) * mult;
    } else if (bi == 1) {
        qw = (m12 - m21) * mult; qx = bv; qy = (m01 + m10) * mult; qz = (m20 + m02) * mult;
    } else if (bi == 2) {
        qw = (m20 - m02) * mult; qx = (m01 + m10) * mult; qy = bv; qz = (m12 + m21) * mult;
    } else {
        qw = (m01 - m10) * mult; qx = (m20 + m02) * mult; qy = (m12 + m21) * mult; qz = bv;
    }
    rot = make_float4(qw, qx, qy, qz);  // GS:407 stores (w,x,y,z)
    // Jacobian: UVMatrix[col][row], inverse2x2 (GS:206-220), multiplyMat2x3WithMat2x2 (GS:222-235)
    float U00 = g.ou[1] - g.ou[0], U10 = g.ou[2] - g.ou[0];
    float U01 = g.ov[1] - g.ov[0], U11 = g.ov[2] - g.ov[0];
    float det = U00 * U11 - U01 * U10;
    float I00 = 0.0f, I10 = 0.0f, I01 = 0.0f, I11 = 0.0f;
    if (det != 0.0f) {
        float invDet = 1.0f / det;
        I00 = U11 * invDet;
        I10 = -U10 * invDet;
        I01 = -U01 * invDet;
        I11 = U00 * invDet;
    }
    float v0x = p[3] - p[0], v0y = p[4] - p[1], v0z = p[5] - p[2];
    float v1x = p[6] - p[0], v1y = p[7] - p[1], v1z = p[8] - p[2];
    sx = len3(v0x * I00 + v1x * I01, v0y * I00 + v1y * I01, v0z * I00 + v1z * I01);
    sy = len3(v0x * I10 + v1x * I11, v0y * I10 + v1y * I11, v0z * I10 + v1z * I11);
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
};

constexpr float kGuardPx = 16384.0f;

__device__ __forceinline__ bool raster_setup(const Geo& g, uint32_t R, Raster& s) {
    const float half = (float)R * 0.5f;
    int X[3], Y[3];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float ndx = g.ou[i] * 2.0f - 1.0f, ndy = g.ov[i] * 2.0f - 1.0f;  // GS:439
        float xw = half * ndx + half, yw = half * ndy + half;            // glViewport(0,0,R,R)
        ok = ok && (fabsf(xw) < kGuardPx) && (fabsf(yw) < kGuardPx);     // false for NaN
        X[i] = (int)rintf(xw * 256.0f);
        Y[i] = (int)rintf(yw * 256.0f);
    }
    if (!ok) return false;
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
    int xmin = min(X[0], min(X[1], X[2])), xmax = max(X[0], max(X[1], X[2]));
    int ymin = min(Y[0], min(Y[1], Y[2])), ymax = max(Y[0], max(Y[1], Y[2]));
    s.x0 = max((xmin - 128 + 255) >> 8, 0);
    s.x1 = min((xmax - 128) >> 8, (int)R - 1);
    s.y0 = max((ymin - 128 + 255) >> 8, 0);
    s.y1 = min((ymax - 128) >> 8, (int)R - 1);
    return s.x0 <= s.x1 && s.y0 <= s.y1;
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

void launch_mip_level(const uint32_t* src, uint32_t sw, uint32_t sh, uint32_t* dst, uint32_t dw, uint32_t dh,
                      hipStream_t st) {
    uint32_t n = dw * dh;
    hipLaunchKernelGGL(k_mip, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, src, sw, sh, dst, dw, dh);
}

// ============================================================================================
// K1: per-triangle fragment count
// ============================================================================================
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

__global__ void __launch_bounds__(kBlock) k_count(SceneDev sc, uint32_t R, uint32_t* __restrict__ cnt,
                                                  uint32_t* __restrict__ partials) {
    __shared__ uint32_t red[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = blockIdx.x * kTriPerBlock;
    const uint32_t lastT = min(blockBase + kTriPerBlock, sc.n_tri) - 1;
    const uint32_t m0 = find_mesh(sc, sc.tri_first + blockBase);
    const bool uniform_mesh = (m0 + 1 >= sc.n_meshes) || (sc.mesh_first[m0 + 1] > sc.tri_first + lastT);
    uint32_t sum = 0;
    for (int it = 0; it < kTriPerBlock / kBlock; ++it) {
        const uint32_t t = blockBase + it * kBlock + threadIdx.x;
        const bool valid = t < sc.n_tri;
        Raster rs;
        bool ok = false;
        if (valid) ok = setup_raster_for(sc, t, m0, uniform_mesh, R, rs);
        const int rows = ok ? rs.y1 - rs.y0 + 1 : 0;
        uint32_t c = 0;
        if (ok && rows <= kRowsThread) {
            for (int y = rs.y0; y <= rs.y1; ++y) {
                int xa, xb;
                row_span(rs, y, xa, xb);
                c += (uint32_t)max(xb - xa + 1, 0);
            }
        }
        // triangles spanning many rows: the whole wave counts one triangle, one row per lane
        unsigned long long big = __ballot(ok && rows > kRowsThread);
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

// ============================================================================================
// K_offsets: off[t] = exclusive prefix of cnt; start[m] = triangle that owns output index m*kEmitF
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_offsets(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ partials,
                                                    uint32_t n_tri, uint32_t* __restrict__ off, uint32_t* __restrict__ start,
                                                    uint32_t n_start) {
    __shared__ uint32_t wsum[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t blockBase = blockIdx.x * kTriPerBlock;
    uint32_t run = partials[blockIdx.x];
    for (int it = 0; it < kTriPerBlock / kBlock; ++it) {
        const uint32_t t = blockBase + it * kBlock + threadIdx.x;
        const uint32_t c = t < n_tri ? cnt[t] : 0;
        uint32_t incl = wave_incl_scan(c, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) {
            if (w < wave) woff += wsum[w];
            tot += wsum[w];
        }
        const uint32_t o0 = run + woff + incl - c;
        if (t < n_tri) {
            off[t] = o0;
            if (t == n_tri - 1) off[n_tri] = o0 + c;
            if (c) {
                const uint32_t mf = (o0 + kEmitF - 1) / kEmitF, ml = (o0 + c - 1) / kEmitF;
                for (uint32_t m = mf; m <= ml && m < n_start; ++m) start[m] = t;
            }
        }
        run += tot;
        __syncthreads();
    }
}

void launch_offsets(const uint32_t* cnt, const uint32_t* partials, uint32_t n_tri, uint32_t* off, uint32_t* start,
                    uint32_t n_start, hipStream_t st) {
    if (!n_tri) return;
    hipLaunchKernelGGL(k_offsets, dim3(n_count_blocks(n_tri)), dim3(kBlock), 0, st, cnt, partials, n_tri, off, start,
                       n_start);
}

// ============================================================================================
// fragment-shader restatement (converterFS.glsl:44-104) with software trilinear sampling
// (sampler state glUtils.cpp:292-313: REPEAT, LINEAR_MIPMAP_LINEAR / LINEAR, levels 0..4)
// ============================================================================================
constexpr float kUnorm8 = 0.003921568859368563f;  // fp32 nearest to 1/255

__device__ __forceinline__ float frac_repeat(float u) {
    float f = u - floorf(u);
    if (!(f >= 0.0f)) f = 0.0f;
    if (f > 1.0f) f = 1.0f;
    return f;
}

// un-normalised bilinear sum of raw byte values on one level
__device__ __forceinline__ void bilinear(const TexDesc* __restrict__ t, uint32_t level, float uf, float vf, float out[4]) {
    const uint32_t W = max(1u, t->w >> level), H = max(1u, t->h >> level);
    const uint32_t* __restrict__ img = t->texels + t->off[level];
    const float up = uf * (float)W - 0.5f, vp = vf * (float)H - 0.5f;
    const float fi = floorf(up), fj = floorf(vp);
    const float a = up - fi, b = vp - fj;
    int i0 = (int)fi, j0 = (int)fj;
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += (int)W;
    if (j0 < 0) j0 += (int)H;
    if (i0 >= (int)W) i0 -= (int)W;
    if (j0 >= (int)H) j0 -= (int)H;
    if (i1 >= (int)W) i1 -= (int)W;
    if (j1 >= (int)H) j1 -= (int)H;
    if (i1 >= (int)W) i1 -= (int)W;
    if (j1 >= (int)H) j1 -= (int)H;
    const uint32_t t00 = img[(size_t)j0 * W + i0], t10 = img[(size_t)j0 * W + i1];
    const uint32_t t01 = img[(size_t)j1 * W + i0], t11 = img[(size_t)j1 * W + i1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        const float c00 = (float)((t00 >> (8 * ch)) & 255u), c10 = (float)((t10 >> (8 * ch)) & 255u);
        const float c01 = (float)((t01 >> (8 * ch)) & 255u), c11 = (float)((t11 >> (8 * ch)) & 255u);
        out[ch] = ((w00 * c00 + w10 * c10) + w01 * c01) + w11 * c11;
    }
}

__device__ __forceinline__ float lod_lambda(const TexDesc* __restrict__ t, float dudx, float dvdx, float dudy, float dvdy) {
    const float fw = (float)t->w, fh = (float)t->h;
    const float sx = dudx * fw, tx = dvdx * fh, sy = dudy * fw, ty = dvdy * fh;
    const float rx = sqrtf(sx * sx + tx * tx), ry = sqrtf(sy * sy + ty * ty);
    return log2f(fmaxf(rx, ry));
}

__device__ __forceinline__ void sample_lod(const TexDesc* __restrict__ t, float u, float v, float lambda, float out[4]) {
    const float uf = frac_repeat(u), vf = frac_repeat(v);
    const uint32_t nl = t->n_levels;
    float t1[4];
    if (!(lambda > 0.0f)) {
        bilinear(t, 0, uf, vf, t1);
#pragma unroll
        for (int ch = 0; ch < 4; ch++) out[ch] = t1[ch] * kUnorm8;
        return;
    }
    if (lambda >= (float)(nl - 1)) {
        bilinear(t, nl - 1, uf, vf, t1);
#pragma unroll
        for (int ch = 0; ch < 4; ch++) out[ch] = t1[ch] * kUnorm8;
        return;
    }
    const float d = floorf(lambda), f = lambda - d;
    float t2[4];
    bilinear(t, (uint32_t)d, uf, vf, t1);
    bilinear(t, (uint32_t)d + 1, uf, vf, t2);
#pragma unroll
    for (int ch = 0; ch < 4; ch++) out[ch] = ((1.0f - f) * t1[ch] + f * t2[ch]) * kUnorm8;
}

// Everything the GS + rasteriser + FS produce for ONE fragment (triangle t, pixel x,y).
__device__ __forceinline__ void shade_fragment(const SceneDev& sc, uint32_t t, int x, int y, uint32_t mesh_hint,
                                               bool uniform_mesh, uint32_t R, float4 rec[6]) {
    const TriPlanes& tp = sc.tri;
    float p[9];
    load_positions(tp, t, p);
    const uint32_t m = uniform_mesh ? mesh_hint : find_mesh(sc, sc.tri_first + t);
    const MeshParams* __restrict__ mp = sc.meshes + m;
    Geo g;
    geo_setup(p, mp->bmin, mp->bmax, g);
    Raster rs;
    raster_setup(g, R, rs);
    float sx, sy;
    float4 rot;
    geo_flat(p, g, sx, sy, rot);

    // screen-linear barycentrics from the exact integer edge functions
    const long long Px = 256ll * x + 128, Py = 256ll * y + 128;
    const long long E1 = (long long)rs.a[1] * Px + (long long)rs.b[1] * Py + rs.c[1];
    const long long E2 = (long long)rs.a[2] * Px + (long long)rs.b[2] * Py + rs.c[2];
    const float inva = 1.0f / (float)rs.area2;
    const float l1 = (float)E1 * inva, l2 = (float)E2 * inva;

    // smooth varyings (converterGS.glsl:432-441): Position, Normal, Tangent, UV
    const float4 b0 = tp.B0[t];
    const float2 b1 = tp.B1[t];
    const float4 c0 = tp.C0[t], c1 = tp.C1[t];
    const float c2 = tp.C2[t];
    const float4 d0 = tp.D0[t], d1 = tp.D1[t], d2 = tp.D2[t];
#define M2S_LERP(f0, f1, f2) (((f0) + l1 * ((f1) - (f0))) + l2 * ((f2) - (f0)))
    const float Pxw = M2S_LERP(p[0], p[3], p[6]), Pyw = M2S_LERP(p[1], p[4], p[7]), Pzw = M2S_LERP(p[2], p[5], p[8]);
    const float Nx = M2S_LERP(c0.x, c0.w, c1.z), Ny = M2S_LERP(c0.y, c1.x, c1.w), Nz = M2S_LERP(c0.z, c1.y, c2);
    const float Tx = M2S_LERP(d0.x, d1.x, d2.x), Ty = M2S_LERP(d0.y, d1.y, d2.y), Tz = M2S_LERP(d0.z, d1.z, d2.z);
    const float Tw = M2S_LERP(d0.w, d1.w, d2.w);
    const float U = M2S_LERP(b0.x, b0.z, b1.x), V = M2S_LERP(b0.y, b0.w, b1.y);
#undef M2S_LERP

    const TexDesc* __restrict__ ta = &mp->tex[0];
    const TexDesc* __restrict__ tn = &mp->tex[1];
    const TexDesc* __restrict__ tm = &mp->tex[2];
    const bool hasA = ta->texels != nullptr, hasN = tn->texels != nullptr, hasM = tm->texels != nullptr;
    float dudx = 0, dvdx = 0, dudy = 0, dvdy = 0;
    if (hasA || hasN || hasM) {
        // UV is affine in window space (all w = 1, GS:439), so the derivatives are per-triangle constants
        const float g1x = (float)((long long)rs.a[1] * 256) * inva, g2x = (float)((long long)rs.a[2] * 256) * inva;
        const float g1y = (float)((long long)rs.b[1] * 256) * inva, g2y = (float)((long long)rs.b[2] * 256) * inva;
        const float du1 = b0.z - b0.x, du2 = b1.x - b0.x, dv1 = b0.w - b0.y, dv2 = b1.y - b0.y;
        dudx = g1x * du1 + g2x * du2; dvdx = g1x * dv1 + g2x * dv2;
        dudy = g1y * du1 + g2y * du2; dvdy = g1y * dv1 + g2y * dv2;
    }
    // FS:53-62
    float col[4] = { 1.0f, 1.0f, 1.0f, 1.0f };
    if (hasA) sample_lod(ta, U, V, lod_lambda(ta, dudx, dvdx, dudy, dvdy), col);
    // FS:66-81
    float ox = Nx, oy = Ny, oz = Nz;
    if (hasN) {
        float s[4];
        sample_lod(tn, U, V, lod_lambda(tn, dudx, dvdx, dudy, dvdy), s);
        float rx = s[0] * 2.0f - 1.0f, ry = s[1] * 2.0f - 1.0f, rz = s[2] * 2.0f - 1.0f;
        float inv = 1.0f / len3(rx, ry, rz);
        rx *= inv; ry *= inv; rz *= inv;
        float bx = Ny * Tz - Nz * Ty, by = Nz * Tx - Nx * Tz, bz = Nx * Ty - Ny * Tx;  // cross(Normal, Tangent.xyz)
        inv = 1.0f / len3(bx, by, bz);
        bx = (bx * inv) * Tw; by = (by * inv) * Tw; bz = (bz * inv) * Tw;
        inv = 1.0f / len3(Nx, Ny, Nz);
        const float nnx = Nx * inv, nny = Ny * inv, nnz = Nz * inv;
        float wx = (Tx * rx + bx * ry) + nnx * rz, wy = (Ty * rx + by * ry) + nny * rz, wz = (Tz * rx + bz * ry) + nnz * rz;
        inv = 1.0f / len3(wx, wy, wz);
        ox = wx * inv; oy = wy * inv; oz = wz * inv;
    }
    // FS:87-95
    float metal = 0.1f, rough = 0.5f;
    if (hasM) {
        float s[4];
        sample_lod(tm, U, V, lod_lambda(tm, dudx, dvdx, dudy, dvdy), s);
        metal = s[2]; rough = s[1];
    }
    // FS:98-103
    rec[0] = make_float4(Pxw, Pyw, Pzw, 1.0f);
    rec[1] = make_float4(col[0] * mp->color[0], col[1] * mp->color[1], col[2] * mp->color[2], col[3] * mp->color[3]);
    rec[2] = make_float4(sx, sy, 1e-7f, 0.0f);
    rec[3] = make_float4(ox, oy, oz, 0.0f);
    rec[4] = rot;
    rec[5] = make_float4(metal, rough, 0.0f, 1.0f);
}

// ============================================================================================
// K2: emit.  Workgroup b owns output records [b*kEmitF, (b+1)*kEmitF).
// ============================================================================================
__global__ void __launch_bounds__(kBlock) k_emit(SceneDev sc, uint32_t R, const uint32_t* __restrict__ off,
                                                 const uint32_t* __restrict__ start,
                                                 const unsigned long long* __restrict__ total_p, unsigned long long limit,
                                                 float4* __restrict__ out) {
    __shared__ uint2 entries[kEmitF];                 // (local triangle, y<<16 | x)
    __shared__ float4 stage[kBlock * kStageStride];   // 28 KiB
    __shared__ uint32_t wrow_off[kBlock / 64][64];
    __shared__ int wrow_xa[kBlock / 64][64];

    const unsigned long long total = *total_p;
    const unsigned long long nw = total < limit ? total : limit;  // records actually stored
    // XCD-aware mapping: hardware places workgroup b on XCD b % 8 and each XCD has a private 4 MiB L2.
    // Give every XCD one CONTIGUOUS slice of the output (hence of the mesh surface and of texture
    // space) instead of every 8th block, so texture / vertex lines are fetched by one L2, not eight.
    const uint32_t nblk = (uint32_t)((nw + kEmitF - 1) / kEmitF);
    const uint32_t per_xcd = (nblk + 7) / 8;
    const uint32_t lblock = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || lblock >= nblk) return;
    const unsigned long long base64 = (unsigned long long)lblock * kEmitF;
    const uint32_t base = (uint32_t)base64;
    const uint32_t end = (uint32_t)(nw - base64 < (unsigned long long)kEmitF ? nw : base64 + kEmitF);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t T = sc.n_tri;

    // ---- phase A: expand triangles into (triangle, pixel) entries, in canonical order ----
    const uint32_t t_first = start[lblock];
    const uint32_t m0 = find_mesh(sc, sc.tri_first + t_first);
    uint32_t t_last_seen = t_first;
    for (uint32_t tc = t_first;; tc += kBlock) {
        const uint32_t t = tc + threadIdx.x;
        uint32_t o0 = 0, o1 = 0;
        if (t < T) { o0 = off[t]; o1 = off[t + 1]; }
        const bool active = (o1 > o0) && (o0 < end) && (o1 > base);
        Raster rs;
        bool ok = false;
        if (active) ok = setup_raster_for(sc, t, 0, false, R, rs);
        const int rows = ok ? rs.y1 - rs.y0 + 1 : 0;
        if (ok && rows <= kRowsThread) {
            uint32_t k = o0;
            for (int y = rs.y0; y <= rs.y1 && k < end; ++y) {
                int xa, xb;
                row_span(rs, y, xa, xb);
                for (int x = xa; x <= xb; ++x, ++k)
                    if (k >= base && k < end) entries[k - base] = make_uint2(t, ((uint32_t)y << 16) | (uint32_t)x);
            }
        }
        unsigned long long big = __ballot(ok && rows > kRowsThread);
        while (big) {
            const int src = __ffsll((long long)big) - 1;
            big &= big - 1;
            const Raster b = shfl_raster(rs, src);
            const uint32_t bt = __shfl(t, src);
            uint32_t acc = __shfl(o0, src);
            for (int yc = b.y0; yc <= b.y1 && acc < end; yc += 64) {
                const int y = yc + lane;
                int xa = 0, xb = -1;
                if (y <= b.y1) row_span(b, y, xa, xb);
                const uint32_t len = (uint32_t)max(xb - xa + 1, 0);
                const uint32_t incl = wave_incl_scan(len, lane);
                const uint32_t chunk = __shfl(incl, 63);
                if (acc + chunk > base) {
                    wave_lds_sync();
                    wrow_off[wave][lane] = incl - len;
                    wrow_xa[wave][lane] = xa;
                    wave_lds_sync();
                    const uint32_t lo = acc < base ? base - acc : 0;
                    const uint32_t hi = acc + chunk > end ? end - acc : chunk;
                    for (uint32_t k = lo + lane; k < hi; k += 64) {
                        int r = 0;  // largest r with wrow_off[r] <= k
#pragma unroll
                        for (int step = 32; step >= 1; step >>= 1)
                            if (wrow_off[wave][r + step] <= k) r += step;
                        const int x = wrow_xa[wave][r] + (int)(k - wrow_off[wave][r]);
                        entries[acc + k - base] = make_uint2(bt, ((uint32_t)(yc + r) << 16) | (uint32_t)x);
                    }
                }
                acc += chunk;
            }
        }
        t_last_seen = min(tc + kBlock - 1, T - 1);
        if (tc + kBlock >= T) break;
        if (off[tc + kBlock] >= end) break;
    }
    const uint32_t m1 = find_mesh(sc, sc.tri_first + t_last_seen);
    const bool uniform_mesh = (m0 == m1);
    __syncthreads();

    // ---- phase B: one thread per Gaussian; records staged in LDS, then 16 B/lane coalesced stores ----
    const uint32_t n_here = end - base;
    for (uint32_t e0 = 0; e0 < n_here; e0 += kBlock) {
        const uint32_t e = e0 + threadIdx.x;
        if (e < n_here) {
            const uint2 en = entries[e];
            float4 rec[6];
            shade_fragment(sc, en.x, (int)(en.y & 0xFFFFu), (int)(en.y >> 16), m0, uniform_mesh, R, rec);
#pragma unroll
            for (int k = 0; k < 6; k++) stage[threadIdx.x * kStageStride + k] = rec[k];
        }
        __syncthreads();
        const uint32_t nrec = min((uint32_t)kBlock, n_here - e0);
        float4* __restrict__ dst = out + ((size_t)base + e0) * 6;
        for (uint32_t q = threadIdx.x; q < nrec * 6; q += kBlock) {
            const uint32_t r = q / 6, k = q - r * 6;
            dst[q] = stage[r * kStageStride + k];
        }
        __syncthreads();
    }
}

void launch_emit(const SceneDev& sc, uint32_t R, const uint32_t* off, const uint32_t* start,
                 const unsigned long long* total, uint64_t limit, float4* out, uint32_t n_blocks, hipStream_t st) {
    if (!n_blocks || !sc.n_tri) return;
    n_blocks = (n_blocks + 7u) & ~7u;  // the XCD swizzle needs whole groups of 8
    hipLaunchKernelGGL(k_emit, dim3(n_blocks), dim3(kBlock), 0, st, sc, R, off, start, total,
                       (unsigned long long)limit, out);
}

}  // namespace m2s
