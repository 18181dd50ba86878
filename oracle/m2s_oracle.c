/*
 * m2s_oracle.c — CPU ORACLE (test infrastructure only; see m2s_oracle.h).
 *
 * Plain C restatement of the reference conversion pass.  Every function cites the
 * reference file:line it follows (paths relative to the reference repo root).
 *
 * Build:  gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * (-ffp-contract=off is REQUIRED: the pinned semantics are "fp32, one rounding per
 *  operation, no fused multiply-add", so that every discrete decision — longest edge,
 *  triplanar axis, sub-pixel snapping — is reproducible bit-for-bit on the GPU.)
 *
 * PARITY: the reference ships no golden vectors or tests for this path, and its OpenGL implementation cannot
 * run here (no GL/EGL/OSMesa).  What pins this file instead is the reference ITSELF, run on this machine:
 *   - everything the shaders compute (converterVS/GS/FS.glsl: edge swap, triplanar axis, bbox-normalised UVs,
 *     Jacobian scale, quat_cast, TBN normal, colour, metallic/roughness, record layout) is checked BIT FOR BIT
 *     against the reference's own GLSL source, executed as C++ through its vendored glm
 *     (oracle/ref_glsl_check.cpp, tests/test_ref_glsl.py, fixtures tests/golden/ref_host/glsl_*);
 *   - the PLY writers are checked byte for byte against the reference's parsers.cpp (oracle/ref_host_check.cpp).
 * STILL UNPINNED (fixed-function GL, no reference code exists for it): which pixel centres a triangle covers
 * (top-left rule on a 1/256 px grid), varying interpolation, mip generation, LOD selection and trilinear
 * filtering.  Those follow the OpenGL 4.6 specification as cited at each function and are frozen by our own
 * known-answer tests (tests/test_oracle_kat.py) and golden fixtures.
 */
#include "m2s_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* small fp32 vector helpers; evaluation order is part of the pinned semantics           */
/* ------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 v3sub(v3 a, v3 b) { v3 r = { a.x - b.x, a.y - b.y, a.z - b.z }; return r; }
static inline float v3len(v3 a) { return sqrtf((a.x * a.x + a.y * a.y) + a.z * a.z); }
/* normalize(v) is pinned as v * (1/|v|): one IEEE reciprocal, three multiplies */
static inline v3 v3normalize(v3 a) { float inv = 1.0f / v3len(a); v3 r = { a.x * inv, a.y * inv, a.z * inv }; return r; }
static inline v3 v3cross(v3 a, v3 b) {
    v3 r = { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x };
    return r;
}

/* converterGS.glsl:131-183 (itself a GLSL translation of glm::quat_cast).
 * m[c][r] is column c, row r.  Returns (x,y,z,w) like the GLSL vec4 q. */
static void quat_cast(const float m[3][3], float q[4]) {
    float fourXSquaredMinus1 = m[0][0] - m[1][1] - m[2][2];
    float fourYSquaredMinus1 = m[1][1] - m[0][0] - m[2][2];
    float fourZSquaredMinus1 = m[2][2] - m[0][0] - m[1][1];
    float fourWSquaredMinus1 = m[0][0] + m[1][1] + m[2][2];
    int biggestIndex = 0;
    float fourBiggestSquaredMinus1 = fourWSquaredMinus1;
    if (fourXSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourXSquaredMinus1; biggestIndex = 1; }
    if (fourYSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourYSquaredMinus1; biggestIndex = 2; }
    if (fourZSquaredMinus1 > fourBiggestSquaredMinus1) { fourBiggestSquaredMinus1 = fourZSquaredMinus1; biggestIndex = 3; }
    float biggestVal = sqrtf(fourBiggestSquaredMinus1 + 1.0f) * 0.5f;
    float mult = 0.25f / biggestVal;
    float x, y, z, w;
    if (biggestIndex == 0) {
        w = biggestVal;
        x = (m[1][2] - m[2][1]) * mult;
        y = (m[2][0] - m[0][2]) * mult;
        z = (m[0][1] - m[1][0]) * mult;
    } else if (biggestIndex == 1) {
        w = (m[1][2] - m[2][1]) * mult;
        x = biggestVal;
        y = (m[0][1] + m[1][0]) * mult;
        z = (m[2][0] + m[0][2]) * mult;
    } else if (biggestIndex == 2) {
        w = (m[2][0] - m[0][2]) * mult;
        x = (m[0][1] + m[1][0]) * mult;
        y = biggestVal;
        z = (m[1][2] + m[2][1]) * mult;
    } else {
        w = (m[0][1] - m[1][0]) * mult;
        x = (m[2][0] + m[0][2]) * mult;
        y = (m[1][2] + m[2][1]) * mult;
        z = biggestVal;
    }
    q[0] = x; q[1] = y; q[2] = z; q[3] = w;
}

/* exported for the glm cross-check harness (oracle/ref_glm_check.cpp) */
void orc_quat_cast(const float m9_colmajor[9], float q_xyzw[4]) {
    float m[3][3];
    for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m[c][r] = m9_colmajor[c * 3 + r];
    quat_cast(m, q_xyzw);
}

/* ------------------------------------------------------------------------------------ */
/* textures: GenerateMipmap + LINEAR_MIPMAP_LINEAR / REPEAT  (glUtils.cpp:292-313)        */
/* ------------------------------------------------------------------------------------ */
static inline uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

uint32_t orc_mip_levels(uint32_t w, uint32_t h) {
    /* GL_TEXTURE_BASE_LEVEL 0, GL_TEXTURE_MAX_LEVEL 4 (glUtils.cpp:312-313); a complete chain of
     * a w x h image has floor(log2(max(w,h)))+1 levels, so the usable count is min of both. */
    uint32_t m = umax(w, h), n = 1;
    while (m > 1 && n < 5) { m >>= 1; n++; }
    return n;
}

uint64_t orc_mip_total_texels(uint32_t w, uint32_t h) {
    uint32_t n = orc_mip_levels(w, h);
    uint64_t tot = 0;
    for (uint32_t l = 0; l < n; l++) { tot += (uint64_t)umax(1, w >> l) * umax(1, h >> l); }
    return tot;
}

uint32_t orc_build_mips(const uint8_t* rgba8, uint32_t w, uint32_t h, uint8_t* dst,
                        uint64_t* level_offsets) {
    uint32_t n = orc_mip_levels(w, h);
    memcpy(dst, rgba8, (size_t)w * h * 4);
    uint64_t off = 0;
    level_offsets[0] = 0;
    for (uint32_t l = 1; l < n; l++) {
        uint32_t sw = umax(1, w >> (l - 1)), sh = umax(1, h >> (l - 1));
        uint32_t dw = umax(1, w >> l), dh = umax(1, h >> l);
        const uint8_t* src = dst + off * 4;
        off += (uint64_t)sw * sh;
        level_offsets[l] = off;
        uint8_t* d = dst + off * 4;
        for (uint32_t y = 0; y < dh; y++) {
            uint32_t y0 = 2 * y < sh ? 2 * y : sh - 1, y1 = 2 * y + 1 < sh ? 2 * y + 1 : sh - 1;
            for (uint32_t x = 0; x < dw; x++) {
                uint32_t x0 = 2 * x < sw ? 2 * x : sw - 1, x1 = 2 * x + 1 < sw ? 2 * x + 1 : sw - 1;
                for (int ch = 0; ch < 4; ch++) {
                    uint32_t s = src[((size_t)y0 * sw + x0) * 4 + ch] + src[((size_t)y0 * sw + x1) * 4 + ch] +
                                 src[((size_t)y1 * sw + x0) * 4 + ch] + src[((size_t)y1 * sw + x1) * 4 + ch];
                    d[((size_t)y * dw + x) * 4 + ch] = (uint8_t)((s + 2) >> 2); /* round half up */
                }
            }
        }
    }
    return n;
}

typedef struct {
    const uint8_t* chain; /* NULL = absent */
    uint8_t* owned;
    uint32_t w, h, n_levels;
    uint64_t off[5];
} tex_t;

static inline float frac_repeat(float u) {
    float f = u - floorf(u); /* REPEAT: integer part of the coordinate is ignored */
    if (!(f >= 0.0f)) f = 0.0f; /* NaN / -inf guard (never hit by finite inputs) */
    if (f > 1.0f) f = 1.0f;
    return f;
}

static void bilinear(const tex_t* t, uint32_t level, float uf, float vf, float out[4]) {
    uint32_t W = umax(1, t->w >> level), H = umax(1, t->h >> level);
    const uint8_t* img = t->chain + t->off[level] * 4;
    float up = uf * (float)W - 0.5f, vp = vf * (float)H - 0.5f;
    float fi = floorf(up), fj = floorf(vp);
    float a = up - fi, b = vp - fj;
    int i0 = (int)fi, j0 = (int)fj; /* in [-1, W-1] */
    int i1 = i0 + 1, j1 = j0 + 1;
    if (i0 < 0) i0 += (int)W;
    if (j0 < 0) j0 += (int)H;
    if (i1 >= (int)W) i1 -= (int)W;
    if (j1 >= (int)H) j1 -= (int)H;
    if (i0 >= (int)W) i0 -= (int)W; /* uf == 1.0 edge */
    if (j0 >= (int)H) j0 -= (int)H;
    if (i1 >= (int)W) i1 -= (int)W;
    if (j1 >= (int)H) j1 -= (int)H;
    const uint8_t* t00 = img + ((size_t)j0 * W + i0) * 4;
    const uint8_t* t10 = img + ((size_t)j0 * W + i1) * 4;
    const uint8_t* t01 = img + ((size_t)j1 * W + i0) * 4;
    const uint8_t* t11 = img + ((size_t)j1 * W + i1) * 4;
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    /* un-normalised weighted sum of the raw byte values; the UNORM8 scale is applied once per sample */
    for (int ch = 0; ch < 4; ch++) {
        float c00 = (float)t00[ch], c10 = (float)t10[ch], c01 = (float)t01[ch], c11 = (float)t11[ch];
        out[ch] = ((w00 * c00 + w10 * c10) + w01 * c01) + w11 * c11;
    }
}

#define UNORM8_SCALE 0.003921568859368563f /* fp32 nearest to 1/255 */

/* GL 4.6 core 8.14: lambda <= 0 -> magnification (LINEAR on level 0);
 * else LINEAR_MIPMAP_LINEAR between floor(lambda) and floor(lambda)+1, clamped to the last level. */
static void sample_lod(const tex_t* t, float u, float v, float lambda, float out[4]) {
    float uf = frac_repeat(u), vf = frac_repeat(v);
    float q = (float)(t->n_levels - 1);
    float t1[4], t2[4];
    if (!(lambda > 0.0f)) { bilinear(t, 0, uf, vf, t1); for (int ch = 0; ch < 4; ch++) out[ch] = t1[ch] * UNORM8_SCALE; return; }
    if (lambda >= q) { bilinear(t, t->n_levels - 1, uf, vf, t1); for (int ch = 0; ch < 4; ch++) out[ch] = t1[ch] * UNORM8_SCALE; return; }
    float d = floorf(lambda), f = lambda - d;
    bilinear(t, (uint32_t)d, uf, vf, t1);
    bilinear(t, (uint32_t)d + 1, uf, vf, t2);
    for (int ch = 0; ch < 4; ch++) out[ch] = ((1.0f - f) * t1[ch] + f * t2[ch]) * UNORM8_SCALE;
}

void orc_sample(const uint8_t* mipchain, uint32_t w, uint32_t h, float u, float v, float lambda,
                float* out) {
    tex_t t; memset(&t, 0, sizeof t);
    t.chain = mipchain; t.w = w; t.h = h; t.n_levels = orc_mip_levels(w, h);
    uint64_t off = 0;
    for (uint32_t l = 0; l < t.n_levels; l++) { t.off[l] = off; off += (uint64_t)umax(1, w >> l) * umax(1, h >> l); }
    sample_lod(&t, u, v, lambda, out);
}

/* ------------------------------------------------------------------------------------ */
/* per-triangle setup                                                                    */
/* ------------------------------------------------------------------------------------ */
typedef struct {
    int64_t a[3], b[3], c[3]; /* E_i(Px,Py) = a*Px + b*Py + c on the 24.8 grid; interior > 0 */
    int bias[3];              /* 1: boundary of edge i belongs to the triangle            */
    int64_t area2;            /* > 0 when rasterisable                                    */
    int x0, x1, y0, y1;       /* inclusive pixel bbox (clamped to the viewport)           */
    float scale_x, scale_y;   /* |Ju|, |Jv|                                               */
    float rot[4];             /* (w,x,y,z)                                                */
    float ou[3], ov[3];       /* bbox-normalised orthogonal UVs (GS:353-399); gl_Position.xy = uv*2-1 */
    v3 p[3];
} tri_setup;

#define GUARD_PX 16384.0f

/* Fixed-function part (no reference code exists for it): GL 4.6 13.8.1 viewport transform with
 * glViewport(0,0,R,R) (ConversionPass.cpp:45): xw = (R/2)*xd + R/2; snap to 1/256 px, round-to-nearest-even;
 * integer edge functions, both windings (no culling, ConversionPass.cpp:48), top-left style ownership.
 * Input: the three clip-space positions' x,y (z = 0, w = 1).  Returns 0 if no fragment can result. */
static int raster_setup(const float ndx[3], const float ndy[3], uint32_t R, tri_setup* s) {
    float half = (float)R * 0.5f;
    int64_t X[3], Y[3];
    for (int i = 0; i < 3; i++) {
        float xw = half * ndx[i] + half, yw = half * ndy[i] + half;
        if (!(fabsf(xw) < GUARD_PX) || !(fabsf(yw) < GUARD_PX)) return 0; /* NaN/inf/absurd */
        X[i] = (int64_t)rintf(xw * 256.0f);
        Y[i] = (int64_t)rintf(yw * 256.0f);
    }
    /* edge i is opposite vertex i: E0 = orient(V1,V2,P), E1 = orient(V2,V0,P), E2 = orient(V0,V1,P) */
    static const int ea[3] = { 1, 2, 0 }, eb[3] = { 2, 0, 1 };
    for (int i = 0; i < 3; i++) {
        int64_t Ax = X[ea[i]], Ay = Y[ea[i]], Bx = X[eb[i]], By = Y[eb[i]];
        s->a[i] = -(By - Ay);
        s->b[i] = (Bx - Ax);
        s->c[i] = (By - Ay) * Ax - (Bx - Ax) * Ay;
    }
    int64_t area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
    if (area2 == 0) return 0;
    if (area2 < 0) { /* no culling (ConversionPass.cpp:48): accept both windings */
        area2 = -area2;
        for (int i = 0; i < 3; i++) { s->a[i] = -s->a[i]; s->b[i] = -s->b[i]; s->c[i] = -s->c[i]; }
    }
    s->area2 = area2;
    for (int i = 0; i < 3; i++) s->bias[i] = (s->a[i] > 0) || (s->a[i] == 0 && s->b[i] > 0);
    /* pixel bbox: centres 256*x+128 within [min,max], clamped to the R x R viewport */
    int64_t xmin = X[0], xmax = X[0], ymin = Y[0], ymax = Y[0];
    for (int i = 1; i < 3; i++) {
        if (X[i] < xmin) xmin = X[i];
        if (X[i] > xmax) xmax = X[i];
        if (Y[i] < ymin) ymin = Y[i];
        if (Y[i] > ymax) ymax = Y[i];
    }
    int64_t bx0 = (xmin - 128 + 255) >> 8, bx1 = (xmax - 128) >> 8;
    int64_t by0 = (ymin - 128 + 255) >> 8, by1 = (ymax - 128) >> 8;
    if (bx0 < 0) bx0 = 0;
    if (by0 < 0) by0 = 0;
    if (bx1 > (int64_t)R - 1) bx1 = (int64_t)R - 1;
    if (by1 > (int64_t)R - 1) by1 = (int64_t)R - 1;
    s->x0 = (int)bx0; s->x1 = (int)bx1; s->y0 = (int)by0; s->y1 = (int)by1;
    return bx0 <= bx1 && by0 <= by1;
}


/* converterGS.glsl:326-443 + viewport transform (ConversionPass.cpp:45) + pinned raster setup.
 * Returns 0 if the triangle can produce no fragments. */
static int setup_triangle(const float* v0, const float* v1, const float* v2, const float bmin[3],
                          const float bmax[3], uint32_t R, tri_setup* s) {
    v3 p0 = { v0[0], v0[1], v0[2] }, p1 = { v1[0], v1[1], v1[2] }, p2 = { v2[0], v2[1], v2[2] };
    s->p[0] = p0; s->p[1] = p1; s->p[2] = p2;
    /* GS:327-342 edges + longest-edge swap (strict >, else-if) */
    v3 e1 = v3sub(p1, p0), e2 = v3sub(p2, p0), e3 = v3sub(p2, p1);
    float l1 = v3len(e1), l2 = v3len(e2), l3 = v3len(e3);
    if (l2 > l1 && l2 > l3) { v3 t = e1; e1 = e2; e2 = t; }
    else if (l3 > l1 && l3 > l2) { v3 t = e1; e1 = e3; e3 = t; }
    (void)e3;
    /* GS:345-351 */
    v3 xa = v3normalize(e1);
    v3 nrm = v3normalize(v3cross(xa, e2));
    float ax = fabsf(nrm.x), ay = fabsf(nrm.y), az = fabsf(nrm.z);
    /* GS:353-399 triplanar axis + bbox-normalised orthogonal UVs */
    int A, B;
    if (ax > ay && ax > az) { A = 1; B = 2; }
    else if (ay > az) { A = 0; B = 2; }
    else { A = 0; B = 1; }
    float rangeA = bmax[A] - bmin[A], rangeB = bmax[B] - bmin[B];
    float range = fmaxf(rangeA, rangeB);
    /* u = rel / range: a true IEEE division, as the shader writes it (GS:362-363,375-376,388-389).  Checked
     * bit for bit against the reference's own GLSL run through glm (oracle/ref_glsl_check.cpp). */
    const float* pv[3] = { v0, v1, v2 };
    float ou[3], ov[3];
    for (int i = 0; i < 3; i++) {
        ou[i] = (pv[i][A] - bmin[A]) / range;
        ov[i] = (pv[i][B] - bmin[B]) / range;
        s->ou[i] = ou[i]; s->ov[i] = ov[i];
    }
    /* GS:401-407 rotation */
    v3 ya = v3normalize(v3cross(nrm, xa));
    float m[3][3] = { { xa.x, xa.y, xa.z }, { ya.x, ya.y, ya.z }, { nrm.x, nrm.y, nrm.z } };
    float q[4];
    quat_cast(m, q);
    s->rot[0] = q[3]; s->rot[1] = q[0]; s->rot[2] = q[1]; s->rot[3] = q[2];
    /* GS:269-300, 206-235, 409-430 Jacobian and scale (m[col][row]) */
    float U00 = ou[1] - ou[0], U10 = ou[2] - ou[0]; /* UVMatrix[0][0], UVMatrix[1][0] */
    float U01 = ov[1] - ov[0], U11 = ov[2] - ov[0]; /* UVMatrix[0][1], UVMatrix[1][1] */
    float det = U00 * U11 - U01 * U10;
    float I00 = 0.0f, I10 = 0.0f, I01 = 0.0f, I11 = 0.0f;
    if (det != 0.0f) {
        float invDet = 1.0f / det;
        I00 = U11 * invDet;
        I10 = -U10 * invDet;
        I01 = -U01 * invDet;
        I11 = U00 * invDet;
    }
    v3 V0 = v3sub(p1, p0), V1 = v3sub(p2, p0);
    v3 Ju = { V0.x * I00 + V1.x * I01, V0.y * I00 + V1.y * I01, V0.z * I00 + V1.z * I01 };
    v3 Jv = { V0.x * I10 + V1.x * I11, V0.y * I10 + V1.y * I11, V0.z * I10 + V1.z * I11 };
    s->scale_x = v3len(Ju);
    s->scale_y = v3len(Jv);

    /* GS:439 gl_Position = (uv*2-1, 0, 1) */
    float ndx[3], ndy[3];
    for (int i = 0; i < 3; i++) { ndx[i] = ou[i] * 2.0f - 1.0f; ndy[i] = ov[i] * 2.0f - 1.0f; }
    return raster_setup(ndx, ndy, R, s);
}
static inline int covered(const tri_setup* s, int x, int y, int64_t E[3]) {
    int64_t Px = 256 * (int64_t)x + 128, Py = 256 * (int64_t)y + 128;
    for (int i = 0; i < 3; i++) {
        E[i] = s->a[i] * Px + s->b[i] * Py + s->c[i];
        if (E[i] < 0 || (E[i] == 0 && !s->bias[i])) return 0;
    }
    return 1;
}

static uint32_t count_triangle(const tri_setup* s) {
    uint32_t n = 0;
    int64_t E[3];
    for (int y = s->y0; y <= s->y1; y++)
        for (int x = s->x0; x <= s->x1; x++) n += covered(s, x, y, E);
    return n;
}

typedef struct {
    float bmin[3], bmax[3], color[4];
    tex_t tex[3];
    const float* verts;
    uint32_t stride;
    uint64_t first_tri, n_tri;
} mesh_t;

/* DIAGNOSTIC switch (never set by the parity tests): 1 = compute the level of detail the way Mesa llvmpipe does,
 * lambda = 0.5 * fast_log2(rho^2) with fast_log2(x) = exponent(x) + (mantissa(x) - 1) — a piecewise-LINEAR log2
 * (gallivm lp_build_fast_log2), up to 0.043 below the true log2.  tools/ref_gl_decompose.py measured it: with this one
 * substitution the oracle reproduces llvmpipe's fp32 sampler to 4e-7 on minified textures (tests/test_ref_gl.py); without it
 * the two differ by up to 2e-2 on noise.  The pinned semantics stay the specification's log2 (GL 4.6 core 8.14.1, eq. 8.8). */
static int g_lod_mode = 0;
void orc_debug_set_lod_mode(int mode) { g_lod_mode = mode; }

/* level-of-detail for one texture: GL 4.6 core 8.14.1 eq. 8.7-8.8, derivatives exact (affine) */
static float lod_lambda(const tex_t* t, float dudx, float dvdx, float dudy, float dvdy) {
    float sx = dudx * (float)t->w, tx = dvdx * (float)t->h;
    float sy = dudy * (float)t->w, ty = dvdy * (float)t->h;
    if (g_lod_mode == 1) {
        const float r2 = fmaxf(sx * sx + tx * tx, sy * sy + ty * ty);
        int e;
        const float m = frexpf(r2, &e);          /* r2 = m * 2^e, m in [0.5, 1) */
        return r2 > 0.0f ? 0.5f * ((float)(e - 1) + (2.0f * m - 1.0f)) : -126.0f;
    }
    float rx = sqrtf(sx * sx + tx * tx), ry = sqrtf(sy * sy + ty * ty);
    float rho = fmaxf(rx, ry);
    return log2f(rho);
}

/* converterFS.glsl:44-104 for ONE fragment: f = the interpolated varyings (pos 0-2, normal 3-5, tangent 6-9,
 * uv 10-11), lam = the per-texture level of detail, flat inputs Scale/Quaternion from the GS. */
static void shade_fragment(const mesh_t* m, const float f[12], const float lam[3], float scale_x, float scale_y,
                           const float rot[4], float* o) {
    const float* P = f; const float* N = f + 3; const float* T = f + 6; const float* UV = f + 10;
    /* colour: FS:53-62,99 */
    float col[4] = { 1, 1, 1, 1 };
    if (m->tex[0].chain) sample_lod(&m->tex[0], UV[0], UV[1], lam[0], col);
    /* normal: FS:66-81 */
    float nout[3] = { N[0], N[1], N[2] };
    if (m->tex[1].chain) {
        float tn[4];
        sample_lod(&m->tex[1], UV[0], UV[1], lam[1], tn);
        v3 r = { tn[0] * 2.0f - 1.0f, tn[1] * 2.0f - 1.0f, tn[2] * 2.0f - 1.0f };
        r = v3normalize(r);
        v3 Nv = { N[0], N[1], N[2] }, Tv = { T[0], T[1], T[2] };
        v3 bt = v3normalize(v3cross(Nv, Tv));
        bt.x *= T[3]; bt.y *= T[3]; bt.z *= T[3];
        v3 Nn = v3normalize(Nv);
        v3 w = { (Tv.x * r.x + bt.x * r.y) + Nn.x * r.z, (Tv.y * r.x + bt.y * r.y) + Nn.y * r.z,
                 (Tv.z * r.x + bt.z * r.y) + Nn.z * r.z };
        w = v3normalize(w);
        nout[0] = w.x; nout[1] = w.y; nout[2] = w.z;
    }
    /* metallic-roughness: FS:87-95 */
    float metal = 0.1f, rough = 0.5f;
    if (m->tex[2].chain) {
        float mr[4];
        sample_lod(&m->tex[2], UV[0], UV[1], lam[2], mr);
        metal = mr[2]; rough = mr[1];
    }
    /* record: FS:98-103 */
    o[0] = P[0]; o[1] = P[1]; o[2] = P[2]; o[3] = 1.0f;
    for (int k = 0; k < 4; k++) o[4 + k] = col[k] * m->color[k];
    o[8] = scale_x; o[9] = scale_y; o[10] = 1e-7f; o[11] = 0.0f;
    o[12] = nout[0]; o[13] = nout[1]; o[14] = nout[2]; o[15] = 0.0f;
    o[16] = rot[0]; o[17] = rot[1]; o[18] = rot[2]; o[19] = rot[3];
    o[20] = metal; o[21] = rough; o[22] = 0.0f; o[23] = 1.0f;
}

/* converterFS.glsl:44-104 for every covered pixel of one triangle; returns fragments visited. */
static uint64_t emit_triangle(const tri_setup* s, const mesh_t* m, const float* v0, const float* v1,
                              const float* v2, uint64_t gtri, uint64_t base, uint64_t cap, float* out,
                              uint64_t out_capacity, uint64_t* keys) {
    /* attribute layout per vertex: pos 0-2, normal 3-5, tangent 6-9, uv 10-11 (converterVS.glsl:9-12) */
    float inva = 1.0f / (float)s->area2; /* pinned: barycentrics = (float)E_i * (1/(float)area2) */
    float g1x = (float)(s->a[1] * 256) * inva, g2x = (float)(s->a[2] * 256) * inva;
    float g1y = (float)(s->b[1] * 256) * inva, g2y = (float)(s->b[2] * 256) * inva;
    float du1 = v1[10] - v0[10], du2 = v2[10] - v0[10];
    float dv1 = v1[11] - v0[11], dv2 = v2[11] - v0[11];
    float dudx = g1x * du1 + g2x * du2, dvdx = g1x * dv1 + g2x * dv2;
    float dudy = g1y * du1 + g2y * du2, dvdy = g1y * dv1 + g2y * dv2;
    float lam[3] = { 0, 0, 0 };
    for (int k = 0; k < 3; k++)
        if (m->tex[k].chain) lam[k] = lod_lambda(&m->tex[k], dudx, dvdx, dudy, dvdy);

    uint64_t n = 0;
    int64_t E[3];
    for (int y = s->y0; y <= s->y1; y++) {
        for (int x = s->x0; x <= s->x1; x++) {
            if (!covered(s, x, y, E)) continue;
            uint64_t idx = base + n;
            n++;
            if (cap && idx >= cap) continue;      /* converterFS.glsl:49-51 */
            if (!out || idx >= out_capacity) continue;
            float l1 = (float)E[1] * inva, l2 = (float)E[2] * inva;
            float f[12];
            for (int k = 0; k < 12; k++) f[k] = (v0[k] + l1 * (v1[k] - v0[k])) + l2 * (v2[k] - v0[k]);
            shade_fragment(m, f, lam, s->scale_x, s->scale_y, s->rot, out + idx * ORC_RECORD_FLOATS);
            if (keys) keys[idx] = (gtri << 24) | ((uint64_t)y << 12) | (uint64_t)x;
        }
    }
    return n;
}

uint32_t orc_reference_cap(uint32_t R, uint32_t n_meshes) {
    /* ConversionPass.cpp:21-24: unsigned 32-bit product, then min with MAX_GAUSSIANS_TO_SORT */
    uint32_t mc = n_meshes > 1 ? n_meshes : 1;
    uint32_t mx = R * R * 6u * mc;
    return mx < 7000000u ? mx : 7000000u;
}

static void prepare_meshes(const orc_mesh* in, uint32_t n, mesh_t* out) {
    uint64_t first = 0;
    for (uint32_t i = 0; i < n; i++) {
        mesh_t* m = &out[i];
        memset(m, 0, sizeof *m);
        memcpy(m->bmin, in[i].bbox_min, 12);
        memcpy(m->bmax, in[i].bbox_max, 12);
        memcpy(m->color, in[i].base_color, 16);
        m->verts = in[i].vertices;
        m->stride = in[i].stride_floats;
        m->first_tri = first;
        m->n_tri = in[i].n_vertices / 3;
        first += m->n_tri;
        for (int k = 0; k < 3; k++) {
            const orc_texture* t = &in[i].tex[k];
            if (!t->rgba8 || !t->width || !t->height) continue;
            tex_t* d = &m->tex[k];
            d->w = t->width; d->h = t->height;
            d->owned = (uint8_t*)malloc(orc_mip_total_texels(d->w, d->h) * 4);
            d->n_levels = orc_build_mips(t->rgba8, d->w, d->h, d->owned, d->off);
            d->chain = d->owned;
        }
    }
}

static void free_meshes(mesh_t* m, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) for (int k = 0; k < 3; k++) free(m[i].tex[k].owned);
    free(m);
}

static inline const mesh_t* find_mesh(const mesh_t* ms, uint32_t n, uint64_t t) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) { uint32_t mid = (lo + hi) / 2; if (ms[mid].first_tri <= t) lo = mid; else hi = mid; }
    /* = last mesh with first_tri <= t; empty meshes sharing a first_tri are skipped by construction */
    return &ms[lo];
}

/* A prepared scene keeps the mip chains (== what glGenerateMipmap leaves on the GPU at load time), so
 * that a timed conversion excludes them exactly like the GPU path does. */
struct orc_scene { mesh_t* ms; uint32_t n; };

orc_scene* orc_scene_create(const orc_mesh* meshes, uint32_t n_meshes) {
    orc_scene* sc = (orc_scene*)malloc(sizeof *sc);
    sc->n = n_meshes;
    sc->ms = (mesh_t*)malloc(sizeof(mesh_t) * (n_meshes ? n_meshes : 1));
    prepare_meshes(meshes, n_meshes, sc->ms);
    return sc;
}

void orc_scene_destroy(orc_scene* sc) {
    if (!sc) return;
    free_meshes(sc->ms, sc->n);
    free(sc);
}

uint64_t orc_scene_convert(const orc_scene* sc, uint32_t R, uint64_t tri_first, uint64_t tri_count, uint64_t cap,
                           float* out, uint64_t out_capacity, uint64_t* keys, int n_threads) {
    const uint32_t n_meshes = sc->n;
    if (n_meshes == 0) return 0;
    const mesh_t* ms = sc->ms;
    uint64_t T = ms[n_meshes - 1].first_tri + ms[n_meshes - 1].n_tri;
    if (tri_first > T) tri_first = T;
    uint64_t tri_end = (tri_count == UINT64_MAX || tri_first + tri_count > T) ? T : tri_first + tri_count;
    uint64_t nt = tri_end - tri_first;
    uint64_t* offs = (uint64_t*)malloc(sizeof(uint64_t) * (nt + 1));
    (void)n_threads;
    /* pass 1: counts (ConversionPass.cpp:50-52 loops meshes in order; triangles in draw order) */
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads > 0 ? n_threads : 1)
    for (int64_t i = 0; i < (int64_t)nt; i++) {
        uint64_t t = tri_first + (uint64_t)i;
        const mesh_t* m = find_mesh(ms, n_meshes, t);
        const float* v = m->verts + (t - m->first_tri) * 3 * m->stride;
        tri_setup s;
        offs[i + 1] = setup_triangle(v, v + m->stride, v + 2 * m->stride, m->bmin, m->bmax, R, &s)
                          ? count_triangle(&s) : 0;
    }
    offs[0] = 0;
    for (uint64_t i = 0; i < nt; i++) offs[i + 1] += offs[i];
    uint64_t total = offs[nt];
    /* pass 2: emit at canonical offsets */
    if (out && out_capacity) {
#pragma omp parallel for schedule(dynamic, 1024) num_threads(n_threads > 0 ? n_threads : 1)
        for (int64_t i = 0; i < (int64_t)nt; i++) {
            if (offs[i + 1] == offs[i]) continue;
            uint64_t t = tri_first + (uint64_t)i;
            const mesh_t* m = find_mesh(ms, n_meshes, t);
            const float* v = m->verts + (t - m->first_tri) * 3 * m->stride;
            tri_setup s;
            setup_triangle(v, v + m->stride, v + 2 * m->stride, m->bmin, m->bmax, R, &s);
            emit_triangle(&s, m, v, v + m->stride, v + 2 * m->stride, t, offs[i], cap, out, out_capacity, keys);
        }
    }
    free(offs);
    return total;
}

uint64_t orc_convert(const orc_mesh* meshes, uint32_t n_meshes, uint32_t R, uint64_t tri_first,
                     uint64_t tri_count, uint64_t cap, float* out, uint64_t out_capacity, uint64_t* keys,
                     int n_threads) {
    if (n_meshes == 0) return 0;
    orc_scene* sc = orc_scene_create(meshes, n_meshes);
    uint64_t total = orc_scene_convert(sc, R, tri_first, tri_count, cap, out, out_capacity, keys, n_threads);
    orc_scene_destroy(sc);
    return total;
}

uint64_t orc_count_per_triangle(const orc_mesh* meshes, uint32_t n_meshes, uint32_t R, uint32_t* counts) {
    if (n_meshes == 0) return 0;
    mesh_t* ms = (mesh_t*)malloc(sizeof(mesh_t) * n_meshes);
    orc_mesh* notex = (orc_mesh*)malloc(sizeof(orc_mesh) * n_meshes);
    memcpy(notex, meshes, sizeof(orc_mesh) * n_meshes);
    for (uint32_t i = 0; i < n_meshes; i++) memset(notex[i].tex, 0, sizeof notex[i].tex);
    prepare_meshes(notex, n_meshes, ms);
    free(notex);
    uint64_t T = ms[n_meshes - 1].first_tri + ms[n_meshes - 1].n_tri, total = 0;
    for (uint64_t t = 0; t < T; t++) {
        const mesh_t* m = find_mesh(ms, n_meshes, t);
        const float* v = m->verts + (t - m->first_tri) * 3 * m->stride;
        tri_setup s;
        counts[t] = setup_triangle(v, v + m->stride, v + 2 * m->stride, m->bmin, m->bmax, R, &s)
                        ? count_triangle(&s) : 0;
        total += counts[t];
    }
    free_meshes(ms, n_meshes);
    return total;
}

/* ------------------------------------------------------------------------------------ */
/* test hooks for oracle/ref_glsl_check.cpp (the reference's shader source run as C++)     */
/* ------------------------------------------------------------------------------------ */
int orc_debug_gs(const float* v0, const float* v1, const float* v2, const float bmin[3], const float bmax[3],
                 uint32_t R, float ndc_xy[6], float scale_xyz[3], float rot_wxyz[4]) {
    tri_setup s;
    memset(&s, 0, sizeof s);
    int ok = setup_triangle(v0, v1, v2, bmin, bmax, R, &s);
    for (int i = 0; i < 3; i++) { ndc_xy[2 * i] = s.ou[i] * 2.0f - 1.0f; ndc_xy[2 * i + 1] = s.ov[i] * 2.0f - 1.0f; }
    scale_xyz[0] = s.scale_x; scale_xyz[1] = s.scale_y; scale_xyz[2] = 1e-7f;
    for (int i = 0; i < 4; i++) rot_wxyz[i] = s.rot[i];
    return ok;
}

/* Fixed-function stages for the software-GL harness (oracle/ref_pipeline_check.cpp): the fragments, in canonical
 * (row, column) order, of the triangle whose clip-space positions are (ndc_x, ndc_y, 0, 1).  xy[2i..] = pixel,
 * l12[2i..] = screen-linear weights of vertices 1 and 2, grad = d(l1)/dx, d(l2)/dx, d(l1)/dy, d(l2)/dy per pixel. */
uint64_t orc_debug_raster(const float ndc_xy[6], uint32_t R, uint64_t max_frag, int32_t* xy, float* l12, float grad[4]) {
    tri_setup s;
    memset(&s, 0, sizeof s);
    const float ndx[3] = { ndc_xy[0], ndc_xy[2], ndc_xy[4] }, ndy[3] = { ndc_xy[1], ndc_xy[3], ndc_xy[5] };
    if (!raster_setup(ndx, ndy, R, &s)) return 0;
    float inva = 1.0f / (float)s.area2;
    grad[0] = (float)(s.a[1] * 256) * inva; grad[1] = (float)(s.a[2] * 256) * inva;
    grad[2] = (float)(s.b[1] * 256) * inva; grad[3] = (float)(s.b[2] * 256) * inva;
    uint64_t n = 0;
    int64_t E[3];
    for (int y = s.y0; y <= s.y1; y++)
        for (int x = s.x0; x <= s.x1; x++) {
            if (!covered(&s, x, y, E)) continue;
            if (n < max_frag) {
                xy[2 * n] = x; xy[2 * n + 1] = y;
                l12[2 * n] = (float)E[1] * inva; l12[2 * n + 1] = (float)E[2] * inva;
            }
            n++;
        }
    return n;
}

float orc_debug_lod(uint32_t w, uint32_t h, float dudx, float dvdx, float dudy, float dvdy) {
    tex_t t;
    memset(&t, 0, sizeof t);
    t.w = w; t.h = h;
    return lod_lambda(&t, dudx, dvdx, dudy, dvdy);
}

void orc_debug_sample(const orc_scene* sc, uint32_t mesh, int slot, float u, float v, float lambda, float out[4]) {
    sample_lod(&sc->ms[mesh].tex[slot], u, v, lambda, out);
}

void orc_debug_fs(const orc_scene* sc, uint32_t mesh, const float varyings[12], const float lam[3],
                  const float scale_xy[2], const float rot_wxyz[4], float record[ORC_RECORD_FLOATS]) {
    shade_fragment(&sc->ms[mesh], varyings, lam, scale_xy[0], scale_xy[1], rot_wxyz, record);
}

/* ------------------------------------------------------------------------------------ */
/* PLY export  (SceneManager.cpp:651-678 -> parsers.cpp:631-651)                          */
/* ------------------------------------------------------------------------------------ */
#define SH_COEFF0 0.28209479177387814f /* params.hpp:17 */

/* utils.hpp:270 (std::clamp then -log(1/(a+1e-8)-1)) */
static float inv_sigmoid(float alpha) {
    alpha = (alpha < 0.0f) ? 0.0f : (1.0f < alpha) ? 1.0f : alpha;
    return -logf((1.0f / (alpha + 1e-8f)) - 1.0f);
}
/* glm::clamp(x,lo,hi) = min(max(x,lo),hi); glm::max(a,b) = (a<b)?b:a; glm::min(a,b) = (b<a)?b:a */
static float glm_clamp(float x, float lo, float hi) {
    float mx = (x < lo) ? lo : x;
    return (hi < mx) ? hi : mx;
}
/* parsers.cpp:370-375 */
static uint8_t to_byte(float v) {
    float clamped = glm_clamp(v, 0.0f, 1.0f);
    float rounded = roundf(clamped * 255.0f);
    return (uint8_t)rounded;
}
/* parsers.cpp:320-337 (note the joint sign test in OctWrap: both components flip together) */
static void encode_octa(const float nrm[3], float out[2]) {
    float d = fabsf(nrm[0]) + fabsf(nrm[1]) + fabsf(nrm[2]) + 1e-8f;
    float nx = nrm[0] / d, ny = nrm[1] / d, nz = nrm[2] / d;
    float rx, ry;
    if (nz >= 0.0f) { rx = nx; ry = ny; }
    else {
        float sgn = (nx >= 0 && ny >= 0) ? 1.0f : -1.0f;
        rx = (1.0f - fabsf(ny)) * sgn;
        ry = (1.0f - fabsf(nx)) * sgn;
    }
    out[0] = rx * 0.5f + 0.5f;
    out[1] = ry * 0.5f + 0.5f;
}

static void put_f(uint8_t** p, float v) { memcpy(*p, &v, 4); *p += 4; }

int orc_write_ply(const char* path, const float* records, uint64_t n, unsigned format, float scale_multiplier) {
    FILE* f = fopen(path, "wb");
    if (!f) return 1;
    if (format > 2) format = 0; /* parsers.cpp:646-648 default */
    fprintf(f, "ply\nformat binary_little_endian 1.0\nelement vertex %llu\n", (unsigned long long)n);
    if (format == 0) { /* parsers.cpp:438-466 */
        fprintf(f, "property float x\nproperty float y\nproperty float z\n"
                   "property float nx\nproperty float ny\nproperty float nz\n"
                   "property float f_dc_0\nproperty float f_dc_1\nproperty float f_dc_2\n");
        for (int i = 0; i <= 44; i++) fprintf(f, "property float f_rest_%d\n", i);
        fprintf(f, "property float opacity\nproperty float scale_0\nproperty float scale_1\nproperty float scale_2\n"
                   "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n");
    } else if (format == 1) { /* parsers.cpp:240-266 */
        fprintf(f, "property float x\nproperty float y\nproperty float z\n"
                   "property float nx\nproperty float ny\nproperty float nz\n"
                   "property float f_dc_0\nproperty float f_dc_1\nproperty float f_dc_2\n"
                   "property float metallicFactor\nproperty float roughnessFactor\n"
                   "property float opacity\nproperty float scale_0\nproperty float scale_1\nproperty float scale_2\n"
                   "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n");
    } else { /* parsers.cpp:342-367 */
        fprintf(f, "property float x\nproperty float y\nproperty float z\n"
                   "property uint8 red\nproperty uint8 green\nproperty uint8 blue\nproperty uint8 opacity\n"
                   "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n"
                   "property float scale_0\nproperty float scale_1\nproperty float scale_2\n"
                   "property uint8 octa_nx\nproperty uint8 octa_ny\n"
                   "property uint8 roughness\nproperty uint8 metallic\n");
    }
    fprintf(f, "end_header\n");
    const size_t row = format == 0 ? 62 * 4 : format == 1 ? 19 * 4 : 48;
    const size_t chunk = 4096;
    uint8_t* buf = (uint8_t*)malloc(row * chunk);
    for (uint64_t i0 = 0; i0 < n; i0 += chunk) {
        uint64_t cnt = n - i0 < chunk ? n - i0 : chunk;
        uint8_t* p = buf;
        for (uint64_t i = i0; i < i0 + cnt; i++) {
            const float* g = records + i * ORC_RECORD_FLOATS;
            const float *pos = g, *col = g + 4, *scl = g + 8, *nrm = g + 12, *rot = g + 16, *pbr = g + 20;
            if (format == 0 || format == 1) {
                put_f(&p, pos[0]); put_f(&p, pos[1]); put_f(&p, pos[2]);
                put_f(&p, nrm[0]); put_f(&p, nrm[1]); put_f(&p, nrm[2]);
                /* utils.cpp:45-49 getShFromColor */
                put_f(&p, (col[0] - 0.5f) / SH_COEFF0);
                put_f(&p, (col[1] - 0.5f) / SH_COEFF0);
                put_f(&p, (col[2] - 0.5f) / SH_COEFF0);
                if (format == 0) { memset(p, 0, 45 * 4); p += 45 * 4; }      /* parsers.cpp:485-489 */
                else { put_f(&p, pbr[0]); put_f(&p, pbr[1]); }                /* parsers.cpp:290-291 */
                put_f(&p, inv_sigmoid(col[3]));
                put_f(&p, logf(scl[0] * scale_multiplier));
                put_f(&p, logf(scl[1] * scale_multiplier));
                put_f(&p, logf(scl[2] * scale_multiplier));
                /* rotation vec4 is stored (w,x,y,z) so .x,.y,.z,.w is w,x,y,z: parsers.cpp:507-510 */
                put_f(&p, rot[0]); put_f(&p, rot[1]); put_f(&p, rot[2]); put_f(&p, rot[3]);
            } else { /* parsers.cpp:378-424 */
                put_f(&p, pos[0]); put_f(&p, pos[1]); put_f(&p, pos[2]);
                *p++ = to_byte(col[0]); *p++ = to_byte(col[1]); *p++ = to_byte(col[2]); *p++ = to_byte(col[3]);
                put_f(&p, rot[0]); put_f(&p, rot[1]); put_f(&p, rot[2]); put_f(&p, rot[3]);
                float minXY = (scl[1] < scl[0]) ? scl[1] : scl[0]; /* std::min(a,b) = (b<a)?b:a, parsers.cpp:403 */
                put_f(&p, logf(scl[0] * scale_multiplier));
                put_f(&p, logf(scl[1] * scale_multiplier));
                put_f(&p, logf(minXY * scale_multiplier));
                float oc[2];
                encode_octa(nrm, oc);
                *p++ = (uint8_t)glm_clamp(roundf(oc[0] * 255.0f), 0.0f, 255.0f);
                *p++ = (uint8_t)glm_clamp(roundf(oc[1] * 255.0f), 0.0f, 255.0f);
                *p++ = to_byte(pbr[1]); /* roughness */
                *p++ = to_byte(pbr[0]); /* metallic  */
            }
        }
        if (fwrite(buf, 1, (size_t)(p - buf), f) != (size_t)(p - buf)) { free(buf); fclose(f); return 2; }
    }
    free(buf);
    return fclose(f) ? 3 : 0;
}
