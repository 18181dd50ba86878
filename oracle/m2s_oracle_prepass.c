/*
 * m2s_oracle_prepass.c — CPU ORACLE for the viewer prepass.  TEST INFRASTRUCTURE: see m2s_oracle_prepass.h.
 *
 * Every function names the reference lines it restates.  fp32 throughout, no contraction (-ffp-contract=off), and the
 * operation order of the vendored glm 1.0.1 for every vector / matrix operator the shader uses, so that the result is
 * bit-identical to the shader executed through glm (oracle/ref_prepass_check.cpp).
 */
#include "m2s_oracle_prepass.h"

#include <math.h>
#include <string.h>

typedef struct { float x, y, z, w; } v4;
typedef struct { float c[3][3]; } m3;   /* c[col][row], like glm */
typedef struct { float c[4][4]; } m4;

/* ---- glm operators ------------------------------------------------------------------------------------------ */
/* glm/detail/type_mat4x4.inl:536-582: (m0*x + m1*y) + (m2*z + m3*w) per component */
static v4 m4_mul_v4(const m4* m, v4 v) {
    float r[4];
    for (int i = 0; i < 4; ++i) {
        const float a0 = m->c[0][i] * v.x + m->c[1][i] * v.y;
        const float a1 = m->c[2][i] * v.z + m->c[3][i] * v.w;
        r[i] = a0 + a1;
    }
    v4 o = { r[0], r[1], r[2], r[3] };
    return o;
}
/* glm/detail/type_mat3x3.inl:486-520 */
static m3 m3_mul(const m3* a, const m3* b) {
    m3 r;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 3; ++i)
            r.c[c][i] = a->c[0][i] * b->c[c][0] + a->c[1][i] * b->c[c][1] + a->c[2][i] * b->c[c][2];
    return r;
}
static m3 m3_transpose(const m3* a) {
    m3 r;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 3; ++i) r.c[c][i] = a->c[i][c];
    return r;
}
/* glm/detail/func_matrix.inl:322-344 */
static m3 m3_inverse(const m3* mm) {
#define M(c_, r_) mm->c[c_][r_]
    const float ood = 1.0f / (+M(0, 0) * (M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2))
                              - M(1, 0) * (M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2))
                              + M(2, 0) * (M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)));
    m3 r;
    r.c[0][0] = +(M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2)) * ood;
    r.c[1][0] = -(M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2)) * ood;
    r.c[2][0] = +(M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1)) * ood;
    r.c[0][1] = -(M(0, 1) * M(2, 2) - M(2, 1) * M(0, 2)) * ood;
    r.c[1][1] = +(M(0, 0) * M(2, 2) - M(2, 0) * M(0, 2)) * ood;
    r.c[2][1] = -(M(0, 0) * M(2, 1) - M(2, 0) * M(0, 1)) * ood;
    r.c[0][2] = +(M(0, 1) * M(1, 2) - M(1, 1) * M(0, 2)) * ood;
    r.c[1][2] = -(M(0, 0) * M(1, 2) - M(1, 0) * M(0, 2)) * ood;
    r.c[2][2] = +(M(0, 0) * M(1, 1) - M(1, 0) * M(0, 1)) * ood;
#undef M
    return r;
}
/* glm/detail/func_matrix.inl:347-405 */
static m4 m4_inverse(const m4* mm) {
#define M(c_, r_) mm->c[c_][r_]
    const float C00 = M(2, 2) * M(3, 3) - M(3, 2) * M(2, 3), C02 = M(1, 2) * M(3, 3) - M(3, 2) * M(1, 3), C03 = M(1, 2) * M(2, 3) - M(2, 2) * M(1, 3);
    const float C04 = M(2, 1) * M(3, 3) - M(3, 1) * M(2, 3), C06 = M(1, 1) * M(3, 3) - M(3, 1) * M(1, 3), C07 = M(1, 1) * M(2, 3) - M(2, 1) * M(1, 3);
    const float C08 = M(2, 1) * M(3, 2) - M(3, 1) * M(2, 2), C10 = M(1, 1) * M(3, 2) - M(3, 1) * M(1, 2), C11 = M(1, 1) * M(2, 2) - M(2, 1) * M(1, 2);
    const float C12 = M(2, 0) * M(3, 3) - M(3, 0) * M(2, 3), C14 = M(1, 0) * M(3, 3) - M(3, 0) * M(1, 3), C15 = M(1, 0) * M(2, 3) - M(2, 0) * M(1, 3);
    const float C16 = M(2, 0) * M(3, 2) - M(3, 0) * M(2, 2), C18 = M(1, 0) * M(3, 2) - M(3, 0) * M(1, 2), C19 = M(1, 0) * M(2, 2) - M(2, 0) * M(1, 2);
    const float C20 = M(2, 0) * M(3, 1) - M(3, 0) * M(2, 1), C22 = M(1, 0) * M(3, 1) - M(3, 0) * M(1, 1), C23 = M(1, 0) * M(2, 1) - M(2, 0) * M(1, 1);
    const float F0[4] = { C00, C00, C02, C03 }, F1[4] = { C04, C04, C06, C07 }, F2[4] = { C08, C08, C10, C11 };
    const float F3[4] = { C12, C12, C14, C15 }, F4[4] = { C16, C16, C18, C19 }, F5[4] = { C20, C20, C22, C23 };
    const float V0[4] = { M(1, 0), M(0, 0), M(0, 0), M(0, 0) }, V1[4] = { M(1, 1), M(0, 1), M(0, 1), M(0, 1) };
    const float V2[4] = { M(1, 2), M(0, 2), M(0, 2), M(0, 2) }, V3[4] = { M(1, 3), M(0, 3), M(0, 3), M(0, 3) };
    static const float SA[4] = { +1, -1, +1, -1 }, SB[4] = { -1, +1, -1, +1 };
    m4 inv;
    for (int i = 0; i < 4; ++i) {
        inv.c[0][i] = (V1[i] * F0[i] - V2[i] * F1[i] + V3[i] * F2[i]) * SA[i];
        inv.c[1][i] = (V0[i] * F0[i] - V2[i] * F3[i] + V3[i] * F4[i]) * SB[i];
        inv.c[2][i] = (V0[i] * F1[i] - V1[i] * F3[i] + V3[i] * F5[i]) * SA[i];
        inv.c[3][i] = (V0[i] * F2[i] - V1[i] * F4[i] + V2[i] * F5[i]) * SB[i];
    }
    const float d0 = M(0, 0) * inv.c[0][0], d1 = M(0, 1) * inv.c[1][0], d2 = M(0, 2) * inv.c[2][0], d3 = M(0, 3) * inv.c[3][0];
    const float ood = 1.0f / ((d0 + d1) + (d2 + d3));
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) inv.c[c][i] = inv.c[c][i] * ood;
#undef M
    return inv;
}
static float glm_min(float a, float b) { return (b < a) ? b : a; }   /* func_common.inl: min(x,y) = (y < x) ? y : x */
static float glm_max(float a, float b) { return (a < b) ? b : a; }
static float glm_clamp01(float x) { return glm_min(glm_max(x, 0.0f), 1.0f); }

/* ---- common.glsl -------------------------------------------------------------------------------------------- */
/* common.glsl:12-19 */
static float random2d(float cx, float cy) {
    const float a = 12.9898f, b = 78.233f, c = 43758.5453f;
    const float dt = cx * a + cy * b;
    const float sn = dt - 3.14f * floorf(dt / 3.14f);        /* mod(x, y) = x - y*floor(x/y) */
    const float v = sinf(sn) * c;
    return v - floorf(v);                                     /* fract */
}
/* common.glsl:21-46: quat = the stored vec4, i.e. (x,y,z,w) = rotation[0..3] = (qw,qx,qy,qz) */
static m3 cast_quat_to_mat3(const float q[4]) {
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    m3 r;
    r.c[0][0] = 1.f - 2.f * (z * z + w * w);
    r.c[0][1] = 2.f * (y * z - x * w);
    r.c[0][2] = 2.f * (y * w + x * z);
    r.c[1][0] = 2.f * (y * z + x * w);
    r.c[1][1] = 1.f - 2.f * (y * y + w * w);
    r.c[1][2] = 2.f * (z * w - x * y);
    r.c[2][0] = 2.f * (y * w - x * z);
    r.c[2][1] = 2.f * (z * w + x * y);
    r.c[2][2] = 1.f - 2.f * (y * y + z * z);
    return r;
}
/* common.glsl:48-59 */
static m3 compute_cov3d(const m3* rot, const float s[3]) {
    m3 sm;
    memset(&sm, 0, sizeof sm);
    sm.c[0][0] = s[0]; sm.c[1][1] = s[1]; sm.c[2][2] = s[2];
    const m3 mm = m3_mul(&sm, rot);
    const m3 mt = m3_transpose(&mm);
    return m3_mul(&mt, &mm);
}
/* common.glsl:78-82 */
static float exponential_depth(float view_depth, const float nf[2]) {
    const float nd = (view_depth - nf[0]) / (nf[1] - nf[0]);
    const float il = glm_clamp01(nd);
    return glm_clamp01(expf(-20.0f * il));
}

/* renderer.cpp:290-296: GL_NEAREST, GL_CLAMP_TO_EDGE, one level.  GL 4.6 §8.14.2: i = floor(u * W), clamped. */
static float depth_fetch(const orc_prepass_params* p, float u, float v) {
    const float fu = floorf(u * (float)p->depth_w), fv = floorf(v * (float)p->depth_h);
    long i = fu >= 0.0f ? (fu < (float)p->depth_w ? (long)fu : (long)p->depth_w - 1) : 0;   /* NaN -> 0 */
    long j = fv >= 0.0f ? (fv < (float)p->depth_h ? (long)fv : (long)p->depth_h - 1) : 0;
    return p->depth[(uint64_t)j * p->depth_w + (uint64_t)i];
}

static float len4(const float* v) { return sqrtf((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3])); }

/* gaussianSplattingPrepassCS.glsl:58-204 for one invocation; returns 1 when the Gaussian survives. */
static int prepass_one(const orc_prepass_params* p, const m4* M, const m4* V, const m4* P, float std_dev, const m4* MinvT,
                       const m3* model_rot_inv, uint32_t inv_x, uint32_t inv_y, const float* g, float* q, float* depth_out) {
    const float* gpos = g;            /* position */
    const float* gcol = g + 4;        /* color    */
    const float* gscl = g + 8;        /* scale    */
    const float* gnrm = g + 12;       /* normal   */
    const float* grot = g + 16;       /* rotation */
    const float* gpbr = g + 20;       /* pbr      */

    const v4 p1 = { gpos[0], gpos[1], gpos[2], 1.0f };
    const v4 ws = m4_mul_v4(M, p1);                                        /* :67 */
    const v4 ws1 = { ws.x, ws.y, ws.z, 1.0f };
    const v4 vs = m4_mul_v4(V, ws1);                                       /* :69 */
    v4 pos2d = m4_mul_v4(P, vs);                                           /* :71 */
    const float clip = 1.05f * pos2d.w;                                    /* :73 */
    if (pos2d.z < -clip || pos2d.x < -clip || pos2d.x > clip || pos2d.y < -clip || pos2d.y > clip) return 0;   /* :75-77 */

    if (p->depth_test_mesh == 1 && gcol[3] > .95f && p->format == 0) {     /* :80-92 */
        const float ndx = pos2d.x / pos2d.w, ndy = pos2d.y / pos2d.w;
        const float u = ndx * 0.5f + 0.5f, v = ndy * 0.5f + 0.5f;
        const float depth = depth_fetch(p, u, v);
        const float my_depth = (pos2d.z / pos2d.w) * 0.5f + 0.5f;
        const float eps = 0.00002f;
        if (my_depth > depth + eps) return 0;
    }

    const float multiplier = (p->format == 0 || p->format == 3) ? std_dev : 1.0f;   /* :94 */
    const float l0 = len4(M->c[0]), l1 = len4(M->c[1]);
    const float ms[3] = { l0, l0, l1 };                                    /* :95 (x, x, y — as written) */
    float scale[3];
    for (int i = 0; i < 3; ++i) scale[i] = (gscl[i] * multiplier) * (ms[i] * ms[i]);   /* :96 */

    m3 rot = cast_quat_to_mat3(grot);                                      /* :100 */
    rot = m3_mul(&rot, model_rot_inv);                                     /* :102-108 */
    const m3 cov3d = compute_cov3d(&rot, scale);                           /* :110 */

    float out_color[4] = { 0, 0, 0, 0 };
    float nrm[4] = { 1, 0, 0, 0 };
    const float computed_depth = exponential_depth(-vs.z, p->near_far);    /* :115 */

    if (p->format == 0 || (p->format == 1 && p->ply_has_pbr != 0) || p->format == 3) {   /* :118-122 */
        const v4 n1 = { gnrm[0], gnrm[1], gnrm[2], 1.0f };
        const v4 nw = m4_mul_v4(MinvT, n1);
        nrm[0] = nw.x * 0.5f + 0.5f; nrm[1] = nw.y * 0.5f + 0.5f; nrm[2] = nw.z * 0.5f + 0.5f; nrm[3] = gcol[3];
    } else if (p->format == 1) {                                           /* :124-131 */
        const uint32_t mi = (uint32_t)((gscl[1] < gscl[2]) && (gscl[1] < gscl[0])) + (uint32_t)((gscl[2] < gscl[1]) && (gscl[2] < gscl[0])) * 2u;
        nrm[0] = rot.c[mi][0] * 0.5f + 0.5f; nrm[1] = rot.c[mi][1] * 0.5f + 0.5f; nrm[2] = rot.c[mi][2] * 0.5f + 0.5f; nrm[3] = gcol[3];
    }

    if (p->render_mode == 0 || p->render_mode == 6) memcpy(out_color, gcol, sizeof out_color);            /* :133-137 */
    else if (p->render_mode == 1) { out_color[0] = out_color[1] = out_color[2] = computed_depth; out_color[3] = gcol[3]; }
    else if (p->render_mode == 2) memcpy(out_color, nrm, sizeof out_color);
    if (p->render_mode == 3) {                                             /* :146-149 */
        const float fx = (float)inv_x, fy = (float)inv_y;
        out_color[0] = random2d(fx, fy);
        out_color[1] = random2d(fy, fx);
        out_color[2] = random2d(fy * 1.234f, fx * 1.234f);
        out_color[3] = 1.0f;
    }

    pos2d.x = pos2d.x / pos2d.w; pos2d.y = pos2d.y / pos2d.w; pos2d.z = pos2d.z / pos2d.w;   /* :151 */

    const float p00 = P->c[0][0], p11 = P->c[1][1], p32 = P->c[3][2];
    const float rx = p->resolution[0], ry = p->resolution[1];
    const float tz_sq = vs.z * vs.z;                                       /* :154-159 */
    const float jsx = -(p00 * rx) / (2.0f * vs.z);
    const float jsy = -(p11 * ry) / (2.0f * vs.z);
    const float jtx = (p00 * vs.x * rx) / (2.0f * tz_sq);
    const float jty = (p11 * vs.y * ry) / (2.0f * tz_sq);
    const float jtz = ((p->near_far[1] - p->near_far[0]) * p32) / (2.0f * tz_sq);
    m3 J;
    memset(&J, 0, sizeof J);
    J.c[0][0] = jsx; J.c[1][1] = jsy; J.c[2][0] = jtx; J.c[2][1] = jty; J.c[2][2] = jtz;   /* :161-163 */
    m3 W;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 3; ++i) W.c[c][i] = V->c[c][i];                /* :165 */
    const m3 JW = m3_mul(&J, &W);
    const m3 JWt = m3_transpose(&JW);
    const m3 t0 = m3_mul(&JW, &cov3d);
    const m3 Vp = m3_mul(&t0, &JWt);                                       /* :168 */
    float c00 = Vp.c[0][0], c01 = Vp.c[0][1], c10 = Vp.c[1][0], c11 = Vp.c[1][1];   /* :170 */
    c00 += 0.3f;                                                           /* :173-174 */
    c11 += 0.3f;
    const float mid = c00 + c11;
    const float da = c00 - c11, db = 2.0f * c01;
    const float delta = sqrtf(da * da + db * db);                          /* :178 */
    const float lambda1 = 0.5f * (mid + delta), lambda2 = 0.5f * (mid - delta);
    if (lambda2 < 0.0f) return 0;                                          /* :183 */

    const float dvy = (-c00 + c01 + lambda1) / (c01 - c11 + lambda1);      /* :185 */
    const float inv_len = 1.0f / sqrtf(1.0f * 1.0f + dvy * dvy);           /* normalize = v * inversesqrt(dot(v,v)) */
    const float dx = 1.0f * inv_len, dy = dvy * inv_len;
    const float major_r = glm_min(3.0f * sqrtf(lambda1), 1024.0f), minor_r = glm_min(3.0f * sqrtf(lambda2), 1024.0f);
    const float mjx = major_r * dx, mjy = major_r * dy;                    /* :186-187 */
    const float mnx = minor_r * dy, mny = minor_r * (-dx);
    const float hx = rx * 0.5f, hy = ry * 0.5f;                            /* :189-190 */

    q[0] = pos2d.x; q[1] = pos2d.y; q[2] = pos2d.z; q[3] = pos2d.w;        /* :194 */
    q[4] = mjx / hx; q[5] = mjy / hy; q[6] = mnx / hx; q[7] = mny / hy;    /* :195 */
    memcpy(q + 8, out_color, 16);                                          /* :196 */
    const float det = c00 * c11 - c01 * c10;                               /* common.glsl:61-76 */
    float i00 = 0.0f, i01 = 0.0f, i11 = 0.0f;
    if (det != 0.0f) { i00 = c11 / det; i01 = -c01 / det; i11 = c00 / det; }
    q[12] = i00; q[13] = i01; q[14] = i11; q[15] = -vs.z;                  /* :199 */
    q[16] = nrm[0]; q[17] = nrm[1]; q[18] = nrm[2]; q[19] = gpbr[0];       /* :201 */
    q[20] = ws.x; q[21] = ws.y; q[22] = ws.z; q[23] = gpbr[1];             /* :202 */
    *depth_out = vs.z;                                                     /* :204 */
    return 1;
}

uint64_t orc_prepass(const orc_prepass_params* p, const float* records, uint64_t n, float* quads, float* depths) {
    m4 M, V, P;
    memcpy(&M, p->model_to_world, sizeof M);
    memcpy(&V, p->world_to_view, sizeof V);
    memcpy(&P, p->view_to_clip, sizeof P);
    const float std_dev = p->gaussian_std / (float)p->resolution_target;   /* GaussiansPrepass.cpp:18 */
    /* loop invariants of the shader: transpose(inverse(u_modelToWorld)) (:120), inverse(modelRotation) (:102-108) */
    const m4 Minv = m4_inverse(&M);
    m4 MinvT;
    for (int c = 0; c < 4; ++c)
        for (int i = 0; i < 4; ++i) MinvT.c[c][i] = Minv.c[i][c];
    m3 mr;
    for (int c = 0; c < 3; ++c)
        for (int i = 0; i < 3; ++i) mr.c[c][i] = M.c[c][i];
    const m3 mr_inv = m3_inverse(&mr);
    /* GaussiansPrepass.cpp:44-49 + the shader's 16x16 local size (:57): invocation ids of a linear index */
    const uint32_t total = (uint32_t)n;
    const uint32_t groups_needed = (total + 255u) / 256u;
    const uint32_t groups_x = (uint32_t)ceil(sqrt((float)groups_needed));
    const uint32_t global_w = groups_x * 16u;
    uint64_t k = 0;
    for (uint64_t gid = 0; gid < n; ++gid) {
        const uint32_t inv_x = global_w ? (uint32_t)(gid % global_w) : 0u, inv_y = global_w ? (uint32_t)(gid / global_w) : 0u;
        if (prepass_one(p, &M, &V, &P, std_dev, &MinvT, &mr_inv, inv_x, inv_y, records + gid * 24, quads + k * ORC_QUAD_FLOATS, depths + k)) ++k;
    }
    return k;
}
