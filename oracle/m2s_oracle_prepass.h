/*
 * m2s_oracle_prepass.h — CPU ORACLE for the viewer prepass.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as m2s_oracle.h: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it).
 *
 * Plain-C fp32 restatement of
 *   src/shaders/rendering/gaussianSplattingPrepassCS.glsl:58-204   (cull, cov3D -> 2D conic, quad axes)
 *   src/shaders/rendering/common.glsl:12-92                         (random2d, castQuatToMat3, computeCov3D, inverseMat2,
 *                                                                    computeExponentialDepth, encodeNormal)
 *   src/renderer/renderPasses/GaussiansPrepass.cpp:8-56             (u_stdDev = gaussianStd / resolutionTarget, the
 *                                                                    dispatch shape that defines gl_GlobalInvocationID)
 *   src/renderer/renderer.cpp:280-308                               (mesh depth texture: GL_NEAREST, CLAMP_TO_EDGE)
 * Matrix and vector arithmetic follows the vendored glm 1.0.1 (thirdParty/glm) operation for operation, because that is
 * what the pin executes.
 *
 * PARITY: pinned on the reference itself — the compute shader and GaussiansPrepass::execute are compiled from where they
 * lie and run on a minimal software GL (oracle/ref_prepass_check.cpp, tests/test_ref_prepass.py); visible count, every
 * QuadNdcTransformation and every depth agree bit for bit.  PARITY UNPINNED only for what GL leaves to the driver here:
 * nearest-texel selection of the depth texture (GL 4.6 §8.14.2, floor(u*W) clamped) and the arrival order of the atomic
 * append, which the oracle and the product fix as input order.
 */
#ifndef M2S_ORACLE_PREPASS_H
#define M2S_ORACLE_PREPASS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    float world_to_view[16];   /* RenderContext::viewMat, column-major (glm)                   */
    float view_to_clip[16];    /* RenderContext::projMat                                        */
    float model_to_world[16];  /* RenderContext::modelMat                                       */
    float resolution[2];       /* RenderContext::rendererResolution                             */
    float near_far[2];         /* nearPlane, farPlane                                           */
    float gaussian_std;        /* RenderContext::gaussianStd                                    */
    uint32_t resolution_target;/* RenderContext::resolutionTarget                               */
    int32_t render_mode;       /* u_renderMode                                                  */
    uint32_t format;           /* u_format                                                      */
    uint32_t ply_has_pbr;      /* u_plyHasPbr                                                   */
    uint32_t depth_test_mesh;  /* u_depthTestMesh                                               */
    const float* depth;        /* mesh depth texture (window-space depth, row 0 = bottom) or NULL */
    uint32_t depth_w, depth_h;
} orc_prepass_params;

/* 24 floats per output == QuadNdcTransformation (gaussianSplattingPrepassCS.glsl:17-24):
 * gaussianMean2dNdc(4) quadScaleNdc(4) color(4) conic(4) normal(4) wsPos(4). */
#define ORC_QUAD_FLOATS 24

/* Runs the prepass over n records (24 floats each, GaussianVertex).  Survivors are appended in input order:
 * quads[24*k], depths[k].  Returns the number of survivors (the atomic counter). */
uint64_t orc_prepass(const orc_prepass_params* p, const float* records, uint64_t n, float* quads, float* depths);

#ifdef __cplusplus
}
#endif
#endif
