// m2s_viewmath.h — the first lines of the viewer prepass (gaussianSplattingPrepassCS.glsl:67-77), shared by k_prepass (m2s_prepass.hip) and
// by the key kernel of the depth sort that runs BEFORE it (m2s_sort.hip, m2s_prepass_sorted): world / view / clip position of a Gaussian and
// the frustum test, operation for operation — one definition, so that the sort orders by exactly the depth bits the prepass stores and
// culls exactly the records the prepass would cull.
#pragma once
#include <hip/hip_runtime.h>

#pragma clang fp contract(off)

namespace m2s {

// mat4 * vec4 in the shader's (glm's) association: (m0*x + m1*y) + (m2*z + m3*w), column-major m
__device__ __forceinline__ float4 m4_mul(const float* m, float x, float y, float z, float w) {
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = (m[0 + i] * x + m[4 + i] * y) + (m[8 + i] * z + m[12 + i] * w);
    return make_float4(r[0], r[1], r[2], r[3]);
}

// :67-77.  ws = u_modelToWorld * vec4(P, 1); vs = u_worldToView * vec4(ws.xyz, 1); pos2d = u_viewToClip * vs; inside = the guard-band
// frustum test (false = culled; every comparison is false for NaN: such a record is NOT culled here, as in the shader).
__device__ __forceinline__ bool view_project(const float* M, const float* V, const float* P, float px, float py, float pz, float4& ws, float4& vs, float4& pos2d) {
    ws = m4_mul(M, px, py, pz, 1.0f);                                             // :67
    vs = m4_mul(V, ws.x, ws.y, ws.z, 1.0f);                                       // :69
    pos2d = m4_mul(P, vs.x, vs.y, vs.z, vs.w);                                    // :71
    const float clip = 1.05f * pos2d.w;                                           // :73
    return !(pos2d.z < -clip || pos2d.x < -clip || pos2d.x > clip || pos2d.y < -clip || pos2d.y > clip);   // :75-77
}

}  // namespace m2s
