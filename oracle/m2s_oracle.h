/*
 * m2s_oracle.h — CPU ORACLE for the mesh -> 3DGS conversion pass.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (mesh2splat_amd/) never
 * links, imports or calls anything under oracle/.
 *
 * It is a plain-C, fp32 restatement of the reference's conversion pass:
 *   src/shaders/conversion/converterGS.glsl:326-443   (per-triangle setup)
 *   src/shaders/conversion/converterFS.glsl:44-104    (per-fragment shading + record)
 *   src/renderer/renderPasses/ConversionPass.cpp:9-117 (driver: cap, per-mesh uniforms)
 *   src/utils/glUtils.cpp:292-313                      (sampler state)
 *   src/parsers/parsers.cpp:232-514, src/utils/utils.cpp:45-49, utils.hpp:270 (PLY export)
 * plus the OpenGL 4.6 fixed-function behaviour the shaders rely on (viewport transform,
 * pixel-centre rasterisation, screen-linear attribute interpolation, trilinear REPEAT
 * sampling, GenerateMipmap), pinned as written in DESIGN.md section "Pinned semantics".
 *
 * PARITY.  The reference ships no tests, golden vectors or sample assets for this path, and its implementation
 * (GLSL on an OpenGL 4.6 driver inside a Windows GUI app) cannot run here.  The oracle is pinned on the reference
 * itself wherever reference CODE exists, by compiling that code from where it lies (oracle/Makefile -> oracle/_ref/):
 *   - converterVS/GS/FS.glsl executed as C++ through the vendored glm: gl_Position, Scale, Quaternion and the
 *     24-float record agree BIT FOR BIT (oracle/ref_glsl_check.cpp, tests/test_ref_glsl.py);
 *   - parsers.cpp / SceneManager.cpp / tiny_gltf / stb_image with GL stubbed: PLY writers byte for byte, PLY reader
 *     and .glb loader bit for bit (oracle/ref_host_check.cpp, tests/test_ref_host.py);
 *   - the whole path (SceneManager::loadModel -> ConversionPass::execute -> shaders -> SceneManager::exportPly) on a
 *     minimal software GL built from this oracle's fixed-function stages: counter, cap, every record and the .ply
 *     agree bit for bit with orc_convert / orc_write_ply (oracle/ref_pipeline_check.cpp, tests/test_ref_pipeline.py);
 *   - glm::quat_cast and the glm node transforms (oracle/ref_glm_check.cpp, ref_glm_xform_check.cpp).
 * The reference's outputs are committed as fixtures (tests/golden/ref_host/, generator alongside).
 * PARITY UNPINNED for the fixed-function part only — pixel coverage, varying interpolation, mip generation, LOD and
 * trilinear filtering have no reference code (the GL driver does them); they follow the GL 4.6 specification as
 * pinned in DESIGN.md and are frozen by hand-derived known-answer tests (tests/test_oracle_kat.py).
 */
#ifndef M2S_ORACLE_H
#define M2S_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    const uint8_t* rgba8; /* tightly packed RGBA8, row 0 first; NULL = absent */
    uint32_t width, height;
} orc_texture;

typedef struct {
    const float* vertices;   /* de-indexed: 3 vertices per triangle                     */
    uint32_t n_vertices;     /* multiple of 3                                            */
    uint32_t stride_floats;  /* >= 12: pos3 normal3 tangent4 uv2 [normalizedUv2 scale3]  */
    float bbox_min[3], bbox_max[3]; /* the mesh's u_bboxMin/u_bboxMax uniforms           */
    float base_color[4];     /* u_materialFactor                                         */
    orc_texture tex[3];      /* 0 albedo, 1 normal, 2 metallic-roughness                 */
} orc_mesh;

/* 24 floats per record == utils::GaussianDataSSBO (utils.hpp:145-152):
 * position(4) color(4) scale(4) normal(4) rotation(4) pbr(4). */
#define ORC_RECORD_FLOATS 24

/* Reference cap formula, ConversionPass.cpp:21-24 (32-bit unsigned arithmetic). */
uint32_t orc_reference_cap(uint32_t R, uint32_t n_meshes);

/*
 * Convert.  Records are produced in canonical order (mesh, triangle, pixel row y, pixel x).
 *   tri_first/tri_count : restrict to a range of the flattened (mesh-major) triangle list;
 *                         tri_count == UINT64_MAX means "to the end".
 *   cap                 : records with canonical index >= cap are not stored (cap 0 = unlimited)
 *   out / out_capacity  : may be NULL / 0 to only count
 *   keys                : optional, one u64 per stored record: (global_tri << 24) | (y << 12) | x
 *   n_threads           : >1 uses OpenMP over triangles (two-pass count/emit), results identical
 * Returns the total number of fragments (like the reference's atomic counter, NOT clamped).
 */
uint64_t orc_convert(const orc_mesh* meshes, uint32_t n_meshes, uint32_t R,
                     uint64_t tri_first, uint64_t tri_count, uint64_t cap,
                     float* out, uint64_t out_capacity, uint64_t* keys, int n_threads);

/* Same conversion on a prepared scene (mip chains built once, like the GPU upload): used by bench.py's
 * cpu_baseline so that the timed region matches the GPU one (geometry + textures resident -> records). */
typedef struct orc_scene orc_scene;
orc_scene* orc_scene_create(const orc_mesh* meshes, uint32_t n_meshes);
void orc_scene_destroy(orc_scene* sc);
uint64_t orc_scene_convert(const orc_scene* sc, uint32_t R, uint64_t tri_first, uint64_t tri_count, uint64_t cap,
                           float* out, uint64_t out_capacity, uint64_t* keys, int n_threads);

/* Test hooks used by oracle/ref_glsl_check.cpp, which runs the reference's GLSL source as C++ and compares
 * stage by stage: the GS outputs of one triangle (gl_Position.xy of the three vertices, Scale, Quaternion as
 * stored, i.e. w,x,y,z), one texture fetch, and the FS for one set of interpolated varyings. */
int orc_debug_gs(const float* v0, const float* v1, const float* v2, const float bmin[3], const float bmax[3],
                 uint32_t R, float ndc_xy[6], float scale_xyz[3], float rot_wxyz[4]);
uint64_t orc_debug_raster(const float ndc_xy[6], uint32_t R, uint64_t max_frag, int32_t* xy, float* l12, float grad[4]);
float orc_debug_lod(uint32_t w, uint32_t h, float dudx, float dvdx, float dudy, float dvdy);
/* diagnostic only: 0 = pinned (log2 of rho), 1 = Mesa llvmpipe's 0.5 * fast_log2(rho^2); see m2s_oracle.c */
void orc_debug_set_lod_mode(int mode);
void orc_debug_sample(const orc_scene* sc, uint32_t mesh, int slot, float u, float v, float lambda, float out[4]);
void orc_debug_fs(const orc_scene* sc, uint32_t mesh, const float varyings[12], const float lam[3],
                  const float scale_xy[2], const float rot_wxyz[4], float record[ORC_RECORD_FLOATS]);

/* Per-triangle fragment counts only (for shard balancing tests). counts has n_triangles entries. */
uint64_t orc_count_per_triangle(const orc_mesh* meshes, uint32_t n_meshes, uint32_t R,
                                uint32_t* counts);

/* Mip chain, levels 0..min(4, floor(log2(max(w,h)))), 2x2 box filter, round-half-up.
 * Returns number of levels; level_offsets (in texels) gets n_levels entries; dst must hold
 * orc_mip_total_texels(w,h) * 4 bytes. */
uint32_t orc_mip_levels(uint32_t w, uint32_t h);
uint64_t orc_mip_total_texels(uint32_t w, uint32_t h);
uint32_t orc_build_mips(const uint8_t* rgba8, uint32_t w, uint32_t h, uint8_t* dst,
                        uint64_t* level_offsets);

/* Trilinear REPEAT sample with explicit lod lambda (exposed for tests). out = 4 floats. */
void orc_sample(const uint8_t* mipchain, uint32_t w, uint32_t h, float u, float v,
                float lambda, float* out);

/* PLY writers, byte-for-byte restatement of parsers.cpp:232-514.
 * format 0 standard 3DGS (62 floats), 1 PBR (19 floats), 2 compressed PBR (48 bytes).
 * Returns 0 on success.  records are NOT modified. */
int orc_write_ply(const char* path, const float* records, uint64_t n, unsigned format,
                  float scale_multiplier);

#ifdef __cplusplus
}
#endif
#endif
