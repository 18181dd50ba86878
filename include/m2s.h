/*
 * m2s.h — C ABI of the MI355X-native mesh -> 3D-Gaussian-splat conversion pass.
 *
 * This is the drop-in boundary for the ONE hot path of electronicarts/mesh2splat: the
 * per-triangle UV-space rasterisation conversion,
 *     class ConversionPass : IRenderPass { void execute(RenderContext&); }
 *     (src/renderer/renderPasses/ConversionPass.{hpp,cpp}, RenderPass.hpp:11-29)
 * whose device side is the GLSL program converter{VS,GS,FS}.glsl.  The reference has no
 * FFI layer (it is a C++ virtual call inside one executable); the entry points below are
 * what a binding for that call would need — the inputs ConversionPass reads from
 * RenderContext, the outputs it leaves there — with every GL object replaced by plain
 * pointers and sizes.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions: status-code returns, no exceptions cross the ABI, caller-owned input memory
 * is only borrowed for the duration of a call, device memory is owned by the context unless
 * the *_into variant is used, one context per host thread (like the single GL context).
 * All functions are implemented by hand-written HIP kernels for gfx950; there is NO CPU
 * fallback: on a machine without a usable device m2s_create fails with M2S_ERR_NO_DEVICE.
 */
#ifndef M2S_H
#define M2S_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define M2S_ABI_VERSION 1

typedef enum m2s_status {
    M2S_OK = 0,
    M2S_ERR_INVALID = 1,   /* bad argument                                              */
    M2S_ERR_NO_DEVICE = 2, /* no HIP device / device index out of range                 */
    M2S_ERR_HIP = 3,       /* a HIP runtime call failed (m2s_last_error has the string) */
    M2S_ERR_OOM = 4,       /* device or host allocation failed                          */
    M2S_ERR_CAPACITY = 5,  /* destination too small                                     */
    M2S_ERR_IO = 6,        /* file could not be written                                 */
    M2S_ERR_STATE = 7      /* call order violated (e.g. convert before upload)          */
} m2s_status;

typedef struct m2s_ctx m2s_ctx;

/* One RGBA8 image as tiny_gltf leaves it (always 4 components, tiny_gltf.h:2609; row 0 first,
 * no flip) == utils::TextureDataGl (utils.hpp:190-211).  rgba8 == NULL means "map absent"
 * (ConversionPass.cpp:77-99 leaves has*Map = 0). */
typedef struct m2s_texture {
    const uint8_t* rgba8;
    uint32_t width, height;
} m2s_texture;

enum { M2S_TEX_ALBEDO = 0, M2S_TEX_NORMAL = 1, M2S_TEX_METALLIC_ROUGHNESS = 2 }; /* tex units 0,1,2 */

/* One glTF primitive == one std::pair<utils::Mesh, utils::GLMesh> of
 * RenderContext::dataMeshAndGlMesh (RenderContext.hpp:86).
 *   vertices      : the VBO built by SceneManager::setupMeshBuffers (SceneManager.cpp:483-512),
 *                   de-indexed, 3 vertices per triangle; per vertex
 *                   pos3 normal3 tangent4 uv2 [normalizedUv2 scale3].  stride_floats is 17 for
 *                   the reference layout, 12 for the live attributes only (the trailing 5 floats
 *                   are never read: converterGS.glsl uses neither normalizedUv nor scale).
 *   bbox_min/max  : Mesh::bbox == u_bboxMin/u_bboxMax (ConversionPass.cpp:111-112).  NOTE the
 *                   reference computes it cumulatively over the mesh list (SceneManager.cpp:476-527).
 *   base_color    : material.baseColorFactor == u_materialFactor (ConversionPass.cpp:110).
 *   tex           : meshToTextureData[mesh.name][BASE_COLOR|NORMAL|METALLIC_ROUGHNESS]. */
typedef struct m2s_mesh {
    const float* vertices;
    uint32_t n_vertices;
    uint32_t stride_floats;
    float bbox_min[3];
    float bbox_max[3];
    float base_color[4];
    m2s_texture tex[3];
} m2s_mesh;

/* == utils::GaussianDataSSBO (utils.hpp:145-152) == GLSL GaussianVertex (converterFS.glsl:21-28):
 * position (P,1) | color rgba*factor | scale (|Ju|,|Jv|,1e-7,0) | normal (n,0) | rotation (w,x,y,z) |
 * pbr (metallic, roughness, 0, 1).  96 bytes, std430. */
typedef struct m2s_gaussian {
    float position[4];
    float color[4];
    float scale[4];
    float normal[4];
    float rotation[4];
    float pbr[4];
} m2s_gaussian;

/* ---- lifetime -------------------------------------------------------------------------------- */
uint32_t m2s_abi_version(void);
/* Creates a context on HIP device `device` (>= 0). */
m2s_status m2s_create(int device, m2s_ctx** out_ctx);
void m2s_destroy(m2s_ctx* ctx);
/* Message of the last failure on this context (never NULL).  ctx == NULL: last create failure. */
const char* m2s_last_error(const m2s_ctx* ctx);

/* ---- scene upload == SceneManager::setupMeshBuffers + glUtils::generateTextures ---------------- */
/* Restricts the context to triangles [first, first+count) of the flattened (mesh-major, draw-order)
 * triangle list of the NEXT m2s_upload_scene: the multi-GPU shard of this rank.  count ==
 * UINT64_MAX (default) means "to the end".  Mesh uniforms and textures are always complete. */
m2s_status m2s_set_triangle_range(m2s_ctx* ctx, uint64_t first, uint64_t count);
/* Copies vertices to HBM (re-laid out as 144 B/triangle SoA planes), uploads the mesh table and the
 * RGBA8 textures and generates mip levels 1..4 (glUtils.cpp:292-313).  Replaces any previous scene. */
m2s_status m2s_upload_scene(m2s_ctx* ctx, const m2s_mesh* meshes, uint32_t n_meshes);

/* Optional: allocate now what the first upload (M2S_PREPARE_UPLOAD: pinned + device staging chunks) and the first export
 * (M2S_PREPARE_EXPORT: pinned chunks for the device-to-host copies) would otherwise allocate on their own critical path;
 * M2S_PREPARE_KERNELS: have the HIP runtime load the code objects of every kernel now instead of inside the first conversion /
 * export / viewer pass of the process (m2s_upload_scene does it for the conversion pipeline it chooses in any case). */
enum { M2S_PREPARE_UPLOAD = 1, M2S_PREPARE_EXPORT = 2, M2S_PREPARE_KERNELS = 4 };
m2s_status m2s_prepare(m2s_ctx* ctx, uint32_t flags);

/* Wall-clock breakdown (ms) of the last m2s_upload_scene: [0] whole call, [1] geometry (pinned staging + re-layout kernel),
 * [2] textures (staging, mip and combo kernels, final sync), [3] device / pinned allocations. */
m2s_status m2s_last_upload_ms(const m2s_ctx* ctx, float out_ms[4]);

/* The resolutionTarget the NEXT m2s_upload_scene prepares the scene for (0, the default: the R this context last converted at,
 * else 1024).  The reference converts right after SceneManager::loadModel, at RenderContext::resolutionTarget
 * (guiRendererConcreteMediator.cpp:11-29, RenderContext.hpp:64) — and never twice at the same (scene, R), so its FIRST
 * conversion is the one a user waits for.  The upload therefore ends with one exact fragment count at this R behind its own
 * kernels (in [0] of m2s_last_upload_ms; m2s_last_warm_ms alone) and leaves behind what that first conversion needs: the
 * pipeline decision, the XCD band table, the record pool.  A conversion at any other R needs none of it (fragments scale with
 * R^2; its first launch runs without bands). */
m2s_status m2s_set_resolution_hint(m2s_ctx* ctx, uint32_t R);
float m2s_last_warm_ms(const m2s_ctx* ctx);

/* ---- the pass == ConversionPass::execute ------------------------------------------------------- */
/* u_maxGaussians policy (converterFS.glsl:46-51): -1 (default) = the reference formula
 * min(R*R*6*max(1,meshes), 7'000'000) in 32-bit unsigned arithmetic (ConversionPass.cpp:21-24);
 * 0 = unlimited; > 0 = explicit cap.  Records with index >= cap are not stored. */
m2s_status m2s_set_max_gaussians(m2s_ctx* ctx, int64_t cap);
/* Runs the conversion at resolutionTarget R into the context-owned record buffer (re-allocated when
 * its size changes, ConversionPass.cpp:25-33).  Synchronous like execute() (glFinish + counter
 * read-back, ConversionPass.cpp:54-59).  *out_total = value of the fragment counter, i.e. the number
 * of fragments generated — NOT clamped to the cap, exactly like renderContext.numberOfGaussians. */
m2s_status m2s_convert(m2s_ctx* ctx, uint32_t R, uint64_t* out_total);
/* Same, but records go to caller-owned DEVICE memory (e.g. a torch tensor feeding an RCCL gather),
 * kernels are enqueued on `hip_stream` (a hipStream_t, NULL = default stream) and the call returns
 * after that stream has drained.  capacity_records additionally bounds what is stored. */
m2s_status m2s_convert_into(m2s_ctx* ctx, uint32_t R, void* d_records, uint64_t capacity_records,
                            void* hip_stream, uint64_t* out_total);
/* Asynchronous form for back-to-back conversions (a frame loop, one rank of a multi-GPU job): submit enqueues one
 * conversion and returns without waiting; wait blocks until the OLDEST submitted conversion has finished and returns
 * its counter (it then becomes "the last conversion" for m2s_num_stored / m2s_download / m2s_export_ply ...).
 * At most M2S_MAX_IN_FLIGHT conversions may be in flight; they execute in submission order.  d_records == NULL uses
 * the context-owned buffer and the context's stream; otherwise the caller's buffer and `hip_stream`, as in
 * m2s_convert_into.  Conversions that write the same buffer overwrite each other in order, like repeated draws into
 * one SSBO.  The reference's execute() is synchronous (glFinish, ConversionPass.cpp:54): this pair is an extension
 * that removes the launch + completion round trip (16 us of the 197 us a C3 conversion takes) from the critical path.
 * The first conversion of a scene at a given R, and any conversion that needs the second stage or the multi-pass
 * pipeline, is executed synchronously inside submit (same results, no overlap). */
#define M2S_MAX_IN_FLIGHT 4
/* lanes = 2: context-owned submissions alternate between two streams, each with its own look-back chain and record
 * buffer, so consecutive single-kernel conversions OVERLAP (the tail of one, where the GPU drains, with the head of the
 * next: C3 0.132 -> 0.108 ms per conversion at three in flight) instead of running back to back with ~8 us between
 * dependent kernels; multi-pass conversions likewise — the second lane has its own offsets / slice starts / per-triangle
 * setup records, so k_count_scan of one conversion runs beside k_emit2 of the one before.  Records then alternate between two buffers; m2s_device_records / m2s_download / m2s_export_ply
 * follow the conversion last waited for.  Default 1 (one stream, one buffer). */
m2s_status m2s_set_async_lanes(m2s_ctx* ctx, int lanes);
m2s_status m2s_convert_submit(m2s_ctx* ctx, uint32_t R, void* d_records, uint64_t capacity_records, void* hip_stream);
m2s_status m2s_convert_wait(m2s_ctx* ctx, uint64_t* out_total);
/* Number of records actually stored by the last convert: min(total, cap[, capacity]). */
uint64_t m2s_num_stored(const m2s_ctx* ctx);
/* Device pointer of the context-owned records of the last m2s_convert (zero-copy consumers). */
const void* m2s_device_records(const m2s_ctx* ctx);
/* == glGetBufferSubData in SceneManager::exportPly (SceneManager.cpp:659-664). */
m2s_status m2s_download(m2s_ctx* ctx, m2s_gaussian* dst, uint64_t capacity_records);
/* Per-triangle fragment counts of the last convert (shard balancing / diagnostics); n = triangles in range. */
m2s_status m2s_download_triangle_counts(m2s_ctx* ctx, uint32_t* dst, uint64_t n);

/* ---- export == SceneManager::exportPly + parsers::savePlyVector -------------------------------- */
/* format 0 standard 3DGS (62 floats/row), 1 PBR (19 floats/row), 2 compressed PBR (48 B/row); any
 * other value behaves like 0 (parsers.cpp:646-648).  scale_multiplier = gaussianStd / R. */
m2s_status m2s_write_ply(const char* path, const m2s_gaussian* records, uint64_t n, uint32_t format,
                         float scale_multiplier);
/* Downloads the last convert's records and writes them: scaleMultiplier = gaussian_std / R
 * (SceneManager.cpp:668). */
m2s_status m2s_export_ply(m2s_ctx* ctx, const char* path, uint32_t format, float gaussian_std);

/* Several writers, one file (one per GPU rank of a sharded conversion: no record gather needed): writes n host records as
 * rows [first_row, first_row + n) of a .ply whose header announces total_rows.  The file is created if absent and never
 * truncated below its final size; the writer of row 0 also writes the header and sets the file's length.  Writers may run
 * concurrently (different processes); the result is byte-identical to m2s_write_ply of the concatenated records. */
m2s_status m2s_write_ply_slice(const char* path, const m2s_gaussian* records, uint64_t n, uint32_t format, float scale_multiplier,
                               uint64_t first_row, uint64_t total_rows);
/* Same for the first n_rows records of the context's last conversion (n_rows <= m2s_num_stored). */
m2s_status m2s_export_ply_slice(m2s_ctx* ctx, const char* path, uint32_t format, float gaussian_std, uint64_t first_row,
                                uint64_t n_rows, uint64_t total_rows);

/* ---- multi-GPU: triangle-range shards, one process per GPU, RCCL over xGMI ------------------------------------------
 * The reference is a single-GPU program; this block implements BASELINE.json's sharding: every triangle is independent
 * (no depth test, no blending: ConversionPass.cpp:45-48), the reference's one shared word — the append cursor,
 * converterFS.glsl:46 — becomes one counter per rank, and concatenating the per-rank record blocks in rank order
 * reproduces the single-GPU output byte for byte (the device emits in canonical (triangle, row, column) order).
 * Per rank: m2s_dist_shard_ranges -> m2s_set_triangle_range + m2s_upload_scene (cap lifted: m2s_set_max_gaussians(0)) ->
 * m2s_convert* -> m2s_dist_all_gather_counts (8 bytes per rank) -> either m2s_export_ply_slice at the rank's offset (no
 * record leaves its GPU) or m2s_dist_gather_records (every block to every rank / to one root, exact sizes, one RCCL
 * group of sends and receives).  librccl is opened at run time; without it these calls fail with M2S_ERR_STATE. */
#define M2S_DIST_ID_BYTES 128
typedef struct m2s_dist m2s_dist;
/* Rank 0 obtains the communicator id (== ncclGetUniqueId) and hands the 128 bytes to the other ranks by any means. */
m2s_status m2s_dist_unique_id(uint8_t out_id[M2S_DIST_ID_BYTES]);
/* Collective over all ranks (== ncclCommInitRank) on HIP device `device`. */
m2s_status m2s_dist_create(int device, const uint8_t id[M2S_DIST_ID_BYTES], int rank, int world, m2s_dist** out);
/* The same exchange with the ranks as THREADS of one process (the shape of the reference: one executable), one context and
 * one GPU each; no RCCL involved: counts go through a table of the group, records through hipMemcpyPeerAsync (the same xGMI
 * links).  The id of a new group of `world` ranks; every rank then calls m2s_dist_create with it, every m2s_dist_* call
 * below works unchanged, and the group is released by its last m2s_dist_destroy. */
m2s_status m2s_dist_local_id(int world, uint8_t out_id[M2S_DIST_ID_BYTES]);
void m2s_dist_destroy(m2s_dist* d);
int m2s_dist_rank(const m2s_dist* d);
int m2s_dist_world(const m2s_dist* d);
/* What moves the bytes of this communicator: "in-process", "rccl", or "rccl:<path>" when the environment variable M2S_RCCL_PATH
 * named the library (it takes precedence over an RCCL already in the process and over the system's). */
const char* m2s_dist_transport(const m2s_dist* d);
/* d == NULL: message of the last failed m2s_dist_unique_id / _create / _shard_ranges on this thread. */
const char* m2s_dist_last_error(const m2s_dist* d);
/* Host only, deterministic (every rank computes the same plan): cuts the flattened triangle list into `world` contiguous
 * ranges of about equal cost = estimated fragments at density R (projected area on the dominant axis plane, in pixels)
 * + 0.25 per triangle.  first[r], count[r] feed m2s_set_triangle_range on rank r. */
m2s_status m2s_dist_shard_ranges(const m2s_mesh* meshes, uint32_t n_meshes, uint32_t R, int world, uint64_t* first, uint64_t* count);
/* The one mandatory exchange: every rank learns every rank's counter.  counts[world]; offsets[world + 1] (may be NULL) =
 * exclusive prefix = where each rank's block starts in the merged buffer.  publish/collect is the non-blocking form for
 * back-to-back conversions (up to 8 exchanges in flight, completed in order, on a stream of their own). */
m2s_status m2s_dist_all_gather_counts(m2s_dist* d, uint64_t my_total, uint64_t* counts, uint64_t* offsets);
m2s_status m2s_dist_publish_count(m2s_dist* d, uint64_t my_total);
m2s_status m2s_dist_collect_counts(m2s_dist* d, uint64_t* counts, uint64_t* offsets);
/* Global cap semantics of a sharded conversion (converterFS.glsl:46-51): the merged buffer keeps the first `cap` records
 * in rank order (0 = unlimited); keep[r] = how many records rank r contributes. */
void m2s_dist_clamp_to_cap(const uint64_t* counts, int world, uint64_t cap, uint64_t* keep);
/* Record exchange: rank r's block (counts[r] records at d_mine on rank r) lands at record offset sum(counts[0..r)) of
 * d_merged on every rank (root < 0) or on `root` only (d_merged may be NULL elsewhere).  Device pointers; enqueued on
 * hip_stream (the stream the conversion ran on); returns without waiting for it. */
m2s_status m2s_dist_gather_records(m2s_dist* d, const void* d_mine, const uint64_t* counts, void* d_merged, int root, void* hip_stream);
/* Depth sort (== m2s_sort_by_depth == RadixSortPass.cpp:8-90) of records spread over the ranks — BASELINE config 5's "final
 * radix sort of the merged splat buffer" without merging it on one GPU: sample sort with one exact-size record exchange.
 * Collective.  In: every rank's context holds its block (its last conversion, or m2s_set_records).  Out: the context's sorted
 * buffer (m2s_device_sorted_records, m2s_download_sorted) holds this rank's contiguous slice of the globally sorted sequence —
 * *out_n records starting at position *out_offset; the slices in rank order are bit-identical to one GPU sorting the
 * rank-major concatenation (stable: ties keep rank, then original position).  The context's current records become the
 * received, not yet sorted ones. */
m2s_status m2s_dist_sort_by_depth(m2s_dist* d, m2s_ctx* ctx, const float world_to_view[16], uint64_t* out_n, uint64_t* out_offset);
/* Blocks until everything enqueued on hip_stream (the record exchange) has completed. */
m2s_status m2s_dist_wait(m2s_dist* d, void* hip_stream);

/* ---- depth sort == RadixSortPass::execute (RadixSortPass.cpp:8-90) ------------------------------------ */
/* Sorts the records stored by the last m2s_convert by key = floatBitsToUint(view-space z), ascending on the
 * raw bits (radixSortPrepass.glsl:23-33; z = row 2 of world_to_view * (P,1), world_to_view column-major like
 * glm), stable, and gathers the 96-byte records into a second context-owned buffer (radixSortGather.glsl).
 * *out_n = number of records sorted. */
m2s_status m2s_sort_by_depth(m2s_ctx* ctx, const float world_to_view[16], uint64_t* out_n);
/* A caller that is going to sort what it converts (BASELINE config 5: "final radix sort of the merged splat buffer") says so before
 * converting: where the conversion kernel can (k_sparse: the scenes of tens of millions of records), it then also leaves the records'
 * positions behind as a compact 16-byte plane, and the first m2s_sort_by_depth of those records builds its keys from 16 B per record
 * instead of touching every 128-byte line of the 96-byte records (RadixSortPass.cpp:16-90 sorts every frame; this is the first frame).
 * Costs the conversion 16 more bytes written per record; changes no record.  Default: off. */
m2s_status m2s_set_keep_positions(m2s_ctx* ctx, int enabled);
/* 1 if the position plane of the CURRENT records exists (left by their conversion, or by an earlier m2s_sort_by_depth of them). */
int m2s_positions_ready(const m2s_ctx* ctx);
const void* m2s_device_sorted_records(const m2s_ctx* ctx);
const void* m2s_device_sorted_keys(const m2s_ctx* ctx);      /* uint32[n], ascending: the keys of those records */
uint64_t m2s_num_sorted(const m2s_ctx* ctx);
uint32_t m2s_last_resolution(const m2s_ctx* ctx);            /* R of the current records (0: none / uploaded records) */
m2s_status m2s_download_sorted(m2s_ctx* ctx, m2s_gaussian* dst, uint64_t capacity_records);
/* Duration (ms) of the last profiled sort (key build + radix sort + gather). */
float m2s_last_sort_ms(const m2s_ctx* ctx);
/* The three stages of the last m2s_sort_by_depth, ms (m2s_set_profiling on): [0] keys, [1] radix sort, [2] gather of the 96-byte
 * records.  The first sort after the records changed reads the positions out of the records (every line of the buffer) and leaves
 * them behind as a compact plane; later sorts of the same records — the reference sorts every frame — build their keys from that. */
m2s_status m2s_last_sort_stage_ms(const m2s_ctx* ctx, float out_ms[3]);

/* ---- records from elsewhere == Renderer::updateGaussianBuffer after SceneManager::loadPly --------------- */
/* Makes `n` host records (e.g. the output of m2s_read_ply) the context's current records, as the reference does with a
 * loaded .ply (guiRendererConcreteMediator.cpp:30-34 -> glUtils::fillGaussianBufferSsbo, glUtils.cpp:676-684):
 * m2s_num_stored / m2s_device_records / m2s_download / m2s_prepass / m2s_sort_by_depth then refer to them.  The next
 * conversion replaces them.  (The viewer treats such records as format 1: set m2s_prepass_params.format accordingly.) */
m2s_status m2s_upload_records(m2s_ctx* ctx, const m2s_gaussian* records, uint64_t n);

/* Records that already live in device memory (e.g. the merged buffer of m2s_dist_gather_records) become the context's current
 * records without a copy: m2s_num_stored / m2s_download / m2s_export_ply (scale multiplier = std / R) / m2s_prepass /
 * m2s_sort_by_depth then refer to them.  The memory stays the caller's and must outlive those calls.
 * The context caches what it derives from the current records (m2s_sort_by_depth: the plane of positions its keys are built from)
 * under (pointer, n, a counter bumped by every conversion / m2s_upload_records / m2s_set_records).  A caller that REWRITES a buffer
 * in place — the one adopted here, or one handed to m2s_convert_into by a kernel of its own — must call m2s_set_records again
 * afterwards: without it later sorts would order the new records by the old positions. */
m2s_status m2s_set_records(m2s_ctx* ctx, const void* d_records, uint64_t n, uint32_t R);
/* Room for n records in the context-owned pool (grow-only, like the conversion's own allocation); *out_ptr = device address. */
m2s_status m2s_reserve_records(m2s_ctx* ctx, uint64_t n, void** out_ptr);

/* ---- viewer prepass == GaussiansPrepass::execute (GaussiansPrepass.cpp:8-56) ---------------------------- */
/* What the reference's compute shader gaussianSplattingPrepassCS.glsl:58-204 (+ common.glsl) does to every Gaussian
 * before it is drawn: transform, frustum cull (1.05 * w guard band), optional test against the mesh depth texture,
 * 3D covariance -> 2D covariance (+0.3 low-pass) -> conic and the two quad axes in NDC, debug colour modes; survivors
 * are appended to a QuadNdcTransformation array and a view-space depth array (the RadixSortPass key source).
 * Fields mirror the RenderContext members the pass reads (RenderContext.hpp:34-111).  Matrices are column-major (glm). */
typedef struct m2s_prepass_params {
    float world_to_view[16];     /* viewMat   -> u_worldToView                                              */
    float view_to_clip[16];      /* projMat   -> u_viewToClip                                               */
    float model_to_world[16];    /* modelMat  -> u_modelToWorld                                             */
    int32_t resolution[2];       /* rendererResolution (glm::ivec2) -> u_resolution                         */
    float near_far[2];           /* nearPlane, farPlane -> u_nearFar                                        */
    float gaussian_std;          /* gaussianStd;  u_stdDev = gaussianStd / float(resolutionTarget)          */
    uint32_t resolution_target;  /* resolutionTarget                                                        */
    int32_t render_mode;         /* renderMode: 0 colour, 1 depth, 2 normal, 3 geometry (per-invocation hash), 6 as 0; else black */
    uint32_t format;             /* 0 mesh2splat, 1 classic 3DGS .ply, 2 compressed PBR (3 treated like 0)  */
    uint32_t ply_has_pbr;        /* plyHasPbr                                                               */
    uint32_t depth_test_mesh;    /* performMeshDepthTest: 1 = cull opaque (alpha > .95) format-0 Gaussians behind `depth` */
    const float* depth;          /* meshDepthTexture: depth_w x depth_h window-space depth in [0,1], row 0 = bottom of the
                                    window (GL texture orientation), sampled GL_NEAREST / CLAMP_TO_EDGE (renderer.cpp:290-296).
                                    HOST memory unless depth_on_device; read only when depth_test_mesh == 1               */
    uint32_t depth_w, depth_h;
    uint32_t depth_on_device;    /* 1: `depth` is a device pointer on the context's device (no copy)        */
    uint32_t arrival_order;      /* 0: survivors in input order (reproducible).  1: in arrival order, as the reference's atomic
                                    append leaves them (same set, nondeterministic order; ~1.4x faster)                  */
} m2s_prepass_params;

/* == QuadNdcTransformation (gaussianSplattingPrepassCS.glsl:17-24), 96 bytes */
typedef struct m2s_quad {
    float mean2d_ndc[4];     /* clip position, xyz divided by w; w kept                   */
    float quad_scale_ndc[4]; /* major axis xy, minor axis xy, in NDC units                */
    float color[4];          /* per render mode                                           */
    float conic[4];          /* inverse 2D covariance (xx, xy, yy), view depth (-z)       */
    float normal[4];         /* encoded normal xyz, metallic                              */
    float ws_pos[4];         /* world position xyz, roughness                             */
} m2s_quad;

/* Runs the prepass over `n` records at device pointer `d_records` (96-byte m2s_gaussian each), or over the records of the
 * context's last conversion when d_records is NULL (n ignored).  Survivors are stored in context-owned buffers in INPUT
 * order (the reference appends through an atomic counter, i.e. in arrival order; RadixSortPass reorders them anyway).
 * Synchronous, like the reference's dispatch + the counter read-back that follows it (RadixSortPass.cpp:18-22).
 * *out_visible = number of survivors (the reference's atomic counter). */
m2s_status m2s_prepass(m2s_ctx* ctx, const m2s_prepass_params* params, const void* d_records, uint64_t n, uint64_t* out_visible);
const void* m2s_device_quads(const m2s_ctx* ctx);            /* m2s_quad[visible]  (perQuadTransformationsBuffer)   */
const void* m2s_device_prepass_depths(const m2s_ctx* ctx);   /* float[visible]     (gaussianDepthPostFiltering)     */
m2s_status m2s_download_prepass(m2s_ctx* ctx, m2s_quad* dst_quads, float* dst_depths, uint64_t capacity);
/* Duration (ms) of the last profiled prepass kernel. */
float m2s_last_prepass_ms(const m2s_ctx* ctx);

/* == RadixSortPass::execute (RadixSortPass.cpp:8-90) on the output of the last m2s_prepass: key = the raw bits of the
 * view-space depths (radixSortPrepass.glsl:23-33), ascending, stable (LSD radix like glu::RadixSort); the quads are gathered
 * into a second context-owned buffer in that order (radixSortGather.glsl:30-49).  *out_n = number sorted = what the
 * reference writes to drawElementsIndirectCommand.instanceCount (count = 6, first = baseInstance = 0). */
m2s_status m2s_sort_prepass(m2s_ctx* ctx, uint64_t* out_n);
const void* m2s_device_sorted_quads(const m2s_ctx* ctx);     /* m2s_quad[n]  (perQuadTransformationBufferSorted) */
m2s_status m2s_download_sorted_quads(m2s_ctx* ctx, m2s_quad* dst, uint64_t capacity);
/* Duration (ms) of the last profiled m2s_sort_prepass (radix sort + gather). */
float m2s_last_sort_prepass_ms(const m2s_ctx* ctx);
/* One frame's GaussiansPrepass::execute + RadixSortPass::execute (GaussiansPrepass.cpp:8-56, RadixSortPass.cpp:8-90) as ONE pass over the
 * records of the context (its last conversion, or m2s_set_records / m2s_upload_records): the depth sort is taken FIRST — keys = the bits of
 * the view-space depth the prepass is about to store (from the 16-byte position plane the context keeps of its current records), stable
 * radix sort of (key, record index) — and the prepass then reads the records THROUGH that permutation and appends its survivors in that
 * order: the 96-byte gather of RadixSortPass::gatherPost (radixSortGather.glsl:30-49) and the prepass's own read become one pass.  Without a depth
 * image (depth_test_mesh == 0, or format != 0) the sort applies the prepass's frustum test itself — it depends on the position alone — and the
 * prepass runs over the survivors only.
 * Result: m2s_device_sorted_quads / m2s_download_sorted_quads hold exactly what m2s_prepass (input order) followed by m2s_sort_prepass
 * leaves there, byte for byte; *out_visible = their number.  params->arrival_order is ignored (the order IS the result); the unsorted quads
 * of m2s_prepass are not produced.  m2s_last_sort_stage_ms: [0] keys, [1] radix sort, [2] the prepass through the permutation. */
m2s_status m2s_prepass_sorted(m2s_ctx* ctx, const m2s_prepass_params* params, uint64_t* out_visible);

/* ---- scene I/O == SceneManager::loadModel (minus GL) and parsers::loadPlyFile ------------------------ */
/* Host-side scene loaded from a binary glTF file: scene-graph transforms applied, de-indexed 17-float
 * vertex buffers, fallback normals/tangents, cumulative bboxes, RGBA8 textures (PNG) — exactly what
 * SceneManager::parseGltfFile/setupMeshBuffers/loadTextures leave in RenderContext
 * (SceneManager.cpp:195-649).  The returned m2s_mesh array can be passed to m2s_upload_scene and stays
 * valid until m2s_free_host_scene. */
typedef struct m2s_host_scene m2s_host_scene;
m2s_status m2s_load_glb(const char* path, m2s_host_scene** out_scene);
void m2s_free_host_scene(m2s_host_scene* scene);
uint32_t m2s_host_scene_num_meshes(const m2s_host_scene* scene);
const m2s_mesh* m2s_host_scene_meshes(const m2s_host_scene* scene);
const char* m2s_host_scene_mesh_name(const m2s_host_scene* scene, uint32_t i);   /* "<mesh name>_<counter>" */
const char* m2s_host_scene_warnings(const m2s_host_scene* scene);                 /* skipped primitives etc. */
/* Reads a binary .ply written by format 0 or 1 back into records (parsers.cpp:516-629): scale = exp,
 * alpha = sigmoid, colour = SH -> RGB, quaternion normalised.  Free with m2s_free_records. */
m2s_status m2s_read_ply(const char* path, m2s_gaussian** out_records, uint64_t* out_n, int* out_has_pbr);
void m2s_free_records(m2s_gaussian* records);
/* Message of the last failed scene-I/O call on this thread. */
const char* m2s_io_last_error(void);

/* ---- pipeline selection ------------------------------------------------------------------------ */
/* AUTO (default) decides once per (scene, R), from an exact fragment count taken at the first conversion:
 *   - fewer than 11 fragments per triangle on average: the SINGLE-PASS kernel (k_fused2 / its lean form k_fused3; the multi-pass
 *     pipeline from an R on at which a workgroup's fragments do not fit the kernel's LDS stream).  It emits every triangle of <= 16 pixel rows and <= 96 fragments
 *     itself; larger triangles only reserve their slice of the ordered output there and are emitted by a second
 *     kernel (k_emit_big, one workgroup per 1024-fragment chunk); a scene DOMINATED by such triangles falls through
 *     to the multi-pass pipeline;
 *   - otherwise the MULTI-PASS pipeline count -> scan -> offsets -> emit (output-range balanced, any triangle size).
 *   - a mesh far finer than the density asked for (BASELINE config 5) — fewer than 1.75 fragments per triangle on average
 *     in a scene of at least 2 M triangles, fewer than 0.5 in a smaller one (measured crossovers): the SPARSE form of the
 *     single-pass kernel (k_sparse): a cheap conservative test drops the triangles that cannot cover a pixel centre before
 *     the exact per-triangle phase runs on the survivors; k_fused2 where a workgroup does not fit.
 * MULTIPASS forces the multi-pass pipeline, TEAM / LEAN / SPARSE force the single-pass kernel in one of its forms.  Every
 * setting produces bit-identical output.  Changing the setting forgets the remembered decisions. */
enum { M2S_PIPELINE_AUTO = 0, M2S_PIPELINE_MULTIPASS = 1,
       /* 2: the one-wave-per-batch kernel of rounds 1-5 (k_fused), removed in round 6: m2s_set_pipeline rejects the value */
       M2S_PIPELINE_TEAM = 3 /* always the single-pass kernel, workgroup-cooperative form (k_fused2); the multi-pass pipeline where a
                                workgroup does not fit its LDS stream */,
       M2S_PIPELINE_SPARSE = 4 /* always the sparse form of the single-pass kernel (k_sparse) where the scene is large enough for
                                  it (>= ~172 k triangles), k_fused2 / the multi-pass pipeline where a workgroup does not fit its LDS stream */,
       M2S_PIPELINE_LEAN = 5 /* the team kernel in its lean form (k_fused3: four waves per SIMD; shades triangles of at most 8 x 8
                                pixels itself and defers the rest to k_emit_big) where the scene allows it — every mesh samples
                                three equally sized maps or none —, else as TEAM.  AUTO prefers it to k_fused2 under the same
                                condition and returns to k_fused2 at an R where many triangles were deferred */ };
m2s_status m2s_set_pipeline(m2s_ctx* ctx, int pipeline);
/* Which pipeline the last conversion actually ran: M2S_PIPELINE_MULTIPASS, _TEAM (k_fused2), _LEAN (k_fused3) or _SPARSE (k_sparse); 0 before any. */
int m2s_last_pipeline(const m2s_ctx* ctx);

/* ---- measurement ------------------------------------------------------------------------------- */
enum { M2S_K_COUNT = 0, M2S_K_SCAN = 1, M2S_K_OFFSETS = 2, M2S_K_EMIT = 3, M2S_K_FUSED = 4, M2S_K_N = 5 };
/* When enabled every convert brackets each kernel with hipEvents on its stream. */
m2s_status m2s_set_profiling(m2s_ctx* ctx, int enabled);
/* Kernel durations (ms) of the last profiled convert, indexed by M2S_K_*. */
m2s_status m2s_last_kernel_ms(const m2s_ctx* ctx, float out_ms[M2S_K_N]);
/* Scene facts for roofline accounting: triangles in range, meshes. */
uint64_t m2s_num_triangles(const m2s_ctx* ctx);
/* Test hook: sets the counter of single-pass launches (its low 16 bits tag the look-back chain words), so that a test
 * can walk a context across the wrap of that tag without issuing 65 536 conversions.  Requires an idle context. */
m2s_status m2s_debug_set_launch_counter(m2s_ctx* ctx, uint32_t value);

#ifdef __cplusplus
}
#endif
#endif /* M2S_H */
