// ref_pipeline_check.cpp — TEST INFRASTRUCTURE.  Runs the reference's WHOLE conversion path on the CPU:
//
//     SceneManager::loadModel      (SceneManager.cpp:22-35: tiny_gltf parse, vertex upload, textures)
//  -> ConversionPass::execute      (ConversionPass.cpp:9-117: cap, SSBO sizing, per-mesh uniforms, draws)
//       -> converterVS/GS/FS.glsl  (the reference's shaders, as C++ through glm: ref_glsl_env.h)
//  -> SceneManager::exportPly      (SceneManager.cpp:651-678 -> parsers::savePlyVector)
//
// all compiled from where they lie under /root/reference (oracle/Makefile).  The only thing that is NOT the
// reference's is OpenGL itself, which does not exist on this machine: this file implements the ~40 GL entry points
// that path touches as a minimal software GL.  State handling (buffers, vertex arrays, textures, uniforms, the
// atomic counter, the shader-storage buffer) is plain bookkeeping; the fixed-function stages of glDrawArrays —
// viewport transform, rasterisation, varying interpolation, level-of-detail, texture filtering, mip generation —
// call the oracle's pinned implementations (orc_debug_raster, orc_debug_lod, orc_build_mips, orc_sample).  So a
// match between this program's output and orc_convert() says: given our fixed-function semantics, the oracle
// reproduces the reference's host orchestration + shaders exactly (draw order, cumulative bounding boxes, cap and
// counter behaviour, texture-presence flags, material factors, record layout, PLY export).
//
//   ref_pipeline_check in.glb R out_records.bin [out.ply format gaussianStd]
//
// out_records.bin: u32 counter (as read back by ConversionPass.cpp:56-59), u32 maxGaussians (the u_maxGaussians
// uniform), u64 SSBO size in bytes, then min(counter, maxGaussians) records of 96 bytes.
#include "utils/SceneManager.hpp"
#include "renderer/renderPasses/ConversionPass.hpp"

#include "ref_glsl_env.h"

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <thread>

extern "C" {
#include "m2s_oracle.h"
}

#include "ref_swgl.h"

extern "C" {
// ---- the draw call: vertex fetch -> VS -> primitive assembly -> GS -> raster -> FS -------------------------------
static const swgl::Texture* g_unit_tex[8];
static float g_unit_lambda[8];
static glm::vec4 fetch_texel(int unit, glm::vec2 uv) {
    const swgl::Texture* t = g_unit_tex[unit];
    if (!t || !t->w) return glm::vec4(0, 0, 0, 1);   // GL: incomplete texture samples (0,0,0,1)
    float o[4];
    orc_sample(t->chain.data(), t->w, t->h, uv.x, uv.y, g_unit_lambda[unit], o);
    return glm::vec4(o[0], o[1], o[2], o[3]);
}

void GLAPIENTRY glDrawArrays(GLenum mode, GLint first, GLsizei count) {
    using namespace swgl;
    if (mode != GL_TRIANGLES) { fprintf(stderr, "swgl: only GL_TRIANGLES\n"); exit(70); }
    ++n_draws;
    const uint32_t R = (uint32_t)viewport[2];
    if (viewport[0] != 0 || viewport[1] != 0 || viewport[2] != viewport[3]) { fprintf(stderr, "swgl: unexpected viewport\n"); exit(70); }
    if (caps[GL_CULL_FACE] || caps[GL_DEPTH_TEST]) { fprintf(stderr, "swgl: culling / depth test enabled\n"); exit(70); }
    const VertexArray& va = vaos[bound_vao];
    const std::vector<uint8_t>& vb = buffers[va.array_buffer];
    auto attr = [&](int i, GLint v, int k) -> float {
        const auto& a = va.attr[i];
        if (!a.enabled || k >= a.size) return k == 3 ? 1.0f : 0.0f;
        float f;
        std::memcpy(&f, vb.data() + a.offset + (size_t)v * (size_t)a.stride + (size_t)k * 4, 4);
        return f;
    };
    // uniforms -> the shaders' globals (values exactly as the reference's host code set them)
    auto u3 = [&](const char* n) { const auto& v = uniform_f[n]; return glm::vec3(v.at(0), v.at(1), v.at(2)); };
    ref_gs::u_bboxMin = u3("u_bboxMin");
    ref_gs::u_bboxMax = u3("u_bboxMax");
    ref_fs::hasAlbedoMap = uniform_i["hasAlbedoMap"];
    ref_fs::hasNormalMap = uniform_i["hasNormalMap"];
    ref_fs::hasMetallicRoughnessMap = uniform_i["hasMetallicRoughnessMap"];
    { const auto& v = uniform_f["u_materialFactor"]; ref_fs::u_materialFactor = glm::vec4(v.at(0), v.at(1), v.at(2), v.at(3)); }
    ref_fs::u_maxGaussians = uniform_i["u_maxGaussians"];
    ref_fs::albedoTexture.unit = uniform_i.count("albedoTexture") ? uniform_i["albedoTexture"] : 0;
    ref_fs::normalTexture.unit = uniform_i.count("normalTexture") ? uniform_i["normalTexture"] : 0;
    ref_fs::metallicRoughnessTexture.unit = uniform_i.count("metallicRoughnessTexture") ? uniform_i["metallicRoughnessTexture"] : 0;
    for (int u = 0; u < 8; ++u) {
        g_unit_tex[u] = nullptr;
        auto it = unit_binding.find((GLuint)u);
        if (it != unit_binding.end() && it->second && textures.count(it->second)) {
            const Texture& t = textures[it->second];
            // the sampler state the oracle pins must be what the reference asks for (glUtils.cpp:305-312)
            auto par = [&](GLenum p) { auto q = t.params.find(p); return q == t.params.end() ? -1 : q->second; };
            if (par(GL_TEXTURE_WRAP_S) != GL_REPEAT || par(GL_TEXTURE_WRAP_T) != GL_REPEAT || par(GL_TEXTURE_MIN_FILTER) != GL_LINEAR_MIPMAP_LINEAR ||
                par(GL_TEXTURE_MAG_FILTER) != GL_LINEAR || par(GL_TEXTURE_BASE_LEVEL) != 0 || par(GL_TEXTURE_MAX_LEVEL) != 4 || !t.mipmapped) {
                fprintf(stderr, "swgl: sampler state differs from the pinned one\n");
                exit(71);
            }
            g_unit_tex[u] = &t;
        }
    }
    // output bindings (converterFS.glsl:22-26)
    std::vector<uint8_t>& ssbo = buffers[indexed_binding[{ GL_SHADER_STORAGE_BUFFER, 0 }]];
    std::vector<uint8_t>& counter = buffers[indexed_binding[{ GL_ATOMIC_COUNTER_BUFFER, 1 }]];
    const size_t ssbo_records = ssbo.size() / sizeof(ref_fs::GaussianVertex);
    ref_fs::gaussianBuffer.vertices = reinterpret_cast<ref_fs::GaussianVertex*>(ssbo.data());
    std::memcpy(&ref_fs::g_validCounter.v, counter.data(), 4);

    std::vector<int32_t> xy;
    std::vector<float> l12;
    for (GLint t = first; t + 2 < first + count; t += 3) {
        for (int i = 0; i < 3; ++i) {   // vertex fetch (converterVS.glsl:9-14) + VS
            const GLint v = t + i;
            ref_vs::position = glm::vec3(attr(0, v, 0), attr(0, v, 1), attr(0, v, 2));
            ref_vs::normal = glm::vec3(attr(1, v, 0), attr(1, v, 1), attr(1, v, 2));
            ref_vs::tangent = glm::vec4(attr(2, v, 0), attr(2, v, 1), attr(2, v, 2), attr(2, v, 3));
            ref_vs::uv = glm::vec2(attr(3, v, 0), attr(3, v, 1));
            ref_vs::normalizedUv = glm::vec2(attr(4, v, 0), attr(4, v, 1));
            ref_vs::scale = glm::vec3(attr(5, v, 0), attr(5, v, 1), attr(5, v, 2));
            ref_vs::main_();
            ref_gs::gs_in[i].position = ref_vs::vs_out.position; ref_gs::gs_in[i].normal = ref_vs::vs_out.normal;
            ref_gs::gs_in[i].tangent = ref_vs::vs_out.tangent; ref_gs::gs_in[i].uv = ref_vs::vs_out.uv;
            ref_gs::gs_in[i].normalizedUv = ref_vs::vs_out.normalizedUv; ref_gs::gs_in[i].scale = ref_vs::vs_out.scale;
        }
        glsl_env::g_emitted.clear();
        ref_gs::main_();
        const auto& E = glsl_env::g_emitted;
        if (E.size() != 3) { fprintf(stderr, "swgl: GS emitted %zu vertices\n", E.size()); exit(70); }
        // ---- fixed function: clip (w = 1, z = 0: nothing to clip in depth), viewport, rasterise
        const float ndc[6] = { E[0].gl_Position.x, E[0].gl_Position.y, E[1].gl_Position.x, E[1].gl_Position.y, E[2].gl_Position.x, E[2].gl_Position.y };
        float grad[4];
        uint64_t n = orc_debug_raster(ndc, R, 0, nullptr, nullptr, grad);
        if (!n) continue;
        xy.resize(2 * n); l12.resize(2 * n);
        orc_debug_raster(ndc, R, n, xy.data(), l12.data(), grad);
        // level of detail from the screen-space derivatives of UV (affine over the triangle)
        const float du1 = E[1].UV.x - E[0].UV.x, du2 = E[2].UV.x - E[0].UV.x, dv1 = E[1].UV.y - E[0].UV.y, dv2 = E[2].UV.y - E[0].UV.y;
        const float dudx = grad[0] * du1 + grad[1] * du2, dvdx = grad[0] * dv1 + grad[1] * dv2;
        const float dudy = grad[2] * du1 + grad[3] * du2, dvdy = grad[2] * dv1 + grad[3] * dv2;
        for (int u = 0; u < 8; ++u)
            g_unit_lambda[u] = g_unit_tex[u] ? orc_debug_lod(g_unit_tex[u]->w, g_unit_tex[u]->h, dudx, dvdx, dudy, dvdy) : 0.0f;
        for (uint64_t f = 0; f < n; ++f) {
            const float l1 = l12[2 * f], l2 = l12[2 * f + 1];
#define LERP(field, k) ((E[0].field[k] + l1 * (E[1].field[k] - E[0].field[k])) + l2 * (E[2].field[k] - E[0].field[k]))
            ref_fs::Position = glm::vec3(LERP(Position, 0), LERP(Position, 1), LERP(Position, 2));
            ref_fs::Normal = glm::vec3(LERP(Normal, 0), LERP(Normal, 1), LERP(Normal, 2));
            ref_fs::Tangent = glm::vec4(LERP(Tangent, 0), LERP(Tangent, 1), LERP(Tangent, 2), LERP(Tangent, 3));
            ref_fs::UV = glm::vec2(LERP(UV, 0), LERP(UV, 1));
#undef LERP
            ref_fs::Scale = E[2].Scale;             // flat: provoking vertex = last
            ref_fs::Quaternion = E[2].Quaternion;
            // an index beyond the SSBO would be an out-of-bounds write on a GPU; the shader's own cap test comes first
            if ((size_t)ref_fs::g_validCounter.v < (size_t)ref_fs::u_maxGaussians && (size_t)ref_fs::g_validCounter.v >= ssbo_records) {
                fprintf(stderr, "swgl: SSBO overflow (cap %d, capacity %zu)\n", ref_fs::u_maxGaussians, ssbo_records);
                exit(72);
            }
            ref_fs::main_();
        }
    }
    std::memcpy(counter.data(), &ref_fs::g_validCounter.v, 4);
}
}  // extern "C"

int main(int argc, char** argv) {
    if (argc != 4 && argc != 7) {
        fprintf(stderr, "usage: ref_pipeline_check in.glb R out_records.bin [out.ply format gaussianStd]\n");
        return 64;
    }
    glsl_env::g_texture = fetch_texel;
    RenderContext rc;
    // the two buffers Renderer::initialize creates for this pass (renderer.cpp:48-50,75-77)
    glGenBuffers(1, &rc.gaussianBuffer);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianBuffer);
    glBufferData(GL_SHADER_STORAGE_BUFFER, 0, nullptr, GL_DYNAMIC_DRAW);
    glGenBuffers(1, &rc.atomicCounterBufferConversionPass);
    glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, rc.atomicCounterBufferConversionPass);
    glBufferData(GL_ATOMIC_COUNTER_BUFFER, sizeof(uint32_t), nullptr, GL_DYNAMIC_DRAW);
    rc.resolutionTarget = (unsigned)atoi(argv[2]);
    int rcode = 0;
    {
        SceneManager sm(rc);
        if (!sm.loadModel(argv[1], "")) return 2;
        ConversionPass pass;
        const auto t_exec0 = std::chrono::steady_clock::now();
        pass.execute(rc);
        const double exec_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_exec0).count();

        const uint32_t counter = rc.numberOfGaussians;
        const uint32_t cap = (uint32_t)swgl::uniform_i["u_maxGaussians"];
        const std::vector<uint8_t>& ssbo = swgl::buffers[rc.gaussianBuffer];
        const uint64_t ssbo_bytes = ssbo.size();
        const uint64_t stored = counter < cap ? counter : cap;
        std::ofstream f(std::string(argv[3]) == "-" ? "/dev/null" : argv[3], std::ios::binary);
        f.write(reinterpret_cast<const char*>(&counter), 4);
        f.write(reinterpret_cast<const char*>(&cap), 4);
        f.write(reinterpret_cast<const char*>(&ssbo_bytes), 8);
        f.write(reinterpret_cast<const char*>(ssbo.data()), (std::streamsize)(stored * 96));
        // execute_ms: wall time of the reference's ConversionPass::execute on this CPU (one thread) — the reference-side
        // number bench.py reports as cpu_baseline.kind = "reference"
        fprintf(stdout, "{\"counter\": %u, \"max_gaussians\": %u, \"ssbo_bytes\": %llu, \"draws\": %llu, \"meshes\": %zu, \"execute_ms\": %.3f}\n",
                counter, cap, (unsigned long long)ssbo_bytes, (unsigned long long)swgl::n_draws, rc.dataMeshAndGlMesh.size(), exec_ms);

        if (argc == 7) {
            const unsigned fmt = (unsigned)atoi(argv[5]);
            rc.gaussianStd = (float)atof(argv[6]);
            std::remove(argv[4]);
            sm.exportPly(argv[4], fmt);   // writes from a detached thread (SceneManager.cpp:671-676): wait for the file
            const uint64_t row = fmt == 1 ? 76 : fmt == 2 ? 48 : 248;
            rcode = 5;
            for (int i = 0; i < 3000 && rcode; ++i) {
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
                std::ifstream p(argv[4], std::ios::binary | std::ios::ate);
                if (!p) continue;
                const uint64_t sz = (uint64_t)p.tellg();
                if (sz > (uint64_t)counter * row) {   // header + all rows; let the writer close the stream
                    std::this_thread::sleep_for(std::chrono::milliseconds(100));
                    rcode = 0;
                }
            }
        }
    }
    return rcode;
}
