// ConversionPassHip.cpp — alternative body of ConversionPass::execute for electronicarts/mesh2splat: the conversion runs on
// an AMD MI355X through libm2s_hip.so (include/m2s.h) instead of the OpenGL VS/GS/rasteriser/FS draw.
//
// Drop-in: it replaces src/renderer/renderPasses/ConversionPass.cpp in the build (same class, same header, same caller:
// Renderer::renderFrame, renderer.cpp:152-160) and needs NO other change to the reference — everything it reads is what
// RenderContext already holds (RenderPass.hpp:11-29, RenderContext.hpp:64-90), and what it leaves behind is what the reference's
// consumers read: renderContext.numberOfGaussians and the records in renderContext.gaussianBuffer, so that the reference's
// own SceneManager::exportPly (SceneManager.cpp:651-678) and viewer passes work on them unmodified.
//
// This very file is compiled against the reference's sources and run by oracle/_ref/ref_dropin_check (oracle/Makefile);
// INTEGRATION.md quotes it.
#include "ConversionPass.hpp"

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "m2s.h"   // this repository's include/m2s.h ; link with -lm2s_hip

namespace {
m2s_ctx* g_ctx = nullptr;
uint64_t g_uploaded_signature = 0;      // of the model the device-resident scene was built from (0: none)

// What SceneManager::loadModel leaves in RenderContext::dataMeshAndGlMesh, reduced to 64 bits: per mesh its name, face count,
// cumulative bounding box, base colour, and the VBO SceneManager::setupMeshBuffers generated for it.  (NOT the vector's
// data() pointer: loadModel does clear() + reserve(n) + push_back, SceneManager.cpp:471-573, which keeps the allocation when
// the next model has as many meshes or fewer — a second one-mesh model would pass for the first.)
uint64_t model_signature(const RenderContext& rc) {
    uint64_t h = 1469598103934665603ull;                           // FNV-1a
    auto mix = [&h](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    const uint64_t n = rc.dataMeshAndGlMesh.size();
    mix(&n, sizeof n);
    for (const auto& meshAndGl : rc.dataMeshAndGlMesh) {
        const utils::Mesh& mesh = meshAndGl.first;
        const uint64_t faces = mesh.faces.size(), vertices = meshAndGl.second.vertexCount;
        mix(mesh.name.data(), mesh.name.size());
        mix(&faces, sizeof faces);
        mix(&vertices, sizeof vertices);
        mix(&meshAndGl.second.vbo, sizeof meshAndGl.second.vbo);
        mix(&mesh.bbox.min, 12);
        mix(&mesh.bbox.max, 12);
        mix(&mesh.material.baseColorFactor, 16);
        if (faces) { mix(&mesh.faces.front(), sizeof(utils::Face)); mix(&mesh.faces.back(), sizeof(utils::Face)); }
    }
    return h ? h : 1;
}

void check(m2s_status s) {
    if (s != M2S_OK) throw std::runtime_error(m2s_last_error(g_ctx));
}
}  // namespace

void ConversionPass::execute(RenderContext& renderContext)
{
    if (!g_ctx) check(m2s_create(/*device*/ 0, &g_ctx));

    // ---- scene -> device, once per loaded model (the reference uploads its VBOs / textures in SceneManager::loadModel) ----
    const uint64_t signature = model_signature(renderContext);
    if (signature != g_uploaded_signature) {
        std::vector<std::vector<float>> vertices(renderContext.dataMeshAndGlMesh.size());
        std::vector<std::vector<unsigned char>> rgba;           // textures the reference keeps with 3 channels, expanded
        rgba.reserve(3 * renderContext.dataMeshAndGlMesh.size());
        std::vector<m2s_mesh> meshes;
        size_t k = 0;
        for (auto& meshAndGl : renderContext.dataMeshAndGlMesh) {
            const utils::Mesh& mesh = meshAndGl.first;
            // the de-indexed 17-float vertex stream SceneManager::setupMeshBuffers builds for glBufferData
            // (SceneManager.cpp:483-512), rebuilt from Mesh::faces (utils.hpp:155-182)
            std::vector<float>& v = vertices[k++];
            v.reserve(mesh.faces.size() * 3 * 17);
            for (const utils::Face& face : mesh.faces)
                for (int i = 0; i < 3; ++i) {
                    const float row[17] = { face.pos[i].x, face.pos[i].y, face.pos[i].z,
                                            face.normal[i].x, face.normal[i].y, face.normal[i].z,
                                            face.tangent[i].x, face.tangent[i].y, face.tangent[i].z, face.tangent[i].w,
                                            face.uv[i].x, face.uv[i].y,
                                            face.normalizedUvs[i].x, face.normalizedUvs[i].y,
                                            face.scale.x, face.scale.y, face.scale.z };
                    v.insert(v.end(), row, row + 17);
                }
            m2s_mesh m;
            std::memset(&m, 0, sizeof m);
            m.vertices = v.data();
            m.n_vertices = (uint32_t)(mesh.faces.size() * 3);
            m.stride_floats = 17;
            std::memcpy(m.bbox_min, &mesh.bbox.min, 12);          // the cumulative bbox, as setupMeshBuffers computed it
            std::memcpy(m.bbox_max, &mesh.bbox.max, 12);
            std::memcpy(m.base_color, &mesh.material.baseColorFactor, 16);
            auto it = renderContext.meshToTextureData.find(mesh.name);
            if (it != renderContext.meshToTextureData.end()) {
                const char* keys[3] = { BASE_COLOR_TEXTURE, NORMAL_TEXTURE, METALLIC_ROUGHNESS_TEXTURE };   // texture units 0, 1, 2
                for (int t = 0; t < 3; ++t) {
                    auto tex = it->second.find(keys[t]);
                    if (tex == it->second.end() || tex->second.textureData.empty() || !tex->second.width || !tex->second.height) continue;
                    const utils::TextureDataGl& td = tex->second;
                    const unsigned char* texels = td.textureData.data();
                    if (td.channels != 4) {                       // glUtils::generateTextures uploads these as GL_RGB (alpha reads 1)
                        rgba.emplace_back((size_t)td.width * td.height * 4, (unsigned char)255);
                        for (size_t p = 0; p < (size_t)td.width * td.height; ++p)
                            for (unsigned c = 0; c < 3; ++c) rgba.back()[p * 4 + c] = texels[p * 3 + c];
                        texels = rgba.back().data();
                    }
                    m.tex[t].rgba8 = texels;
                    m.tex[t].width = td.width;
                    m.tex[t].height = td.height;
                }
            }
            meshes.push_back(m);
        }
        // the upload ends with what the conversion below needs at THIS resolutionTarget (exact count, pipeline choice, XCD band
        // table, record pool), so that the first conversion of a model costs what a repeated one does
        check(m2s_set_resolution_hint(g_ctx, renderContext.resolutionTarget));
        check(m2s_upload_scene(g_ctx, meshes.data(), (uint32_t)meshes.size()));   // host memory is only borrowed during the call
        g_uploaded_signature = signature;
    }

    // ---- the pass: cap = min(6 R^2 meshes, 7 000 000) as ConversionPass.cpp:21-24 (the library's default policy) ----
    uint64_t total = 0;
    check(m2s_convert(g_ctx, renderContext.resolutionTarget, &total));
    renderContext.numberOfGaussians = (GLint)total;                // the counter, NOT clamped (ConversionPass.cpp:56-59)

    // ---- records -> renderContext.gaussianBuffer, sized as the reference sizes it (ConversionPass.cpp:21-33) ----
    // (a viewer that shares memory with HIP would import m2s_device_records() instead of taking this copy through the host)
    unsigned int meshCount = static_cast<unsigned int>(std::max(size_t(1), renderContext.dataMeshAndGlMesh.size()));
    unsigned int maxGaussians = renderContext.resolutionTarget * renderContext.resolutionTarget * 6 * meshCount;
    maxGaussians = std::min(maxGaussians, static_cast<unsigned int>(MAX_GAUSSIANS_TO_SORT));
    const GLsizeiptr bufferSize = static_cast<GLsizeiptr>(maxGaussians) * sizeof(glm::vec4) * 6;
    GLint currentSize;
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, renderContext.gaussianBuffer);
    glGetBufferParameteriv(GL_SHADER_STORAGE_BUFFER, GL_BUFFER_SIZE, &currentSize);
    if (currentSize != bufferSize) glBufferData(GL_SHADER_STORAGE_BUFFER, bufferSize, nullptr, GL_DYNAMIC_DRAW);
    const uint64_t stored = m2s_num_stored(g_ctx);                 // min(counter, cap)
    std::vector<m2s_gaussian> records(stored);
    if (stored) {
        check(m2s_download(g_ctx, records.data(), stored));
        glBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(stored * sizeof(m2s_gaussian)), records.data());
    }
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, 0);
}
