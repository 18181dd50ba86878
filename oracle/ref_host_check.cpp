// ref_host_check.cpp — TEST INFRASTRUCTURE.  Runs the REFERENCE's own host code (compiled from where it
// lies under /root/reference by oracle/Makefile, never copied here) so that the tests can pin this
// repository's loader and PLY writers/readers against it:
//
//   ref_host_check scene    in.glb  out.bin           SceneManager::loadModel (SceneManager.cpp:22-35):
//                                                     parseGltfFile -> setupMeshBuffers -> loadTextures
//   ref_host_check plywrite rec.bin out.ply fmt mult  parsers::savePlyVector (parsers.cpp:631-651)
//   ref_host_check plyread  in.ply  out.bin           parsers::loadPlyFile   (parsers.cpp:516-629)
//
// The reference uploads its vertex vector with glBufferData and its textures with glTexImage2D; there is
// no GL on this machine, so the nine GLEW entry points that translation unit touches are defined HERE as
// recording stubs (the glBufferData stub keeps a copy of what the reference would have sent to the GPU:
// that copy is the loader's real output), and glUtils::generateTextures (pure GL upload) is an empty body.
// Nothing in this file is product code and nothing here restates reference logic.
#include "utils/SceneManager.hpp"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>

// ---- GL stand-ins -------------------------------------------------------------------------------------
static std::vector<std::vector<float>> g_uploads;   // one per glBufferData call = one per mesh
static GLuint g_next_id = 1;

static void GLAPIENTRY stub_gen(GLsizei n, GLuint* ids) { for (GLsizei i = 0; i < n; ++i) ids[i] = g_next_id++; }
static void GLAPIENTRY stub_bind_buffer(GLenum, GLuint) {}
static void GLAPIENTRY stub_bind_vao(GLuint) {}
static void GLAPIENTRY stub_buffer_data(GLenum, GLsizeiptr size, const void* data, GLenum) {
    const float* f = static_cast<const float*>(data);
    g_uploads.emplace_back(f, f + size / sizeof(float));
}
static void GLAPIENTRY stub_enable_attrib(GLuint) {}
static void GLAPIENTRY stub_attrib_pointer(GLuint, GLint, GLenum, GLboolean, GLsizei, const void*) {}
static void GLAPIENTRY stub_delete_program(GLuint) {}
static GLboolean GLAPIENTRY stub_is_program(GLuint) { return GL_FALSE; }

PFNGLGENVERTEXARRAYSPROC __glewGenVertexArrays = stub_gen;
PFNGLGENBUFFERSPROC __glewGenBuffers = stub_gen;
PFNGLBINDBUFFERPROC __glewBindBuffer = stub_bind_buffer;
PFNGLBINDVERTEXARRAYPROC __glewBindVertexArray = stub_bind_vao;
PFNGLBUFFERDATAPROC __glewBufferData = stub_buffer_data;
PFNGLENABLEVERTEXATTRIBARRAYPROC __glewEnableVertexAttribArray = stub_enable_attrib;
PFNGLVERTEXATTRIBPOINTERPROC __glewVertexAttribPointer = stub_attrib_pointer;
PFNGLDELETEPROGRAMPROC __glewDeleteProgram = stub_delete_program;
PFNGLISPROGRAMPROC __glewIsProgram = stub_is_program;

namespace glUtils {
void generateTextures(std::map<std::string, std::map<std::string, utils::TextureDataGl>>&) {}
}

// ---- little-endian dump helpers -------------------------------------------------------------------------
static void put(std::ofstream& f, const void* p, size_t n) { f.write(static_cast<const char*>(p), (std::streamsize)n); }
static void put_u32(std::ofstream& f, uint32_t v) { put(f, &v, 4); }
static void put_str(std::ofstream& f, const std::string& s) { put_u32(f, (uint32_t)s.size()); put(f, s.data(), s.size()); }

static int do_scene(const char* in, const char* out) {
    RenderContext rc;
    {
        SceneManager sm(rc);
        if (!sm.loadModel(in, "")) return 2;
        std::ofstream f(out, std::ios::binary);
        const auto& ms = rc.dataMeshAndGlMesh;
        if (g_uploads.size() != ms.size()) { fprintf(stderr, "upload count mismatch\n"); return 3; }
        put_u32(f, (uint32_t)ms.size());
        static const char* keys[3] = { BASE_COLOR_TEXTURE, NORMAL_TEXTURE, METALLIC_ROUGHNESS_TEXTURE };
        for (size_t i = 0; i < ms.size(); ++i) {
            const utils::Mesh& m = ms[i].first;
            put_str(f, m.name);
            put_u32(f, (uint32_t)ms[i].second.vertexCount);
            put_u32(f, (uint32_t)g_uploads[i].size());
            put(f, g_uploads[i].data(), g_uploads[i].size() * sizeof(float));
            put(f, &m.bbox.min, 12);
            put(f, &m.bbox.max, 12);
            put(f, &m.material.baseColorFactor, 16);
            auto it = rc.meshToTextureData.find(m.name);
            for (int k = 0; k < 3; ++k) {
                const utils::TextureDataGl* t = nullptr;
                if (it != rc.meshToTextureData.end()) {
                    auto jt = it->second.find(keys[k]);
                    if (jt != it->second.end()) t = &jt->second;
                }
                if (!t) { put_u32(f, 0); put_u32(f, 0); put_u32(f, 0); put_u32(f, 0); continue; }
                put_u32(f, t->width); put_u32(f, t->height); put_u32(f, t->channels); put_u32(f, (uint32_t)t->textureData.size());
                put(f, t->textureData.data(), t->textureData.size());
            }
        }
    }
    return 0;
}

static bool read_records(const char* path, std::vector<utils::GaussianDataSSBO>& g) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    f.seekg(0);
    g.resize((size_t)n / sizeof(utils::GaussianDataSSBO));
    f.read(reinterpret_cast<char*>(g.data()), (std::streamsize)(g.size() * sizeof(utils::GaussianDataSSBO)));
    return true;
}

int main(int argc, char** argv) {
    static_assert(sizeof(utils::GaussianDataSSBO) == 96, "record layout");
    const std::string mode = argc > 1 ? argv[1] : "";
    if (mode == "scene" && argc == 4) return do_scene(argv[2], argv[3]);
    if (mode == "plywrite" && argc == 6) {
        std::vector<utils::GaussianDataSSBO> g;
        if (!read_records(argv[2], g)) return 2;
        parsers::savePlyVector(argv[3], g, (unsigned)atoi(argv[4]), (float)atof(argv[5]));
        return 0;
    }
    if (mode == "plyread" && argc == 4) {
        std::vector<utils::GaussianDataSSBO> g;
        bool pbr = false;
        parsers::loadPlyFile(argv[2], g, pbr);
        std::ofstream f(argv[3], std::ios::binary);
        put_u32(f, pbr ? 1u : 0u);
        put_u32(f, (uint32_t)g.size());
        put(f, g.data(), g.size() * sizeof(utils::GaussianDataSSBO));
        return 0;
    }
    fprintf(stderr, "usage: ref_host_check scene in.glb out.bin | plywrite rec.bin out.ply fmt mult | plyread in.ply out.bin\n");
    return 64;
}
