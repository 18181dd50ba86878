// ref_gl_check.cpp — TEST INFRASTRUCTURE.  The reference's WHOLE conversion path on a REAL OpenGL implementation:
//
//     SceneManager::loadModel -> glUtils::reloadShaderPrograms (the three UNMODIFIED .glsl files under
//     /root/reference/src/shaders/conversion, compiled by the GL's own GLSL compiler) -> ConversionPass::execute
//     (ConversionPass.cpp:9-117: glDrawArrays into the R x R viewport, fragments appended through the atomic counter)
//     -> glGetBufferSubData of the SSBO and the counter
//
// all of it the reference's own objects (oracle/Makefile compiles them from where they lie), on Mesa llvmpipe brought up by
// ref_gl_boot.c.  This is what ref_pipeline_check cannot give: there the fixed-function stages (rasterisation, varying
// interpolation, glGenerateMipmap, LOD selection, trilinear filtering) are the oracle's pinned ones; here they are a GL
// implementation's.  GL leaves their last bits implementation-defined, so the comparison (tests/test_ref_gl.py) reports
// differences instead of demanding bit equality: fragment count, set difference of covered pixels, value deviations.
//
//   ref_gl_check in.glb R out_records.bin [coverage.bin [mips.bin]]
//
// out_records.bin: u32 counter, u32 maxGaussians, u64 SSBO bytes, then min(counter, maxGaussians) records of 96 bytes in
// ARRIVAL order (the atomic counter's).  coverage.bin (optional): a second pass with the reference's VS + GS and a
// three-line fragment shader of ours that appends (mesh, triangle, x, y) per fragment: u32 n, then n x 4 u32.
// mips.bin (optional): what glGenerateMipmap made of the FIRST mesh's base-colour texture (glUtils.cpp:292-313): u32 w, u32 h,
// u32 levels, then levels 0..levels-1 as RGBA8 (glGetTexImage).
#include "utils/SceneManager.hpp"
#include "renderer/renderPasses/ConversionPass.hpp"

#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

extern "C" void* (*ref_gl_boot(const char** why))(const char*);

// The three conversion shaders, byte for byte as they lie under /root/reference/src/shaders/conversion when oracle/Makefile
// runs (generated into oracle/_ref/gen/, which is not part of the repository): the GPU box has no /root/reference, and the
// reference's glUtils::readShaderFile wants files, so they are written to a scratch directory at start-up.
#include "_ref/gen/conversion_shaders.inc"

// ---- entry points: GLEW-style pointers the reference's objects reference, filled from the real GL ----------------------
#define GLFN_LIST(X) \
    X(ACTIVETEXTURE, ActiveTexture) X(ATTACHSHADER, AttachShader) X(BINDBUFFER, BindBuffer) X(BINDBUFFERBASE, BindBufferBase) \
    X(BINDFRAMEBUFFER, BindFramebuffer) X(BINDRENDERBUFFER, BindRenderbuffer) X(BINDVERTEXARRAY, BindVertexArray) X(BUFFERDATA, BufferData) \
    X(BUFFERSUBDATA, BufferSubData) X(CHECKFRAMEBUFFERSTATUS, CheckFramebufferStatus) X(COMPILESHADER, CompileShader) \
    X(CREATEPROGRAM, CreateProgram) X(CREATESHADER, CreateShader) X(DELETEFRAMEBUFFERS, DeleteFramebuffers) X(DELETEPROGRAM, DeleteProgram) \
    X(DELETERENDERBUFFERS, DeleteRenderbuffers) X(DELETESHADER, DeleteShader) X(ENABLEVERTEXATTRIBARRAY, EnableVertexAttribArray) \
    X(FRAMEBUFFERRENDERBUFFER, FramebufferRenderbuffer) X(GENBUFFERS, GenBuffers) X(GENFRAMEBUFFERS, GenFramebuffers) \
    X(GENRENDERBUFFERS, GenRenderbuffers) X(GENVERTEXARRAYS, GenVertexArrays) X(GENERATEMIPMAP, GenerateMipmap) \
    X(GETBUFFERPARAMETERIV, GetBufferParameteriv) X(GETBUFFERSUBDATA, GetBufferSubData) X(GETPROGRAMINFOLOG, GetProgramInfoLog) \
    X(GETPROGRAMIV, GetProgramiv) X(GETSHADERINFOLOG, GetShaderInfoLog) X(GETSHADERIV, GetShaderiv) X(GETUNIFORMLOCATION, GetUniformLocation) \
    X(ISPROGRAM, IsProgram) X(LINKPROGRAM, LinkProgram) X(MAPBUFFERRANGE, MapBufferRange) X(MEMORYBARRIER, MemoryBarrier) \
    X(RENDERBUFFERSTORAGE, RenderbufferStorage) X(SHADERSOURCE, ShaderSource) X(UNIFORM1F, Uniform1f) X(UNIFORM1I, Uniform1i) \
    X(UNIFORM1UI, Uniform1ui) X(UNIFORM1UIV, Uniform1uiv) X(UNIFORM2F, Uniform2f) X(UNIFORM2I, Uniform2i) X(UNIFORM3F, Uniform3f) \
    X(UNIFORM4F, Uniform4f) X(UNIFORMMATRIX4FV, UniformMatrix4fv) X(USEPROGRAM, UseProgram) X(VERTEXATTRIBPOINTER, VertexAttribPointer)
#define X(UP, Name) PFNGL##UP##PROC __glew##Name = nullptr;
GLFN_LIST(X)
#undef X

// GL 1.1 entry points the reference links directly
static void (GLAPIENTRY* p_glBindTexture)(GLenum, GLuint);
static void (GLAPIENTRY* p_glDeleteTextures)(GLsizei, const GLuint*);
static void (GLAPIENTRY* p_glDisable)(GLenum);
static void (GLAPIENTRY* p_glDrawArrays)(GLenum, GLint, GLsizei);
static void (GLAPIENTRY* p_glEnable)(GLenum);
static void (GLAPIENTRY* p_glFinish)(void);
static void (GLAPIENTRY* p_glGenTextures)(GLsizei, GLuint*);
static void (GLAPIENTRY* p_glTexImage2D)(GLenum, GLint, GLint, GLsizei, GLsizei, GLint, GLenum, GLenum, const void*);
static void (GLAPIENTRY* p_glTexParameteri)(GLenum, GLenum, GLint);
static void (GLAPIENTRY* p_glViewport)(GLint, GLint, GLsizei, GLsizei);
static GLenum (GLAPIENTRY* p_glGetError)(void);
static const GLubyte* (GLAPIENTRY* p_glGetString)(GLenum);
extern "C" {
void GLAPIENTRY glBindTexture(GLenum t, GLuint id) { p_glBindTexture(t, id); }
void GLAPIENTRY glDeleteTextures(GLsizei n, const GLuint* ids) { p_glDeleteTextures(n, ids); }
void GLAPIENTRY glDisable(GLenum c) { p_glDisable(c); }
void GLAPIENTRY glDrawArrays(GLenum m, GLint f, GLsizei c) { p_glDrawArrays(m, f, c); }
void GLAPIENTRY glEnable(GLenum c) { p_glEnable(c); }
void GLAPIENTRY glFinish(void) { p_glFinish(); }
void GLAPIENTRY glGenTextures(GLsizei n, GLuint* ids) { p_glGenTextures(n, ids); }
void GLAPIENTRY glTexImage2D(GLenum t, GLint l, GLint i, GLsizei w, GLsizei h, GLint b, GLenum f, GLenum ty, const void* d) { p_glTexImage2D(t, l, i, w, h, b, f, ty, d); }
void GLAPIENTRY glTexParameteri(GLenum t, GLenum p, GLint v) { p_glTexParameteri(t, p, v); }
void GLAPIENTRY glViewport(GLint x, GLint y, GLsizei w, GLsizei h) { p_glViewport(x, y, w, h); }
GLenum GLAPIENTRY glGetError(void) { return p_glGetError(); }
}

static const char* kCoverageFS = R"(#version 460 core
layout(std430, binding = 2) buffer Cov { uvec4 frag[]; } cov;
layout(binding = 3) uniform atomic_uint g_covCounter;
uniform uint u_mesh;
uniform uint u_tri;
void main() { uint i = atomicCounterIncrement(g_covCounter); cov.frag[i] = uvec4(u_mesh, u_tri, uint(gl_FragCoord.x), uint(gl_FragCoord.y)); }
)";

int main(int argc, char** argv) {
    if (argc < 4 || argc > 6) { fprintf(stderr, "usage: ref_gl_check in.glb R out_records.bin [coverage.bin [mips.bin]]\n"); return 64; }
    const char* why = "";
    void* (*gpa)(const char*) = ref_gl_boot(&why);
    if (!gpa) { fprintf(stderr, "no GL: %s\n", why); return 3; }
#define X(UP, Name) __glew##Name = reinterpret_cast<PFNGL##UP##PROC>(gpa("gl" #Name)); if (!__glew##Name) { fprintf(stderr, "GL lacks gl" #Name "\n"); return 3; }
    GLFN_LIST(X)
#undef X
#define L(name) p_##name = reinterpret_cast<decltype(p_##name)>(gpa(#name)); if (!p_##name) { fprintf(stderr, "GL lacks " #name "\n"); return 3; }
    L(glBindTexture) L(glDeleteTextures) L(glDisable) L(glDrawArrays) L(glEnable) L(glFinish) L(glGenTextures) L(glTexImage2D) L(glTexParameteri)
    L(glViewport) L(glGetError) L(glGetString)
#undef L
    char scratch[] = "/tmp/m2s_ref_gl_XXXXXX";
    if (!mkdtemp(scratch)) { perror("mkdtemp"); return 3; }
    const std::string shader_dir = std::string(scratch) + "/";
    {
        const struct { const char* name; const unsigned char* data; unsigned len; } files[3] = {
            { "converterVS.glsl", converterVS_glsl, converterVS_glsl_len }, { "converterGS.glsl", converterGS_glsl, converterGS_glsl_len },
            { "converterFS.glsl", converterFS_glsl, converterFS_glsl_len } };
        for (const auto& f : files) {
            std::ofstream o(shader_dir + f.name, std::ios::binary);
            o.write(reinterpret_cast<const char*>(f.data), f.len);
        }
    }
    int rcode = 0;
    {
        RenderContext rc;
        // the two buffers Renderer::initialize creates for this pass (renderer.cpp:48-50,75-77)
        glGenBuffers(1, &rc.gaussianBuffer);
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianBuffer);
        glBufferData(GL_SHADER_STORAGE_BUFFER, 0, nullptr, GL_DYNAMIC_DRAW);
        glGenBuffers(1, &rc.atomicCounterBufferConversionPass);
        glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, rc.atomicCounterBufferConversionPass);
        glBufferData(GL_ATOMIC_COUNTER_BUFFER, sizeof(uint32_t), nullptr, GL_DYNAMIC_DRAW);
        rc.resolutionTarget = (unsigned)atoi(argv[2]);
        // the converter program, exactly as glUtils::initializeShaderFileMonitoring registers it (glUtils.cpp:117-121)
        rc.shaderRegistry.registerShaderProgram(glUtils::ShaderProgramTypes::ConverterProgram,
                                                { { shader_dir + "converterVS.glsl", GL_VERTEX_SHADER }, { shader_dir + "converterGS.glsl", GL_GEOMETRY_SHADER },
                                                  { shader_dir + "converterFS.glsl", GL_FRAGMENT_SHADER } });
        rc.shaderRegistry.reloadModifiedShaders(true);
        const GLuint prog = rc.shaderRegistry.getProgramID(glUtils::ShaderProgramTypes::ConverterProgram);
        if (!prog || !glIsProgram(prog)) { fprintf(stderr, "the reference's conversion shaders did not compile / link on this GL\n"); return 4; }

        SceneManager sm(rc);
        if (!sm.loadModel(argv[1], "")) return 2;
        ConversionPass pass;
        const auto t0 = std::chrono::steady_clock::now();
        pass.execute(rc);
        const double exec_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const GLenum gl_err = glGetError();

        const uint32_t counter = rc.numberOfGaussians;
        const unsigned meshCount = (unsigned)std::max<size_t>(1, rc.dataMeshAndGlMesh.size());
        uint32_t cap = rc.resolutionTarget * rc.resolutionTarget * 6u * meshCount;      // ConversionPass.cpp:21-24
        cap = std::min<uint32_t>(cap, MAX_GAUSSIANS_TO_SORT);
        GLint ssbo_bytes = 0;
        glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianBuffer);
        glGetBufferParameteriv(GL_SHADER_STORAGE_BUFFER, GL_BUFFER_SIZE, &ssbo_bytes);
        const uint64_t stored = counter < cap ? counter : cap;
        std::vector<uint8_t> rec((size_t)stored * 96);
        if (stored) glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)rec.size(), rec.data());
        const uint64_t sb = (uint64_t)ssbo_bytes;
        std::ofstream f(std::string(argv[3]) == "-" ? "/dev/null" : argv[3], std::ios::binary);
        f.write(reinterpret_cast<const char*>(&counter), 4);
        f.write(reinterpret_cast<const char*>(&cap), 4);
        f.write(reinterpret_cast<const char*>(&sb), 8);
        f.write(reinterpret_cast<const char*>(rec.data()), (std::streamsize)rec.size());

        uint64_t cov_n = 0;
        if (argc == 6 && !rc.dataMeshAndGlMesh.empty()) {
            auto it = rc.meshToTextureData.find(rc.dataMeshAndGlMesh[0].first.name);
            if (it != rc.meshToTextureData.end() && it->second.count(BASE_COLOR_TEXTURE)) {
                auto glGetTexImage_ = reinterpret_cast<void (GLAPIENTRY*)(GLenum, GLint, GLenum, GLenum, void*)>(gpa("glGetTexImage"));
                auto glGetTexLevelParameteriv_ = reinterpret_cast<void (GLAPIENTRY*)(GLenum, GLint, GLenum, GLint*)>(gpa("glGetTexLevelParameteriv"));
                const auto& td = it->second.at(BASE_COLOR_TEXTURE);
                glBindTexture(GL_TEXTURE_2D, td.glTextureID);
                GLint w = 0, h = 0;
                glGetTexLevelParameteriv_(GL_TEXTURE_2D, 0, GL_TEXTURE_WIDTH, &w);
                glGetTexLevelParameteriv_(GL_TEXTURE_2D, 0, GL_TEXTURE_HEIGHT, &h);
                std::ofstream mf(argv[5], std::ios::binary);
                uint32_t hdr[3] = { (uint32_t)w, (uint32_t)h, 0 };
                std::vector<std::vector<uint8_t>> lv;
                for (int l = 0; l <= 4; ++l) {
                    GLint lw = 0, lh = 0;
                    glGetTexLevelParameteriv_(GL_TEXTURE_2D, l, GL_TEXTURE_WIDTH, &lw);
                    glGetTexLevelParameteriv_(GL_TEXTURE_2D, l, GL_TEXTURE_HEIGHT, &lh);
                    if (lw <= 0 || lh <= 0) break;
                    lv.emplace_back((size_t)lw * lh * 4);
                    glGetTexImage_(GL_TEXTURE_2D, l, GL_RGBA, GL_UNSIGNED_BYTE, lv.back().data());
                }
                hdr[2] = (uint32_t)lv.size();
                mf.write(reinterpret_cast<const char*>(hdr), 12);
                for (auto& v : lv) mf.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)v.size());
            }
        }
        if (argc >= 5) {
            // ---- coverage: the reference's VS + GS, our three-line FS; one draw per triangle so that the FS knows it ----
            auto read_file = [](const std::string& p) { std::ifstream s(p); return std::string((std::istreambuf_iterator<char>(s)), std::istreambuf_iterator<char>()); };
            auto compile = [&](GLenum type, const std::string& src) {
                const GLuint sh = glCreateShader(type);
                const char* c = src.c_str();
                glShaderSource(sh, 1, &c, nullptr);
                glCompileShader(sh);
                GLint ok = 0;
                glGetShaderiv(sh, GL_COMPILE_STATUS, &ok);
                if (!ok) { char log[1024]; glGetShaderInfoLog(sh, sizeof log, nullptr, log); fprintf(stderr, "coverage shader: %s\n", log); exit(5); }
                return sh;
            };
            const GLuint cp = glCreateProgram();
            glAttachShader(cp, compile(GL_VERTEX_SHADER, read_file(shader_dir + "converterVS.glsl")));
            glAttachShader(cp, compile(GL_GEOMETRY_SHADER, read_file(shader_dir + "converterGS.glsl")));
            glAttachShader(cp, compile(GL_FRAGMENT_SHADER, kCoverageFS));
            glLinkProgram(cp);
            GLint ok = 0;
            glGetProgramiv(cp, GL_LINK_STATUS, &ok);
            if (!ok) { char log[1024]; glGetProgramInfoLog(cp, sizeof log, nullptr, log); fprintf(stderr, "coverage program: %s\n", log); return 5; }
            GLuint cov_buf = 0, cov_cnt = 0, fbo = 0;
            const uint32_t R = rc.resolutionTarget;
            const size_t cov_cap = (size_t)counter + 4096;
            glGenBuffers(1, &cov_buf);
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, cov_buf);
            glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(cov_cap * 16), nullptr, GL_DYNAMIC_DRAW);
            glGenBuffers(1, &cov_cnt);
            glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, cov_cnt);
            const uint32_t zero = 0;
            glBufferData(GL_ATOMIC_COUNTER_BUFFER, 4, &zero, GL_DYNAMIC_DRAW);
            const GLuint rb = glUtils::setupFrameBuffer(fbo, R, R);      // the reference's own FBO (glUtils.cpp:410-433)
            glBindFramebuffer(GL_FRAMEBUFFER, fbo);
            glBindBufferBase(GL_SHADER_STORAGE_BUFFER, 2, cov_buf);
            glBindBufferBase(GL_ATOMIC_COUNTER_BUFFER, 3, cov_cnt);
            glViewport(0, 0, (GLsizei)R, (GLsizei)R);
            glDisable(GL_DEPTH_TEST); glEnable(GL_BLEND); glDisable(GL_CULL_FACE);          // ConversionPass.cpp:45-48
            glUseProgram(cp);
            uint32_t mi = 0;
            for (auto& mesh : rc.dataMeshAndGlMesh) {
                glUtils::setUniform3f(cp, "u_bboxMin", mesh.first.bbox.min);
                glUtils::setUniform3f(cp, "u_bboxMax", mesh.first.bbox.max);
                glUtils::setUniform1ui(cp, "u_mesh", mi);
                glBindVertexArray(mesh.second.vao);
                for (uint32_t t = 0; 3 * t + 2 < (uint32_t)mesh.second.vertexCount; ++t) {
                    glUtils::setUniform1ui(cp, "u_tri", t);
                    glDrawArrays(GL_TRIANGLES, (GLint)(3 * t), 3);
                }
                ++mi;
            }
            glFinish();
            uint32_t n = 0;
            glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, cov_cnt);
            glGetBufferSubData(GL_ATOMIC_COUNTER_BUFFER, 0, 4, &n);
            cov_n = n;
            std::vector<uint32_t> cov((size_t)std::min<uint64_t>(n, cov_cap) * 4);
            glBindBuffer(GL_SHADER_STORAGE_BUFFER, cov_buf);
            if (!cov.empty()) glGetBufferSubData(GL_SHADER_STORAGE_BUFFER, 0, (GLsizeiptr)(cov.size() * 4), cov.data());
            std::ofstream cf(argv[4], std::ios::binary);
            cf.write(reinterpret_cast<const char*>(&n), 4);
            cf.write(reinterpret_cast<const char*>(cov.data()), (std::streamsize)(cov.size() * 4));
            glBindFramebuffer(GL_FRAMEBUFFER, 0);
            glDeleteRenderbuffers(1, &rb);
            glDeleteFramebuffers(1, &fbo);
        }
        fprintf(stdout, "{\"gl_version\": \"%s\", \"gl_renderer\": \"%s\", \"counter\": %u, \"max_gaussians\": %u, \"ssbo_bytes\": %llu, \"meshes\": %zu, "
                        "\"execute_ms\": %.3f, \"gl_error\": %u, \"coverage_fragments\": %llu}\n",
                (const char*)p_glGetString(GL_VERSION), (const char*)p_glGetString(GL_RENDERER), counter, cap, (unsigned long long)sb,
                rc.dataMeshAndGlMesh.size(), exec_ms, (unsigned)gl_err, (unsigned long long)cov_n);
        fflush(stdout);
    }
    for (const char* n : { "converterVS.glsl", "converterGS.glsl", "converterFS.glsl" }) std::remove((shader_dir + n).c_str());
    rmdir(scratch);
    _Exit(rcode);   // (no GL teardown: the process ends here)
}
