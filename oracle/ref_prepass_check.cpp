// ref_prepass_check.cpp — TEST INFRASTRUCTURE.  Runs the reference's viewer prepass on the CPU:
//
//     GaussiansPrepass::execute               (GaussiansPrepass.cpp:8-56: uniforms, buffer bindings, dispatch shape)
//      -> gaussianSplattingPrepassCS.glsl     (+ common.glsl; the reference's shader, as C++ through glm: glsl2cpp.py)
//
// both compiled from where they lie under /root/reference (oracle/Makefile).  What is NOT the reference's is OpenGL,
// which does not exist on this machine: this file implements the ~20 GL entry points the pass touches.  Uniforms and
// buffer bindings are bookkeeping; glDispatchCompute runs main() once per invocation, in increasing
// gl_GlobalInvocationID order (GL leaves the order open; input order is the one the oracle and the product fix);
// texture() on the depth texture is GL_NEAREST / CLAMP_TO_EDGE as renderer.cpp:290-296 sets it.
//
//   ref_prepass_check in.bin out.bin
//
// in.bin : "M2SP", u32 n, 3 x mat4 (worldToView, viewToClip, modelToWorld; column-major), i32 resolution[2],
//          f32 near, far, gaussianStd, u32 resolutionTarget, i32 renderMode, u32 format, plyHasPbr, depthTestMesh,
//          u32 depth_w, depth_h, n x 24 f32 records (GaussianVertex), depth_w*depth_h f32 (row 0 = bottom)
// out.bin: u32 counter, counter x 24 f32 (QuadNdcTransformation), counter x f32 (depths_vs)
#include "renderer/renderPasses/GaussiansPrepass.hpp"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

// =====================================================================================================
// the shader
// =====================================================================================================
namespace ref_cs {
using namespace glm;
struct sampler2D { int unit; };
struct atomic_uint { unsigned v; };
static inline uint atomicCounterIncrement(atomic_uint& c) { return c.v++; }
static vec4 (*g_texture)(int unit, vec2 uv) = nullptr;
static inline vec4 texture(const sampler2D& s, vec2 uv) { return g_texture(s.unit, uv); }
// GLSL converts integer arguments implicitly (clamp(x, 0, 1)); C++ template deduction does not
static inline float clamp(float x, int lo, int hi) { return glm::clamp(x, float(lo), float(hi)); }
using glm::clamp;
// uvec3 whose two-component swizzles are already vec2: GLSL converts uvec2 -> vec2 implicitly at both uses
// (random2d(vec2) argument, uvec2 * float)
struct InvocationID {
    uint x, y, z;
    vec2 xy() const { return vec2(float(x), float(y)); }
    vec2 yx() const { return vec2(float(y), float(x)); }
};
static InvocationID gl_GlobalInvocationID;
static uvec3 gl_NumWorkGroups;
static const uvec3 gl_WorkGroupSize(16, 16, 1);     // layout(local_size_x = 16, local_size_y = 16), :57
#include "_ref/gen/gaussianSplattingPrepassCS.inc"
}  // namespace ref_cs

// =====================================================================================================
// minimal software GL
// =====================================================================================================
namespace swgl {
static GLuint next_id = 1;
static std::map<GLuint, std::vector<uint8_t>> buffers;
static std::map<GLenum, GLuint> bound_buffer;
static std::map<std::pair<GLenum, GLuint>, GLuint> indexed_binding;
struct Texture { int w = 0, h = 0; std::vector<float> depth; std::map<GLenum, GLint> params; };
static std::map<GLuint, Texture> textures;
static GLuint active_unit = 0;
static std::map<GLuint, GLuint> unit_binding;
static std::map<std::string, GLint> uniform_location;
static std::map<GLint, std::string> uniform_name;
static std::map<std::string, std::vector<float>> uniform_f;
static std::map<std::string, long long> uniform_i;
static uint64_t n_dispatches = 0;

static void gen(GLsizei n, GLuint* ids) { for (GLsizei i = 0; i < n; ++i) ids[i] = next_id++; }
static void GLAPIENTRY GenBuffers(GLsizei n, GLuint* ids) { gen(n, ids); }
static void GLAPIENTRY BindBuffer(GLenum target, GLuint id) { bound_buffer[target] = id; }
static void GLAPIENTRY BindBufferBase(GLenum target, GLuint index, GLuint id) { indexed_binding[{ target, index }] = id; bound_buffer[target] = id; }
static void GLAPIENTRY BufferData(GLenum target, GLsizeiptr size, const void* data, GLenum) {
    std::vector<uint8_t>& b = buffers[bound_buffer[target]];
    b.assign((size_t)size, 0);
    if (data) memcpy(b.data(), data, (size_t)size);
}
static void GLAPIENTRY BufferSubData(GLenum target, GLintptr off, GLsizeiptr size, const void* data) {
    std::vector<uint8_t>& b = buffers[bound_buffer[target]];
    if ((size_t)(off + size) <= b.size()) memcpy(b.data() + off, data, (size_t)size);
}
static void GLAPIENTRY UseProgram(GLuint) {}
static void GLAPIENTRY DeleteProgram(GLuint) {}
static GLboolean GLAPIENTRY IsProgram(GLuint) { return GL_FALSE; }
static GLint GLAPIENTRY GetUniformLocation(GLuint, const GLchar* name) {
    auto it = uniform_location.find(name);
    if (it != uniform_location.end()) return it->second;
    const GLint loc = (GLint)uniform_location.size();
    uniform_location[name] = loc;
    uniform_name[loc] = name;
    return loc;
}
static void GLAPIENTRY Uniform1f(GLint loc, GLfloat v) { uniform_f[uniform_name[loc]] = { v }; }
static void GLAPIENTRY Uniform2f(GLint loc, GLfloat a, GLfloat b) { uniform_f[uniform_name[loc]] = { a, b }; }
static void GLAPIENTRY Uniform1i(GLint loc, GLint v) { uniform_i[uniform_name[loc]] = v; }
static void GLAPIENTRY Uniform1ui(GLint loc, GLuint v) { uniform_i[uniform_name[loc]] = v; }
static void GLAPIENTRY UniformMatrix4fv(GLint loc, GLsizei, GLboolean, const GLfloat* v) { uniform_f[uniform_name[loc]].assign(v, v + 16); }
static void GLAPIENTRY ActiveTexture(GLenum unit) { active_unit = unit - GL_TEXTURE0; }
static void GLAPIENTRY MemoryBarrier_(GLbitfield) {}

static glm::mat4 mat4_of(const char* name) { glm::mat4 m; memcpy(&m[0][0], uniform_f[name].data(), 64); return m; }

// texture(u_depthTexture, uv): GL_NEAREST, CLAMP_TO_EDGE, single level (renderer.cpp:290-296); GL 4.6 §8.14.2
static glm::vec4 fetch_depth(int unit, glm::vec2 uv) {
    const Texture& t = textures[unit_binding[(GLuint)unit]];
    const float fu = std::floor(uv.x * (float)t.w), fv = std::floor(uv.y * (float)t.h);
    const long i = fu >= 0.0f ? (fu < (float)t.w ? (long)fu : (long)t.w - 1) : 0;
    const long j = fv >= 0.0f ? (fv < (float)t.h ? (long)fv : (long)t.h - 1) : 0;
    const float d = t.depth[(size_t)j * t.w + (size_t)i];
    return glm::vec4(d, d, d, 1.0f);
}

static void GLAPIENTRY DispatchCompute(GLuint gx, GLuint gy, GLuint gz) {
    using namespace ref_cs;
    ++n_dispatches;
    // uniforms
    u_depthTexture.unit = (int)uniform_i["u_depthTexture"];
    u_stdDev = uniform_f["u_stdDev"][0];
    u_worldToView = mat4_of("u_worldToView");
    u_viewToClip = mat4_of("u_viewToClip");
    u_modelToWorld = mat4_of("u_modelToWorld");
    u_resolution = glm::vec2(uniform_f["u_resolution"][0], uniform_f["u_resolution"][1]);
    u_nearFar = glm::vec2(uniform_f["u_nearFar"][0], uniform_f["u_nearFar"][1]);
    u_depthTestMesh = (glm::uint)uniform_i["u_depthTestMesh"];
    u_renderMode = (int)uniform_i["u_renderMode"];
    u_format = (glm::uint)uniform_i["u_format"];
    u_plyHasPbr = (glm::uint)uniform_i["u_plyHasPbr"];
    u_gaussianCount = (int)uniform_i["u_gaussianCount"];
    // buffers (bindings 0..2 shader storage, 3 atomic counter: gaussianSplattingPrepassCS.glsl:41-53)
    gaussianBuffer.gaussians = reinterpret_cast<GaussianVertex*>(buffers[indexed_binding[{ GL_SHADER_STORAGE_BUFFER, 0 }]].data());
    gaussianDepthPostFiltering.depths_vs = reinterpret_cast<float*>(buffers[indexed_binding[{ GL_SHADER_STORAGE_BUFFER, 1 }]].data());
    perQuadTransformations.ndcTransformations =
        reinterpret_cast<QuadNdcTransformation*>(buffers[indexed_binding[{ GL_SHADER_STORAGE_BUFFER, 2 }]].data());
    std::vector<uint8_t>& counter = buffers[indexed_binding[{ GL_ATOMIC_COUNTER_BUFFER, 3 }]];
    memcpy(&g_validCounter.v, counter.data(), 4);
    g_texture = fetch_depth;
    gl_NumWorkGroups = glm::uvec3(gx, gy, gz);
    for (GLuint y = 0; y < gy * 16u; ++y)
        for (GLuint x = 0; x < gx * 16u; ++x) {
            gl_GlobalInvocationID.x = x; gl_GlobalInvocationID.y = y; gl_GlobalInvocationID.z = 0;
            main_();
        }
    memcpy(counter.data(), &g_validCounter.v, 4);
}
}  // namespace swgl

PFNGLGENBUFFERSPROC __glewGenBuffers = swgl::GenBuffers;
PFNGLBINDBUFFERPROC __glewBindBuffer = swgl::BindBuffer;
PFNGLBINDBUFFERBASEPROC __glewBindBufferBase = swgl::BindBufferBase;
PFNGLBUFFERDATAPROC __glewBufferData = swgl::BufferData;
PFNGLBUFFERSUBDATAPROC __glewBufferSubData = swgl::BufferSubData;
PFNGLUSEPROGRAMPROC __glewUseProgram = swgl::UseProgram;
PFNGLDELETEPROGRAMPROC __glewDeleteProgram = swgl::DeleteProgram;
PFNGLISPROGRAMPROC __glewIsProgram = swgl::IsProgram;
PFNGLGETUNIFORMLOCATIONPROC __glewGetUniformLocation = swgl::GetUniformLocation;
PFNGLUNIFORM1FPROC __glewUniform1f = swgl::Uniform1f;
PFNGLUNIFORM2FPROC __glewUniform2f = swgl::Uniform2f;
PFNGLUNIFORM1IPROC __glewUniform1i = swgl::Uniform1i;
PFNGLUNIFORM1UIPROC __glewUniform1ui = swgl::Uniform1ui;
PFNGLUNIFORMMATRIX4FVPROC __glewUniformMatrix4fv = swgl::UniformMatrix4fv;
PFNGLACTIVETEXTUREPROC __glewActiveTexture = swgl::ActiveTexture;
PFNGLMEMORYBARRIERPROC __glewMemoryBarrier = swgl::MemoryBarrier_;
PFNGLDISPATCHCOMPUTEPROC __glewDispatchCompute = swgl::DispatchCompute;

// GL 1.1 entry points (linked directly, not through GLEW)
extern "C" {
void GLAPIENTRY glGenTextures(GLsizei n, GLuint* ids) { swgl::gen(n, ids); }
void GLAPIENTRY glBindTexture(GLenum, GLuint id) { swgl::unit_binding[swgl::active_unit] = id; }
void GLAPIENTRY glTexImage2D(GLenum, GLint, GLint, GLsizei w, GLsizei h, GLint, GLenum, GLenum, const void* data) {
    swgl::Texture& t = swgl::textures[swgl::unit_binding[swgl::active_unit]];
    t.w = w; t.h = h;
    t.depth.assign((size_t)w * h, 1.0f);
    if (data) memcpy(t.depth.data(), data, (size_t)w * h * 4);
}
void GLAPIENTRY glTexParameteri(GLenum, GLenum pname, GLint v) { swgl::textures[swgl::unit_binding[swgl::active_unit]].params[pname] = v; }
}

template <typename T> static bool rd(std::ifstream& f, T* v, size_t n = 1) { return (bool)f.read(reinterpret_cast<char*>(v), (std::streamsize)(sizeof(T) * n)); }

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: ref_prepass_check in.bin out.bin\n"); return 64; }
    std::ifstream f(argv[1], std::ios::binary);
    char magic[4];
    uint32_t n = 0;
    if (!rd(f, magic, 4) || memcmp(magic, "M2SP", 4) || !rd(f, &n)) { fprintf(stderr, "bad input\n"); return 2; }
    RenderContext rc;
    int32_t res[2];
    uint32_t depth_test, ply_has_pbr, dw, dh;
    int32_t render_mode;
    bool ok = rd(f, &rc.viewMat[0][0], 16) && rd(f, &rc.projMat[0][0], 16) && rd(f, &rc.modelMat[0][0], 16) && rd(f, res, 2) &&
              rd(f, &rc.nearPlane) && rd(f, &rc.farPlane) && rd(f, &rc.gaussianStd) && rd(f, &rc.resolutionTarget) && rd(f, &render_mode) &&
              rd(f, &rc.format) && rd(f, &ply_has_pbr) && rd(f, &depth_test) && rd(f, &dw) && rd(f, &dh);
    if (!ok) { fprintf(stderr, "short header\n"); return 2; }
    rc.rendererResolution = glm::ivec2(res[0], res[1]);
    rc.renderMode = (unsigned)render_mode;
    rc.plyHasPbr = ply_has_pbr != 0;
    rc.performMeshDepthTest = depth_test;
    rc.numberOfGaussians = n;
    std::vector<float> records((size_t)n * 24), depth((size_t)dw * dh);
    if ((n && !rd(f, records.data(), records.size())) || (depth.size() && !rd(f, depth.data(), depth.size()))) { fprintf(stderr, "short payload\n"); return 2; }

    // the buffers Renderer::initialize creates for this pass (renderer.cpp:48-77) and the mesh depth texture
    // (renderer.cpp:280-308)
    glGenBuffers(1, &rc.gaussianBuffer);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianBuffer);
    glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)(records.size() * 4), records.data(), GL_DYNAMIC_DRAW);
    glGenBuffers(1, &rc.gaussianDepthPostFiltering);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianDepthPostFiltering);
    glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)((size_t)n * 4), nullptr, GL_DYNAMIC_DRAW);
    glGenBuffers(1, &rc.perQuadTransformationsBuffer);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.perQuadTransformationsBuffer);
    glBufferData(GL_SHADER_STORAGE_BUFFER, (GLsizeiptr)((size_t)n * 96), nullptr, GL_DYNAMIC_DRAW);
    glGenBuffers(1, &rc.atomicCounterBuffer);
    glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, rc.atomicCounterBuffer);
    const uint32_t garbage = 0xDEADBEEFu;     // execute() must reset it
    glBufferData(GL_ATOMIC_COUNTER_BUFFER, 4, &garbage, GL_DYNAMIC_DRAW);
    glGenTextures(1, &rc.meshDepthTexture);
    glBindTexture(GL_TEXTURE_2D, rc.meshDepthTexture);
    glTexImage2D(GL_TEXTURE_2D, 0, GL_DEPTH_COMPONENT, (GLsizei)dw, (GLsizei)dh, 0, GL_DEPTH_COMPONENT, GL_FLOAT, depth.empty() ? nullptr : depth.data());
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MAG_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_MIN_FILTER, GL_NEAREST);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_S, GL_CLAMP_TO_EDGE);
    glTexParameteri(GL_TEXTURE_2D, GL_TEXTURE_WRAP_T, GL_CLAMP_TO_EDGE);

    GaussiansPrepass pass;
    pass.execute(rc);

    uint32_t counter = 0;
    memcpy(&counter, swgl::buffers[rc.atomicCounterBuffer].data(), 4);
    std::ofstream o(argv[2], std::ios::binary);
    o.write(reinterpret_cast<const char*>(&counter), 4);
    o.write(reinterpret_cast<const char*>(swgl::buffers[rc.perQuadTransformationsBuffer].data()), (std::streamsize)((size_t)counter * 96));
    o.write(reinterpret_cast<const char*>(swgl::buffers[rc.gaussianDepthPostFiltering].data()), (std::streamsize)((size_t)counter * 4));
    fprintf(stdout, "{\"n\": %u, \"counter\": %u, \"dispatches\": %llu, \"groups\": [%u, %u]}\n", n, counter,
            (unsigned long long)swgl::n_dispatches, ref_cs::gl_NumWorkGroups.x, ref_cs::gl_NumWorkGroups.y);
    return 0;
}
