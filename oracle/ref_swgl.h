// ref_swgl.h — TEST INFRASTRUCTURE: the minimal software GL the reference's host code runs on in oracle/_ref
// (ref_pipeline_check.cpp, ref_dropin_check.cpp): buffers, vertex arrays, textures, uniforms, the atomic counter and the
// shader-storage buffer as plain bookkeeping, the GLEW entry points the reference's translation units link against, and the
// GL 1.1 functions they call directly.  glDrawArrays is NOT here: each harness provides its own.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

extern "C" {
#include "m2s_oracle.h"
}

// =====================================================================================================
// minimal software GL
// =====================================================================================================
namespace swgl {
struct Texture {
    uint32_t w = 0, h = 0;
    std::vector<uint8_t> chain;          // RGBA8 mip chain (orc_build_mips)
    std::map<GLenum, GLint> params;
    bool mipmapped = false;
};
struct VertexArray {
    GLuint array_buffer = 0;             // buffer bound to GL_ARRAY_BUFFER when the attribute pointers were set
    struct Attr { GLint size = 0; GLsizei stride = 0; size_t offset = 0; bool enabled = false; } attr[8];
};
static GLuint next_id = 1;
static std::map<GLuint, std::vector<uint8_t>> buffers;
static std::map<GLenum, GLuint> bound_buffer;          // per target
static std::map<std::pair<GLenum, GLuint>, GLuint> indexed_binding;
static std::map<GLuint, VertexArray> vaos;
static GLuint bound_vao = 0;
static std::map<GLuint, Texture> textures;
static GLuint active_unit = 0;
static std::map<GLuint, GLuint> unit_binding;          // texture unit -> texture id (GL_TEXTURE_2D)
static std::map<std::string, GLint> uniform_location;  // one program: name -> location
static std::map<GLint, std::string> uniform_name;
static std::map<std::string, std::vector<float>> uniform_f;
static std::map<std::string, int> uniform_i;
static GLint viewport[4] = { 0, 0, 0, 0 };
static std::map<GLenum, bool> caps;
static GLenum last_error = GL_NO_ERROR;
static uint64_t n_draws = 0;

static void gen(GLsizei n, GLuint* ids) { for (GLsizei i = 0; i < n; ++i) ids[i] = next_id++; }

// ---- buffers
static void GLAPIENTRY GenBuffers(GLsizei n, GLuint* ids) { gen(n, ids); }
static void GLAPIENTRY BindBuffer(GLenum target, GLuint id) { bound_buffer[target] = id; }
static void GLAPIENTRY BindBufferBase(GLenum target, GLuint index, GLuint id) { indexed_binding[{ target, index }] = id; bound_buffer[target] = id; }
static void GLAPIENTRY BufferData(GLenum target, GLsizeiptr size, const void* data, GLenum) {
    auto& b = buffers[bound_buffer[target]];
    b.assign((size_t)size, 0);
    if (data) std::memcpy(b.data(), data, (size_t)size);
}
static void GLAPIENTRY BufferSubData(GLenum target, GLintptr off, GLsizeiptr size, const void* data) {
    auto& b = buffers[bound_buffer[target]];
    if ((size_t)off + (size_t)size > b.size()) { last_error = GL_INVALID_VALUE; return; }
    std::memcpy(b.data() + off, data, (size_t)size);
}
static void GLAPIENTRY GetBufferSubData(GLenum target, GLintptr off, GLsizeiptr size, void* data) {
    auto& b = buffers[bound_buffer[target]];
    if ((size_t)off + (size_t)size > b.size()) { last_error = GL_INVALID_VALUE; return; }   // GL: nothing is copied
    std::memcpy(data, b.data() + off, (size_t)size);
}
static void GLAPIENTRY GetBufferParameteriv(GLenum target, GLenum pname, GLint* out) {
    if (pname == GL_BUFFER_SIZE) *out = (GLint)buffers[bound_buffer[target]].size();
}
// ---- vertex arrays
static void GLAPIENTRY GenVertexArrays(GLsizei n, GLuint* ids) { gen(n, ids); }
static void GLAPIENTRY BindVertexArray(GLuint id) { bound_vao = id; }
static void GLAPIENTRY EnableVertexAttribArray(GLuint i) { vaos[bound_vao].attr[i].enabled = true; }
static void GLAPIENTRY VertexAttribPointer(GLuint i, GLint size, GLenum type, GLboolean, GLsizei stride, const void* ptr) {
    if (type != GL_FLOAT) { fprintf(stderr, "swgl: non-float attribute\n"); exit(70); }
    VertexArray& va = vaos[bound_vao];
    va.array_buffer = bound_buffer[GL_ARRAY_BUFFER];
    va.attr[i].size = size; va.attr[i].stride = stride; va.attr[i].offset = (size_t)ptr;
}
// ---- framebuffer objects: the pass renders into a dummy attachment; nothing to emulate
static void GLAPIENTRY GenFramebuffers(GLsizei n, GLuint* ids) { gen(n, ids); }
static void GLAPIENTRY GenRenderbuffers(GLsizei n, GLuint* ids) { gen(n, ids); }
static void GLAPIENTRY BindFramebuffer(GLenum, GLuint) {}
static void GLAPIENTRY BindRenderbuffer(GLenum, GLuint) {}
static void GLAPIENTRY RenderbufferStorage(GLenum, GLenum, GLsizei, GLsizei) {}
static void GLAPIENTRY FramebufferRenderbuffer(GLenum, GLenum, GLenum, GLuint) {}
static GLenum GLAPIENTRY CheckFramebufferStatus(GLenum) { return GL_FRAMEBUFFER_COMPLETE; }
static void GLAPIENTRY DeleteFramebuffers(GLsizei, const GLuint*) {}
static void GLAPIENTRY DeleteRenderbuffers(GLsizei, const GLuint*) {}
// ---- program / uniforms (one program: the converter)
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
static void GLAPIENTRY Uniform1i(GLint loc, GLint v) { uniform_i[uniform_name[loc]] = v; }
static void GLAPIENTRY Uniform3f(GLint loc, GLfloat a, GLfloat b, GLfloat c) { uniform_f[uniform_name[loc]] = { a, b, c }; }
static void GLAPIENTRY Uniform4f(GLint loc, GLfloat a, GLfloat b, GLfloat c, GLfloat d) { uniform_f[uniform_name[loc]] = { a, b, c, d }; }
// ---- misc
static void GLAPIENTRY ActiveTexture(GLenum unit) { active_unit = unit - GL_TEXTURE0; }
static void GLAPIENTRY GenerateMipmap(GLenum) {
    Texture& t = textures[unit_binding[active_unit]];
    // GenerateMipmap: levels 1.. by 2x2 box filtering (the oracle's pinned rule); GL_TEXTURE_MAX_LEVEL is applied at sampling
    std::vector<uint8_t> base(t.chain.begin(), t.chain.begin() + (size_t)t.w * t.h * 4);
    t.chain.assign((size_t)orc_mip_total_texels(t.w, t.h) * 4, 0);
    uint64_t offs[8];
    orc_build_mips(base.data(), t.w, t.h, t.chain.data(), offs);
    t.mipmapped = true;
}
static void GLAPIENTRY MemoryBarrier_(GLbitfield) {}
}  // namespace swgl

PFNGLGENBUFFERSPROC __glewGenBuffers = swgl::GenBuffers;
PFNGLBINDBUFFERPROC __glewBindBuffer = swgl::BindBuffer;
PFNGLBINDBUFFERBASEPROC __glewBindBufferBase = swgl::BindBufferBase;
PFNGLBUFFERDATAPROC __glewBufferData = swgl::BufferData;
PFNGLBUFFERSUBDATAPROC __glewBufferSubData = swgl::BufferSubData;
PFNGLGETBUFFERSUBDATAPROC __glewGetBufferSubData = swgl::GetBufferSubData;
PFNGLGETBUFFERPARAMETERIVPROC __glewGetBufferParameteriv = swgl::GetBufferParameteriv;
PFNGLGENVERTEXARRAYSPROC __glewGenVertexArrays = swgl::GenVertexArrays;
PFNGLBINDVERTEXARRAYPROC __glewBindVertexArray = swgl::BindVertexArray;
PFNGLENABLEVERTEXATTRIBARRAYPROC __glewEnableVertexAttribArray = swgl::EnableVertexAttribArray;
PFNGLVERTEXATTRIBPOINTERPROC __glewVertexAttribPointer = swgl::VertexAttribPointer;
PFNGLGENFRAMEBUFFERSPROC __glewGenFramebuffers = swgl::GenFramebuffers;
PFNGLGENRENDERBUFFERSPROC __glewGenRenderbuffers = swgl::GenRenderbuffers;
PFNGLBINDFRAMEBUFFERPROC __glewBindFramebuffer = swgl::BindFramebuffer;
PFNGLBINDRENDERBUFFERPROC __glewBindRenderbuffer = swgl::BindRenderbuffer;
PFNGLRENDERBUFFERSTORAGEPROC __glewRenderbufferStorage = swgl::RenderbufferStorage;
PFNGLFRAMEBUFFERRENDERBUFFERPROC __glewFramebufferRenderbuffer = swgl::FramebufferRenderbuffer;
PFNGLCHECKFRAMEBUFFERSTATUSPROC __glewCheckFramebufferStatus = swgl::CheckFramebufferStatus;
PFNGLDELETEFRAMEBUFFERSPROC __glewDeleteFramebuffers = swgl::DeleteFramebuffers;
PFNGLDELETERENDERBUFFERSPROC __glewDeleteRenderbuffers = swgl::DeleteRenderbuffers;
PFNGLUSEPROGRAMPROC __glewUseProgram = swgl::UseProgram;
PFNGLDELETEPROGRAMPROC __glewDeleteProgram = swgl::DeleteProgram;
PFNGLISPROGRAMPROC __glewIsProgram = swgl::IsProgram;
PFNGLGETUNIFORMLOCATIONPROC __glewGetUniformLocation = swgl::GetUniformLocation;
PFNGLUNIFORM1IPROC __glewUniform1i = swgl::Uniform1i;
PFNGLUNIFORM3FPROC __glewUniform3f = swgl::Uniform3f;
PFNGLUNIFORM4FPROC __glewUniform4f = swgl::Uniform4f;
PFNGLACTIVETEXTUREPROC __glewActiveTexture = swgl::ActiveTexture;
PFNGLGENERATEMIPMAPPROC __glewGenerateMipmap = swgl::GenerateMipmap;
PFNGLMEMORYBARRIERPROC __glewMemoryBarrier = swgl::MemoryBarrier_;

// GL 1.1 entry points (linked directly, not through GLEW)
extern "C" {
void GLAPIENTRY glGenTextures(GLsizei n, GLuint* ids) { swgl::gen(n, ids); }
void GLAPIENTRY glDeleteTextures(GLsizei n, const GLuint* ids) { for (GLsizei i = 0; i < n; ++i) swgl::textures.erase(ids[i]); }
void GLAPIENTRY glBindTexture(GLenum, GLuint id) { swgl::unit_binding[swgl::active_unit] = id; }
void GLAPIENTRY glTexImage2D(GLenum, GLint level, GLint, GLsizei w, GLsizei h, GLint, GLenum format, GLenum type, const void* data) {
    if (level != 0 || type != GL_UNSIGNED_BYTE || (format != GL_RGBA && format != GL_RGB)) { fprintf(stderr, "swgl: unsupported glTexImage2D\n"); exit(70); }
    swgl::Texture& t = swgl::textures[swgl::unit_binding[swgl::active_unit]];
    t.w = (uint32_t)w; t.h = (uint32_t)h;
    t.chain.assign((size_t)w * h * 4, 255);
    const uint8_t* src = static_cast<const uint8_t*>(data);
    const int c = format == GL_RGBA ? 4 : 3;
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int k = 0; k < c; ++k) t.chain[i * 4 + k] = src[i * c + k];
    t.mipmapped = false;
}
void GLAPIENTRY glTexParameteri(GLenum, GLenum pname, GLint v) { swgl::textures[swgl::unit_binding[swgl::active_unit]].params[pname] = v; }
void GLAPIENTRY glViewport(GLint x, GLint y, GLsizei w, GLsizei h) { swgl::viewport[0] = x; swgl::viewport[1] = y; swgl::viewport[2] = w; swgl::viewport[3] = h; }
void GLAPIENTRY glEnable(GLenum cap) { swgl::caps[cap] = true; }
void GLAPIENTRY glDisable(GLenum cap) { swgl::caps[cap] = false; }
void GLAPIENTRY glFinish(void) {}
GLenum GLAPIENTRY glGetError(void) { const GLenum e = swgl::last_error; swgl::last_error = GL_NO_ERROR; return e; }
}  // extern "C"
