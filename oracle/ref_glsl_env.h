// ref_glsl_env.h — TEST INFRASTRUCTURE.  The reference's three conversion shaders as C++ (see glsl2cpp.py): included by
// ref_glsl_check.cpp (stage-by-stage comparison) and ref_pipeline_check.cpp (whole pass on a minimal software GL).
// Namespaces ref_vs / ref_gs / ref_fs hold the shaders' globals (inputs, outputs, uniforms) and their main_() functions.
#pragma once
#ifndef GLM_FORCE_SWIZZLE
#define GLM_FORCE_SWIZZLE   // function swizzles (.xyz()); must precede the first glm include of the translation unit
#endif
#include <glm/glm.hpp>

#include <vector>

// ---- GLSL environment shared by the three stages -----------------------------------------------------------
namespace glsl_env {
using namespace glm;

struct Emitted { vec4 gl_Position; vec3 Position, Scale, Normal; vec2 UV; vec4 Tangent, Quaternion; };
static std::vector<Emitted> g_emitted;

struct sampler2D { int unit; };
struct atomic_uint { unsigned v; };
static inline uint atomicCounterIncrement(atomic_uint& c) { return c.v++; }

// texture(): fixed-function in GL; each harness installs its own fetch (texture unit, uv) -> RGBA
static vec4 (*g_texture)(int unit, vec2 uv) = nullptr;
static inline vec4 texture(const sampler2D& s, vec2 uv) { return g_texture(s.unit, uv); }
}  // namespace glsl_env

namespace ref_vs {
using namespace glm;
#include "_ref/gen/converterVS.inc"
}  // namespace ref_vs

namespace ref_gs {
using namespace glm;
static vec4 gl_Position;
static void EmitVertex();
static void EndPrimitive() {}
#include "_ref/gen/converterGS.inc"
static void EmitVertex() {
    glsl_env::g_emitted.push_back({ gl_Position, Position, Scale, Normal, UV, Tangent, Quaternion });
}
}  // namespace ref_gs

namespace ref_fs {
using namespace glm;
using glsl_env::atomic_uint;
using glsl_env::atomicCounterIncrement;
using glsl_env::sampler2D;
using glsl_env::texture;
#include "_ref/gen/converterFS.inc"
}  // namespace ref_fs
