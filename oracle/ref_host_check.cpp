// ref_host_check.cpp — TEST INFRASTRUCTURE.  Runs the REFERENCE's own host code (compiled from where it
// lies under /root/reference by oracle/Makefile, never copied here) so that the tests can pin this
// repository's loader and PLY writers/readers against it:
//
//   ref_host_check scene    in.glb  out.bin           SceneManager::loadModel (SceneManager.cpp:22-35):
//                                                     parseGltfFile -> setupMeshBuffers -> loadTextures
//   ref_host_check plywrite rec.bin out.ply fmt mult  parsers::savePlyVector (parsers.cpp:631-651)
//   ref_host_check plyread  in.ply  out.bin           parsers::loadPlyFile   (parsers.cpp:516-629)
//   ref_host_check glbwrite spec.bin out.glb          a .glb AUTHORED by the reference's own third-party stack: the scene in
//                                                     spec.bin (layout: do_glbwrite below) becomes a tinygltf::Model and is written
//                                                     by tinygltf::TinyGLTF::WriteGltfSceneToFile (binary, images embedded as PNG by
//                                                     stb_image_write) — a writer that shares no code and no author with this
//                                                     repository's mesh2splat_amd/gltf_io.py, whose files the loader tests
//                                                     otherwise read
//   ref_host_check glbwrite2 spec2.bin out.glb        the same with shared images / materials, several primitives in one mesh, JPEG
//                                                     images, an occlusion map: files shaped like SciFiHelmet.glb / Sponza.glb
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

// ---- glbwrite: spec -> tinygltf::Model -> .glb ------------------------------------------------------------
// spec.bin (little endian): u32 flags (1: 16-bit indices, 2: one interleaved vertex buffer view with byteStride, 4: no indices,
// 8: each mesh under a parent node that carries half of its translation); u32 n_meshes; per mesh: str name; u32 n_vertices;
// f32 position[3 n], normal[3 n], tangent[4 n], uv[2 n]; u32 n_indices; u32 index[n_indices]; f32 baseColorFactor[4];
// f32 translation[3], rotation[4] (x y z w), scale[3]; then three images (base colour, normal, metallic-roughness), each u32 w, h;
// u8 rgba[4 w h] (w = 0: the material has no such texture).
#include "tiny_gltf.h"
namespace {
struct SpecReader {
    std::vector<char> b; size_t at = 0; bool ok = true;
    bool get(void* p, size_t n) { if (at + n > b.size()) { ok = false; return false; } memcpy(p, b.data() + at, n); at += n; return true; }
    uint32_t u32() { uint32_t v = 0; get(&v, 4); return v; }
    std::string str() { const uint32_t n = u32(); std::string s(n, '\0'); if (n) get(&s[0], n); return s; }
    template <class T> std::vector<T> arr(size_t n) { std::vector<T> v(n); if (n) get(v.data(), n * sizeof(T)); return v; }
};
int add_view(tinygltf::Model& m, const void* data, size_t bytes, int target, size_t stride = 0) {
    tinygltf::Buffer& buf = m.buffers[0];
    while (buf.data.size() % 4) buf.data.push_back(0);
    tinygltf::BufferView v;
    v.buffer = 0; v.byteOffset = buf.data.size(); v.byteLength = bytes; v.byteStride = stride; v.target = target;
    const unsigned char* p = static_cast<const unsigned char*>(data);
    buf.data.insert(buf.data.end(), p, p + bytes);
    m.bufferViews.push_back(v);
    return (int)m.bufferViews.size() - 1;
}
int add_accessor(tinygltf::Model& m, int view, size_t offset, int comp, int type, size_t count, const float* minv = nullptr, const float* maxv = nullptr) {
    tinygltf::Accessor a;
    a.bufferView = view; a.byteOffset = offset; a.componentType = comp; a.type = type; a.count = count;
    if (minv) { a.minValues.assign(minv, minv + 3); a.maxValues.assign(maxv, maxv + 3); }
    m.accessors.push_back(a);
    return (int)m.accessors.size() - 1;
}
}  // namespace
static int do_glbwrite(const char* in, const char* out) {
    SpecReader r;
    { std::ifstream f(in, std::ios::binary); r.b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
    const uint32_t flags = r.u32(), n_meshes = r.u32();
    tinygltf::Model m;
    m.asset.version = "2.0";
    m.asset.generator = "tiny_gltf (the reference's copy), driven by oracle/ref_host_check glbwrite";
    m.buffers.emplace_back();
    tinygltf::Scene scene;
    for (uint32_t k = 0; k < n_meshes && r.ok; ++k) {
        const std::string name = r.str();
        const uint32_t nv = r.u32();
        const auto pos = r.arr<float>(3 * (size_t)nv), nrm = r.arr<float>(3 * (size_t)nv), tan = r.arr<float>(4 * (size_t)nv), uv = r.arr<float>(2 * (size_t)nv);
        const uint32_t ni = r.u32();
        const auto idx = r.arr<uint32_t>(ni);
        float color[4], T[3], Rq[4], S[3];
        r.get(color, 16); r.get(T, 12); r.get(Rq, 16); r.get(S, 12);
        if (!r.ok) break;
        float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
        for (uint32_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], pos[3 * i + c]); mx[c] = std::max(mx[c], pos[3 * i + c]); }
        tinygltf::Primitive prim;
        prim.mode = TINYGLTF_MODE_TRIANGLES;
        if (flags & 2u) {      // one interleaved view: position | normal | tangent | uv = 48 bytes per vertex
            std::vector<float> inter((size_t)nv * 12);
            for (uint32_t i = 0; i < nv; ++i) {
                memcpy(&inter[(size_t)i * 12 + 0], &pos[3 * i], 12); memcpy(&inter[(size_t)i * 12 + 3], &nrm[3 * i], 12);
                memcpy(&inter[(size_t)i * 12 + 6], &tan[4 * i], 16); memcpy(&inter[(size_t)i * 12 + 10], &uv[2 * i], 8);
            }
            const int v = add_view(m, inter.data(), inter.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, 48);
            prim.attributes["POSITION"] = add_accessor(m, v, 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv, mn, mx);
            prim.attributes["NORMAL"] = add_accessor(m, v, 12, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv);
            prim.attributes["TANGENT"] = add_accessor(m, v, 24, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC4, nv);
            prim.attributes["TEXCOORD_0"] = add_accessor(m, v, 40, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC2, nv);
        } else {
            prim.attributes["POSITION"] = add_accessor(m, add_view(m, pos.data(), pos.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv, mn, mx);
            prim.attributes["NORMAL"] = add_accessor(m, add_view(m, nrm.data(), nrm.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv);
            prim.attributes["TANGENT"] = add_accessor(m, add_view(m, tan.data(), tan.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC4, nv);
            prim.attributes["TEXCOORD_0"] = add_accessor(m, add_view(m, uv.data(), uv.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC2, nv);
        }
        if (!(flags & 4u) && ni) {
            if (flags & 1u) {
                std::vector<uint16_t> i16(idx.begin(), idx.end());
                prim.indices = add_accessor(m, add_view(m, i16.data(), i16.size() * 2, TINYGLTF_TARGET_ELEMENT_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_UNSIGNED_SHORT, TINYGLTF_TYPE_SCALAR, ni);
            } else
                prim.indices = add_accessor(m, add_view(m, idx.data(), idx.size() * 4, TINYGLTF_TARGET_ELEMENT_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT, TINYGLTF_TYPE_SCALAR, ni);
        }
        tinygltf::Material mat;
        mat.name = name + "_material";
        mat.pbrMetallicRoughness.baseColorFactor.assign(color, color + 4);
        for (int t = 0; t < 3 && r.ok; ++t) {
            const uint32_t w = r.u32(), h = r.u32();
            if (!w) continue;
            tinygltf::Image img;
            img.width = (int)w; img.height = (int)h; img.component = 4; img.bits = 8; img.pixel_type = TINYGLTF_COMPONENT_TYPE_UNSIGNED_BYTE;
            img.image = r.arr<unsigned char>((size_t)w * h * 4);
            img.mimeType = "image/png";
            img.name = name + (t == 0 ? "_albedo" : t == 1 ? "_normal" : "_mr");
            m.images.push_back(img);
            tinygltf::Texture tex;
            tex.source = (int)m.images.size() - 1;
            m.textures.push_back(tex);
            const int ti = (int)m.textures.size() - 1;
            if (t == 0) mat.pbrMetallicRoughness.baseColorTexture.index = ti;
            else if (t == 1) mat.normalTexture.index = ti;
            else mat.pbrMetallicRoughness.metallicRoughnessTexture.index = ti;
        }
        m.materials.push_back(mat);
        prim.material = (int)m.materials.size() - 1;
        tinygltf::Mesh mesh;
        mesh.name = name;
        mesh.primitives.push_back(prim);
        m.meshes.push_back(mesh);
        tinygltf::Node node;
        node.name = name + "_node";
        node.mesh = (int)m.meshes.size() - 1;
        node.rotation.assign(Rq, Rq + 4);
        node.scale.assign(S, S + 3);
        if (flags & 8u) {      // half of the translation on a parent node
            node.translation = { T[0] * 0.5, T[1] * 0.5, T[2] * 0.5 };
            m.nodes.push_back(node);
            tinygltf::Node parent;
            parent.name = name + "_parent";
            parent.translation = { T[0] * 0.5, T[1] * 0.5, T[2] * 0.5 };
            parent.children.push_back((int)m.nodes.size() - 1);
            m.nodes.push_back(parent);
        } else {
            node.translation.assign(T, T + 3);
            m.nodes.push_back(node);
        }
        scene.nodes.push_back((int)m.nodes.size() - 1);
    }
    if (!r.ok) { fprintf(stderr, "glbwrite: truncated spec\n"); return 2; }
    m.scenes.push_back(scene);
    m.defaultScene = 0;
    tinygltf::TinyGLTF writer;
    if (!writer.WriteGltfSceneToFile(&m, out, /*embedImages*/ true, /*embedBuffers*/ true, /*prettyPrint*/ false, /*writeBinary*/ true)) {
        fprintf(stderr, "glbwrite: tiny_gltf could not write %s\n", out);
        return 3;
    }
    return 0;
}


// ---- glbwrite2: the same authoring path for files SHAPED like the assets BASELINE configs 2 and 4 name ----------
// What glbwrite cannot express: images and materials SHARED between primitives (Sponza: ~100 primitives on ~25 materials), several
// primitives in ONE mesh under ONE node, JPEG-encoded images (stb_image_write through tiny_gltf's writer, as for PNG), a fourth
// (occlusion) texture that the reference ignores, indexed geometry with real vertex reuse.
// spec2.bin (little endian): u32 flags (2: one interleaved vertex view with byteStride per primitive; 64: separate views that state
// byteStride = element size; 32: all primitives in one mesh
// under one node, whose TRS is the FIRST primitive's); u32 n_images, each: u32 w, h, jpeg (0 / 1), u8 rgba[4 w h]; u32 n_materials,
// each: str name, f32 baseColorFactor[4], i32 image of {base colour, normal, metallic-roughness, occlusion} (-1: none); u32
// n_primitives, each: str name, u32 n_vertices, f32 position[3 n], normal[3 n], tangent[4 n], uv[2 n], u32 n_indices (0: not
// indexed), u32 index[], u32 material, f32 translation[3], rotation[4] (x y z w), scale[3].
static int do_glbwrite2(const char* in, const char* out) {
    SpecReader r;
    { std::ifstream f(in, std::ios::binary); r.b.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
    const uint32_t flags = r.u32();
    tinygltf::Model m;
    m.asset.version = "2.0";
    m.asset.generator = "tiny_gltf (the reference's copy), driven by oracle/ref_host_check glbwrite2";
    m.buffers.emplace_back();
    const uint32_t n_images = r.u32();
    for (uint32_t k = 0; k < n_images && r.ok; ++k) {
        const uint32_t w = r.u32(), h = r.u32(), jpeg = r.u32();
        tinygltf::Image img;
        img.width = (int)w; img.height = (int)h; img.component = 4; img.bits = 8; img.pixel_type = TINYGLTF_COMPONENT_TYPE_UNSIGNED_BYTE;
        img.image = r.arr<unsigned char>((size_t)w * h * 4);
        img.mimeType = jpeg ? "image/jpeg" : "image/png";
        img.name = "image_" + std::to_string(k);
        m.images.push_back(img);
        tinygltf::Texture tex;
        tex.source = (int)k;
        m.textures.push_back(tex);
    }
    const uint32_t n_materials = r.u32();
    for (uint32_t k = 0; k < n_materials && r.ok; ++k) {
        tinygltf::Material mat;
        mat.name = r.str();
        float color[4];
        r.get(color, 16);
        mat.pbrMetallicRoughness.baseColorFactor.assign(color, color + 4);
        int32_t t[4];
        r.get(t, 16);
        if (t[0] >= 0) mat.pbrMetallicRoughness.baseColorTexture.index = t[0];
        if (t[1] >= 0) mat.normalTexture.index = t[1];
        if (t[2] >= 0) mat.pbrMetallicRoughness.metallicRoughnessTexture.index = t[2];
        if (t[3] >= 0) mat.occlusionTexture.index = t[3];
        m.materials.push_back(mat);
    }
    const uint32_t n_prims = r.u32();
    tinygltf::Scene scene;
    tinygltf::Mesh one_mesh;
    one_mesh.name = "mesh";
    tinygltf::Node one_node;
    one_node.name = "node";
    for (uint32_t k = 0; k < n_prims && r.ok; ++k) {
        const std::string name = r.str();
        const uint32_t nv = r.u32();
        const auto pos = r.arr<float>(3 * (size_t)nv), nrm = r.arr<float>(3 * (size_t)nv), tan = r.arr<float>(4 * (size_t)nv), uv = r.arr<float>(2 * (size_t)nv);
        const uint32_t ni = r.u32();
        const auto idx = r.arr<uint32_t>(ni);
        const uint32_t material = r.u32();
        float T[3], Rq[4], S[3];
        r.get(T, 12); r.get(Rq, 16); r.get(S, 12);
        if (!r.ok) break;
        float mn[3] = { 1e30f, 1e30f, 1e30f }, mx[3] = { -1e30f, -1e30f, -1e30f };
        for (uint32_t i = 0; i < nv; ++i) for (int c = 0; c < 3; ++c) { mn[c] = std::min(mn[c], pos[3 * i + c]); mx[c] = std::max(mx[c], pos[3 * i + c]); }
        tinygltf::Primitive prim;
        prim.mode = TINYGLTF_MODE_TRIANGLES;
        if (flags & 2u) {
            std::vector<float> inter((size_t)nv * 12);
            for (uint32_t i = 0; i < nv; ++i) {
                memcpy(&inter[(size_t)i * 12 + 0], &pos[3 * i], 12); memcpy(&inter[(size_t)i * 12 + 3], &nrm[3 * i], 12);
                memcpy(&inter[(size_t)i * 12 + 6], &tan[4 * i], 16); memcpy(&inter[(size_t)i * 12 + 10], &uv[2 * i], 8);
            }
            const int v = add_view(m, inter.data(), inter.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, 48);
            prim.attributes["POSITION"] = add_accessor(m, v, 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv, mn, mx);
            prim.attributes["NORMAL"] = add_accessor(m, v, 12, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv);
            prim.attributes["TANGENT"] = add_accessor(m, v, 24, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC4, nv);
            prim.attributes["TEXCOORD_0"] = add_accessor(m, v, 40, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC2, nv);
        } else {
            // (64: the views say byteStride = element size, as the exporters of the Khronos sample assets write it)
            const size_t s3 = (flags & 64u) ? 12 : 0, s4 = (flags & 64u) ? 16 : 0, s2 = (flags & 64u) ? 8 : 0;
            prim.attributes["POSITION"] = add_accessor(m, add_view(m, pos.data(), pos.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, s3), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv, mn, mx);
            prim.attributes["NORMAL"] = add_accessor(m, add_view(m, nrm.data(), nrm.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, s3), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC3, nv);
            prim.attributes["TANGENT"] = add_accessor(m, add_view(m, tan.data(), tan.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, s4), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC4, nv);
            prim.attributes["TEXCOORD_0"] = add_accessor(m, add_view(m, uv.data(), uv.size() * 4, TINYGLTF_TARGET_ARRAY_BUFFER, s2), 0, TINYGLTF_COMPONENT_TYPE_FLOAT, TINYGLTF_TYPE_VEC2, nv);
        }
        if (ni) prim.indices = add_accessor(m, add_view(m, idx.data(), idx.size() * 4, TINYGLTF_TARGET_ELEMENT_ARRAY_BUFFER), 0, TINYGLTF_COMPONENT_TYPE_UNSIGNED_INT, TINYGLTF_TYPE_SCALAR, ni);
        if (material >= n_materials) { fprintf(stderr, "glbwrite2: primitive %u names material %u of %u\n", k, material, n_materials); return 2; }
        prim.material = (int)material;
        if (flags & 32u) {
            one_mesh.primitives.push_back(prim);
            if (k == 0) { one_node.translation.assign(T, T + 3); one_node.rotation.assign(Rq, Rq + 4); one_node.scale.assign(S, S + 3); }
        } else {
            tinygltf::Mesh mesh;
            mesh.name = name;
            mesh.primitives.push_back(prim);
            m.meshes.push_back(mesh);
            tinygltf::Node node;
            node.name = name + "_node";
            node.mesh = (int)m.meshes.size() - 1;
            node.translation.assign(T, T + 3); node.rotation.assign(Rq, Rq + 4); node.scale.assign(S, S + 3);
            m.nodes.push_back(node);
            scene.nodes.push_back((int)m.nodes.size() - 1);
        }
    }
    if (!r.ok) { fprintf(stderr, "glbwrite2: truncated spec\n"); return 2; }
    if (flags & 32u) {
        m.meshes.push_back(one_mesh);
        one_node.mesh = 0;
        m.nodes.push_back(one_node);
        scene.nodes.push_back(0);
    }
    m.scenes.push_back(scene);
    m.defaultScene = 0;
    tinygltf::TinyGLTF writer;
    if (!writer.WriteGltfSceneToFile(&m, out, /*embedImages*/ true, /*embedBuffers*/ true, /*prettyPrint*/ false, /*writeBinary*/ true)) {
        fprintf(stderr, "glbwrite2: tiny_gltf could not write %s\n", out);
        return 3;
    }
    return 0;
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
    if (mode == "glbwrite" && argc == 4) return do_glbwrite(argv[2], argv[3]);
    if (mode == "glbwrite2" && argc == 4) return do_glbwrite2(argv[2], argv[3]);
    fprintf(stderr, "usage: ref_host_check scene in.glb out.bin | plywrite rec.bin out.ply fmt mult | plyread in.ply out.bin | glbwrite spec.bin out.glb | glbwrite2 spec2.bin out.glb\n");
    return 64;
}
