// m2s_host_api.cpp — C-ABI entry points of the scene I/O layer (include/m2s.h "scene I/O"):
// .glb loading (== SceneManager::loadModel minus GL) and .ply reading (== parsers::loadPlyFile).
#include "m2s_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>

struct m2s_host_scene {
    m2s_host::HostScene scene;
    std::string warnings_joined;
};

namespace {
thread_local std::string g_io_error;
}

extern "C" {

const char* m2s_io_last_error(void) { return g_io_error.c_str(); }

m2s_status m2s_load_glb(const char* path, m2s_host_scene** out_scene) {
    if (!path || !out_scene) { g_io_error = "NULL argument"; return M2S_ERR_INVALID; }
    *out_scene = nullptr;
    m2s_host_scene* hs = new (std::nothrow) m2s_host_scene();
    if (!hs) { g_io_error = "host allocation failed"; return M2S_ERR_OOM; }
    std::string err;
    bool ok = false;
    try { ok = m2s_host::load_glb(path, hs->scene, err); }
    catch (const std::bad_alloc&) { delete hs; g_io_error = "host allocation failed"; return M2S_ERR_OOM; }
    catch (const std::exception& e) { err = e.what(); }
    if (!ok) {
        delete hs;
        g_io_error = "Failed to parse GLTF file: " + std::string(path) + ": " + err;  // SceneManager.cpp:24-27
        return M2S_ERR_IO;
    }
    for (const auto& w : hs->scene.warnings) hs->warnings_joined += w + "\n";
    *out_scene = hs;
    return M2S_OK;
}

void m2s_free_host_scene(m2s_host_scene* s) { delete s; }
uint32_t m2s_host_scene_num_meshes(const m2s_host_scene* s) { return s ? (uint32_t)s->scene.c_meshes.size() : 0; }
const m2s_mesh* m2s_host_scene_meshes(const m2s_host_scene* s) { return s && !s->scene.c_meshes.empty() ? s->scene.c_meshes.data() : nullptr; }
const char* m2s_host_scene_mesh_name(const m2s_host_scene* s, uint32_t i) {
    return s && i < s->scene.meshes.size() ? s->scene.meshes[i].name.c_str() : "";
}
const char* m2s_host_scene_warnings(const m2s_host_scene* s) { return s ? s->warnings_joined.c_str() : ""; }

// parsers::loadPlyFile (parsers.cpp:516-629): binary little-endian .ply with float properties; records
// are rebuilt as color = SH -> RGB (utils.cpp:51-55), alpha = sigmoid(opacity), scale = exp(scale_i) with
// w = 1, rotation = normalised quaternion stored (w,x,y,z), normal/pbr only when metallicFactor,
// roughnessFactor and nx,ny,nz are all present (hasPbr).
m2s_status m2s_read_ply(const char* path, m2s_gaussian** out_records, uint64_t* out_n, int* out_has_pbr) {
    if (!path || !out_records || !out_n) { g_io_error = "NULL argument"; return M2S_ERR_INVALID; }
    *out_records = nullptr;
    *out_n = 0;
    FILE* f = std::fopen(path, "rb");
    if (!f) { g_io_error = std::string("Error loading PLY file: ") + path; return M2S_ERR_IO; }
    m2s_gaussian* rec = nullptr;
    // nothing may throw across the C ABI: std::string / std::map / std::vector operations below are inside the try
    try {
        auto fail = [&](const std::string& m) { std::fclose(f); std::free(rec); g_io_error = m; return M2S_ERR_IO; };
        char line[512];
        if (!std::fgets(line, sizeof line, f) || std::strncmp(line, "ply", 3) != 0) return fail("not a PLY file");
        uint64_t n = 0;
        bool binary_le = false, in_vertex = false, got_end = false;
        std::map<std::string, size_t> offset;  // property name -> byte offset in a row
        std::map<std::string, int> size_of;
        size_t row = 0;
        while (std::fgets(line, sizeof line, f)) {
            char a[64] = "", b[64] = "", c[64] = "";
            const int k = std::sscanf(line, "%63s %63s %63s", a, b, c);
            if (k >= 1 && !std::strcmp(a, "end_header")) { got_end = true; break; }
            if (k >= 2 && !std::strcmp(a, "format")) binary_le = !std::strcmp(b, "binary_little_endian");
            else if (k >= 3 && !std::strcmp(a, "element")) { in_vertex = !std::strcmp(b, "vertex"); if (in_vertex) n = std::strtoull(c, nullptr, 10); }
            else if (k >= 3 && !std::strcmp(a, "property") && in_vertex) {
                int sz = 0;
                if (!std::strcmp(b, "float") || !std::strcmp(b, "float32") || !std::strcmp(b, "int") || !std::strcmp(b, "uint")) sz = 4;
                else if (!std::strcmp(b, "uchar") || !std::strcmp(b, "uint8") || !std::strcmp(b, "char") || !std::strcmp(b, "int8")) sz = 1;
                else if (!std::strcmp(b, "short") || !std::strcmp(b, "ushort")) sz = 2;
                else if (!std::strcmp(b, "double")) sz = 8;
                else return fail(std::string("unsupported property type ") + b);
                if (row > (1u << 20)) return fail("PLY row too large");
                offset[c] = row;
                size_of[c] = (!std::strcmp(b, "float") || !std::strcmp(b, "float32")) ? 4 : -sz;
                row += (size_t)sz;
            }
        }
        if (!got_end || !binary_le) return fail("only binary_little_endian PLY files are supported");
        auto has_f = [&](const char* p) { auto it = size_of.find(p); return it != size_of.end() && it->second == 4; };
        // the property offsets are resolved ONCE (a name lookup per field per row made multi-million-row files crawl)
        static const char* const kNeed[14] = { "x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                                               "rot_0", "rot_1", "rot_2", "rot_3" };
        static const char* const kPbr[5] = { "nx", "ny", "nz", "metallicFactor", "roughnessFactor" };
        size_t o[14], op[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < 14; ++i) {
            if (!has_f(kNeed[i])) return fail(std::string("missing float property ") + kNeed[i]);   // happly would throw here
            o[i] = offset[kNeed[i]];
        }
        bool has_pbr = true;
        for (int i = 0; i < 5; ++i) has_pbr = has_pbr && has_f(kPbr[i]);
        if (has_pbr) for (int i = 0; i < 5; ++i) op[i] = offset[kPbr[i]];
        // the header's vertex count is untrusted: it must fit the bytes that are actually in the file
        const long body0 = std::ftell(f);
        if (body0 < 0 || std::fseek(f, 0, SEEK_END) != 0) return fail("cannot seek in PLY file");
        const long fsize = std::ftell(f);
        if (fsize < body0 || std::fseek(f, body0, SEEK_SET) != 0) return fail("cannot seek in PLY file");
        const uint64_t body = (uint64_t)(fsize - body0);
        if (n && (row == 0 || n > body / row)) return fail("truncated PLY body");
        if (n > SIZE_MAX / sizeof(m2s_gaussian)) return fail("PLY vertex count too large");
        rec = n ? (m2s_gaussian*)std::malloc((size_t)n * sizeof(m2s_gaussian)) : nullptr;
        if (n && !rec) { std::fclose(f); g_io_error = "host allocation failed"; return M2S_ERR_OOM; }
        std::vector<uint8_t> buf(std::max<size_t>(row, 1) * 4096);
        const float sh_c0 = 0.28209479177387814f;
        uint64_t done = 0;
        while (done < n) {
            const size_t want = (size_t)std::min<uint64_t>(4096, n - done);
            if (std::fread(buf.data(), row, want, f) != want) return fail("truncated PLY body");
            for (size_t i = 0; i < want; ++i) {
                const uint8_t* r = &buf[i * row];
                auto F = [&](size_t off) { float v; std::memcpy(&v, r + off, 4); return v; };
                m2s_gaussian& g = rec[done + i];
                g.position[0] = F(o[0]); g.position[1] = F(o[1]); g.position[2] = F(o[2]); g.position[3] = 1.0f;
                g.color[0] = F(o[3]) * sh_c0 + 0.5f; g.color[1] = F(o[4]) * sh_c0 + 0.5f; g.color[2] = F(o[5]) * sh_c0 + 0.5f;
                g.color[3] = (float)(1.0 / (1.0 + std::exp(-F(o[6]))));   // utils.hpp:269 (double arithmetic, then float)
                g.scale[0] = std::exp(F(o[7])); g.scale[1] = std::exp(F(o[8])); g.scale[2] = std::exp(F(o[9])); g.scale[3] = 1.0f;
                if (has_pbr) { g.normal[0] = F(op[0]); g.normal[1] = F(op[1]); g.normal[2] = F(op[2]); g.normal[3] = 0.0f; }
                else g.normal[0] = g.normal[1] = g.normal[2] = g.normal[3] = 0.0f;
                const float qw = F(o[10]), qx = F(o[11]), qy = F(o[12]), qz = F(o[13]);
                const float len = std::sqrt((qw * qw + qx * qx) + (qy * qy + qz * qz));   // glm::dot(quat, quat) pairs the products
                if (len <= 0.0f) { g.rotation[0] = 1.0f; g.rotation[1] = g.rotation[2] = g.rotation[3] = 0.0f; }   // glm::normalize(quat) of zero
                else { const float inv = 1.0f / len; g.rotation[0] = qw * inv; g.rotation[1] = qx * inv; g.rotation[2] = qy * inv; g.rotation[3] = qz * inv; }
                if (has_pbr) { g.pbr[0] = F(op[3]); g.pbr[1] = F(op[4]); g.pbr[2] = 0.0f; g.pbr[3] = 0.0f; }
                else g.pbr[0] = g.pbr[1] = g.pbr[2] = g.pbr[3] = 0.0f;
            }
            done += want;
        }
        std::fclose(f);
        *out_records = rec;
        *out_n = n;
        if (out_has_pbr) *out_has_pbr = has_pbr ? 1 : 0;
        return M2S_OK;
    } catch (const std::bad_alloc&) {
        std::fclose(f); std::free(rec);
        g_io_error = "host allocation failed";
        return M2S_ERR_OOM;
    } catch (...) {
        std::fclose(f); std::free(rec);
        g_io_error = "unexpected failure while reading the PLY file";
        return M2S_ERR_IO;
    }
}

void m2s_free_records(m2s_gaussian* records) { std::free(records); }

}  // extern "C"
