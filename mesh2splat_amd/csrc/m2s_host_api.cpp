// m2s_host_api.cpp — C-ABI entry points of the scene I/O layer (include/m2s.h "scene I/O"):
// .glb loading (== SceneManager::loadModel minus GL) and .ply reading (== parsers::loadPlyFile).
#include "m2s_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <new>
#include <sstream>
#include <vector>

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

// parsers::loadPlyFile (parsers.cpp:516-629) reads through happly, which accepts the three PLY encodings — ascii,
// binary_little_endian, binary_big_endian — and every scalar property type; so does this reader (round 4: ascii and big-endian
// were refused).  Records are rebuilt as color = SH -> RGB (utils.cpp:51-55), alpha = sigmoid(opacity), scale = exp(scale_i) with
// w = 1, rotation = normalised quaternion stored (w,x,y,z), normal/pbr only when metallicFactor, roughnessFactor and nx,ny,nz are
// all present (hasPbr).  The fourteen mandatory properties must be stored as float (happly's getProperty<float> refuses anything
// it would have to narrow); the vertex element must come first (what every 3DGS writer produces; list properties of later elements
// are never touched).  ASCII numbers are parsed the way happly parses them: operator>> of an istringstream on the token.
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
        enum { kNone, kAscii, kLittle, kBig } format = kNone;
        bool in_vertex = false, got_end = false, vertex_first = true, seen_element = false;
        struct Prop { std::string name; size_t off; int size; bool is_float; bool is_signed; };
        std::vector<Prop> props;               // of the vertex element, in file order
        std::map<std::string, size_t> index;   // property name -> position in props
        size_t row = 0;
        while (std::fgets(line, sizeof line, f)) {
            char a[64] = "", b[64] = "", c[64] = "";
            const int k = std::sscanf(line, "%63s %63s %63s", a, b, c);
            if (k >= 1 && !std::strcmp(a, "end_header")) { got_end = true; break; }
            if (k >= 2 && !std::strcmp(a, "format"))
                format = !std::strcmp(b, "binary_little_endian") ? kLittle : !std::strcmp(b, "binary_big_endian") ? kBig : !std::strcmp(b, "ascii") ? kAscii : kNone;
            else if (k >= 3 && !std::strcmp(a, "element")) {
                in_vertex = !std::strcmp(b, "vertex");
                if (in_vertex) { n = std::strtoull(c, nullptr, 10); vertex_first = !seen_element; }
                seen_element = true;
            }
            else if (k >= 3 && !std::strcmp(a, "property") && in_vertex) {
                int sz = 0;
                bool is_float = false, is_signed = true;
                auto is = [&](const char* x, const char* y) { return !std::strcmp(b, x) || !std::strcmp(b, y); };
                if (is("float", "float32")) { sz = 4; is_float = true; }
                else if (is("double", "float64")) sz = 8;
                else if (is("int", "int32")) sz = 4;
                else if (is("uint", "uint32")) { sz = 4; is_signed = false; }
                else if (is("short", "int16")) sz = 2;
                else if (is("ushort", "uint16")) { sz = 2; is_signed = false; }
                else if (is("char", "int8")) sz = 1;
                else if (is("uchar", "uint8")) { sz = 1; is_signed = false; }
                else if (!std::strcmp(b, "list")) return fail("list properties in the vertex element are not supported");
                else return fail(std::string("unsupported property type ") + b);
                if (row > (1u << 20)) return fail("PLY row too large");
                index[c] = props.size();
                props.push_back(Prop{ c, row, sz, is_float, is_signed });
                row += (size_t)sz;
            }
        }
        if (!got_end || format == kNone) return fail("not a PLY file with an ascii / binary_little_endian / binary_big_endian body");
        if (n && !vertex_first) return fail("the vertex element must be the first element of the file");
        auto has_f = [&](const char* p) { auto it = index.find(p); return it != index.end() && props[it->second].is_float; };
        // the property offsets are resolved ONCE (a name lookup per field per row made multi-million-row files crawl)
        static const char* const kNeed[14] = { "x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2", "opacity", "scale_0", "scale_1", "scale_2",
                                               "rot_0", "rot_1", "rot_2", "rot_3" };
        static const char* const kPbr[5] = { "nx", "ny", "nz", "metallicFactor", "roughnessFactor" };
        size_t o[14], op[5] = { 0, 0, 0, 0, 0 };
        for (int i = 0; i < 14; ++i) {
            if (!has_f(kNeed[i])) return fail(std::string("missing float property ") + kNeed[i]);   // happly would throw here
            o[i] = props[index[kNeed[i]]].off;
        }
        bool has_pbr = true;
        for (int i = 0; i < 5; ++i) has_pbr = has_pbr && has_f(kPbr[i]);
        if (has_pbr) for (int i = 0; i < 5; ++i) op[i] = props[index[kPbr[i]]].off;
        // the header's vertex count is untrusted: it must fit the bytes that are actually in the file
        const long body0 = std::ftell(f);
        if (body0 < 0 || std::fseek(f, 0, SEEK_END) != 0) return fail("cannot seek in PLY file");
        const long fsize = std::ftell(f);
        if (fsize < body0 || std::fseek(f, body0, SEEK_SET) != 0) return fail("cannot seek in PLY file");
        const uint64_t body = (uint64_t)(fsize - body0);
        const uint64_t min_row = format == kAscii ? 2 * (uint64_t)props.size() : (uint64_t)row;     // ascii: a digit and a separator per value
        if (n && (min_row == 0 || n > body / min_row)) return fail("truncated PLY body");
        if (n > SIZE_MAX / sizeof(m2s_gaussian)) return fail("PLY vertex count too large");
        rec = n ? (m2s_gaussian*)std::malloc((size_t)n * sizeof(m2s_gaussian)) : nullptr;
        if (n && !rec) { std::fclose(f); g_io_error = "host allocation failed"; return M2S_ERR_OOM; }
        const float sh_c0 = 0.28209479177387814f;
        // one row (little-endian bytes at the header's offsets) -> one record
        auto convert = [&](const uint8_t* r, m2s_gaussian& g) {
            auto F = [&](size_t off) { float v; std::memcpy(&v, r + off, 4); return v; };
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
        };
        if (format == kAscii) {
            // happly: one line per vertex (empty lines skipped), split at whitespace, one token per property, each parsed by
            // operator>> into the property's type (a token that is not a number of that type yields what operator>> leaves: 0)
            std::vector<uint8_t> r(std::max<size_t>(row, 1));
            std::string text;
            std::vector<char> lbuf(1 << 16);
            for (uint64_t done = 0; done < n;) {
                if (!std::fgets(lbuf.data(), (int)lbuf.size(), f)) return fail("truncated PLY body");
                text.assign(lbuf.data());
                while (!text.empty() && text.back() != '\n' && std::fgets(lbuf.data(), (int)lbuf.size(), f)) text += lbuf.data();   // very long lines
                std::istringstream ls(text);
                std::string tok;
                size_t k = 0;
                for (; k < props.size() && (ls >> tok); ++k) {
                    const Prop& pr = props[k];
                    std::istringstream ts(tok);
                    if (pr.is_float) { float v = 0.0f; ts >> v; std::memcpy(&r[pr.off], &v, 4); }
                    else std::memset(&r[pr.off], 0, (size_t)pr.size);      // (other properties are never read back)
                }
                if (k == 0) continue;                                      // blank line
                if (k < props.size()) return fail("a vertex line of the ascii PLY body has too few values");
                convert(r.data(), rec[done]);
                ++done;
            }
        } else {
            std::vector<uint8_t> buf(std::max<size_t>(row, 1) * 4096);
            uint64_t done = 0;
            while (done < n) {
                const size_t want = (size_t)std::min<uint64_t>(4096, n - done);
                if (std::fread(buf.data(), row, want, f) != want) return fail("truncated PLY body");
                for (size_t i = 0; i < want; ++i) {
                    uint8_t* r = &buf[i * row];
                    if (format == kBig)
                        for (const Prop& pr : props) std::reverse(r + pr.off, r + pr.off + pr.size);
                    convert(r, rec[done + i]);
                }
                done += want;
            }
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
