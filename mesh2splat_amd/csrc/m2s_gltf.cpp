// m2s_gltf.cpp — .glb scene loader: the host half of the reference's load path re-hosted without GL,
// tiny_gltf or glm.  Follows SceneManager::loadModel (src/utils/SceneManager.cpp:22-35):
//   parseGltfFile      :195-459  scene graph -> world matrices, de-indexing into faces, fallback
//                                normals (:406-413) and UV-derived tangents (:421-451), material parse (:99-193)
//   setupMeshBuffers   :468-576  17-float interleaved vertex buffer + CUMULATIVE bbox (:476-477,514-527)
//   loadTextures       :578-649  albedo / normal / metallic-roughness RGBA8 images per mesh
// Hardening beyond the reference: accessor byteStride and normalised-integer
// TEXCOORD_0 are honoured (SceneManager::getBufferData :50-61 ignores both).
#include "m2s_host.h"
#include "m2s_json.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <functional>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>

namespace m2s_host {

namespace {

// ---- the small subset of glm the loader needs, with glm's operation order ------------------------
struct V3 { float x, y, z; };
struct M4 { float c[4][4]; };  // c[col][row]

M4 identity() { M4 m{}; for (int i = 0; i < 4; ++i) m.c[i][i] = 1.0f; return m; }
M4 mul(const M4& a, const M4& b) {  // glm: result[j] = a[0]*b[j][0] + a[1]*b[j][1] + a[2]*b[j][2] + a[3]*b[j][3]
    M4 r{};
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            r.c[j][i] = ((a.c[0][i] * b.c[j][0] + a.c[1][i] * b.c[j][1]) + a.c[2][i] * b.c[j][2]) + a.c[3][i] * b.c[j][3];
    return r;
}
V3 xform_point(const M4& m, V3 p) {  // vec3(m * vec4(p, 1)); glm: (m0*x + m1*y) + (m2*z + m3*w)
    V3 r;
    r.x = (m.c[0][0] * p.x + m.c[1][0] * p.y) + (m.c[2][0] * p.z + m.c[3][0] * 1.0f);
    r.y = (m.c[0][1] * p.x + m.c[1][1] * p.y) + (m.c[2][1] * p.z + m.c[3][1] * 1.0f);
    r.z = (m.c[0][2] * p.x + m.c[1][2] * p.y) + (m.c[2][2] * p.z + m.c[3][2] * 1.0f);
    return r;
}
struct M3 { float c[3][3]; };
M3 upper3(const M4& m) { M3 r; for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) r.c[j][i] = m.c[j][i]; return r; }
V3 mul(const M3& m, V3 v) {  // glm: m[0]*v.x + m[1]*v.y + m[2]*v.z
    return { (m.c[0][0] * v.x + m.c[1][0] * v.y) + m.c[2][0] * v.z, (m.c[0][1] * v.x + m.c[1][1] * v.y) + m.c[2][1] * v.z,
             (m.c[0][2] * v.x + m.c[1][2] * v.y) + m.c[2][2] * v.z };
}
M3 inverse_transpose(const M3& m) {  // glm::transpose(glm::inverse(m))
    const float det = +m.c[0][0] * (m.c[1][1] * m.c[2][2] - m.c[2][1] * m.c[1][2]) - m.c[1][0] * (m.c[0][1] * m.c[2][2] - m.c[2][1] * m.c[0][2]) +
                      m.c[2][0] * (m.c[0][1] * m.c[1][2] - m.c[1][1] * m.c[0][2]);
    const float id = 1.0f / det;
    M3 inv;
    inv.c[0][0] = +(m.c[1][1] * m.c[2][2] - m.c[2][1] * m.c[1][2]) * id;
    inv.c[1][0] = -(m.c[1][0] * m.c[2][2] - m.c[2][0] * m.c[1][2]) * id;
    inv.c[2][0] = +(m.c[1][0] * m.c[2][1] - m.c[2][0] * m.c[1][1]) * id;
    inv.c[0][1] = -(m.c[0][1] * m.c[2][2] - m.c[2][1] * m.c[0][2]) * id;
    inv.c[1][1] = +(m.c[0][0] * m.c[2][2] - m.c[2][0] * m.c[0][2]) * id;
    inv.c[2][1] = -(m.c[0][0] * m.c[2][1] - m.c[2][0] * m.c[0][1]) * id;
    inv.c[0][2] = +(m.c[0][1] * m.c[1][2] - m.c[1][1] * m.c[0][2]) * id;
    inv.c[1][2] = -(m.c[0][0] * m.c[1][2] - m.c[1][0] * m.c[0][2]) * id;
    inv.c[2][2] = +(m.c[0][0] * m.c[1][1] - m.c[1][0] * m.c[0][1]) * id;
    M3 t;
    for (int j = 0; j < 3; ++j) for (int i = 0; i < 3; ++i) t.c[j][i] = inv.c[i][j];
    return t;
}
V3 sub(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
V3 cross(V3 a, V3 b) { return { a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y }; }
float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
V3 normalize(V3 v) { const float s = 1.0f / std::sqrt(dot(v, v)); return { v.x * s, v.y * s, v.z * s }; }  // v * inversesqrt(dot)
V3 scale(V3 v, float s) { return { v.x * s, v.y * s, v.z * s }; }

M4 node_local(const m2s_json::Value& node) {  // SceneManager.cpp:224-255
    const auto& mat = node["matrix"];
    if (mat.is_array() && mat.size() == 16) {
        M4 m{};
        for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) m.c[c][r] = (float)mat[(size_t)(c * 4 + r)].number_or(0.0);
        return m;
    }
    M4 T = identity(), R = identity(), S = identity();
    const auto& t = node["translation"];
    if (t.is_array() && t.size() == 3) {
        // glm::translate(mat4(1), v): Result[3] = m[0]*v[0] + m[1]*v[1] + m[2]*v[2] + m[3], evaluated as written
        // (signed zeros included: a translation of -0 becomes +0, as in glm)
        const float v[3] = { (float)t[(size_t)0].number_or(0.0), (float)t[(size_t)1].number_or(0.0), (float)t[(size_t)2].number_or(0.0) };
        const M4 I = identity();
        for (int i = 0; i < 4; ++i) T.c[3][i] = ((I.c[0][i] * v[0] + I.c[1][i] * v[1]) + I.c[2][i] * v[2]) + I.c[3][i];
    }
    const auto& q = node["rotation"];
    if (q.is_array() && q.size() == 4) {  // glm::mat4_cast(quat(w, x, y, z))
        const float x = (float)q[(size_t)0].number_or(0), y = (float)q[(size_t)1].number_or(0), z = (float)q[(size_t)2].number_or(0),
                    w = (float)q[(size_t)3].number_or(1);
        const float qxx = x * x, qyy = y * y, qzz = z * z, qxz = x * z, qxy = x * y, qyz = y * z, qwx = w * x, qwy = w * y, qwz = w * z;
        R.c[0][0] = 1.0f - 2.0f * (qyy + qzz); R.c[0][1] = 2.0f * (qxy + qwz); R.c[0][2] = 2.0f * (qxz - qwy);
        R.c[1][0] = 2.0f * (qxy - qwz); R.c[1][1] = 1.0f - 2.0f * (qxx + qzz); R.c[1][2] = 2.0f * (qyz + qwx);
        R.c[2][0] = 2.0f * (qxz + qwy); R.c[2][1] = 2.0f * (qyz - qwx); R.c[2][2] = 1.0f - 2.0f * (qxx + qyy);
    }
    const auto& s = node["scale"];
    if (s.is_array() && s.size() == 3)   // glm::scale(mat4(1), v): Result[i] = m[i] * v[i] (whole columns: a negative
        for (int i = 0; i < 3; ++i) {      // factor turns the column's zeros into -0, which glm then carries along)
            const float f = (float)s[(size_t)i].number_or(1.0);
            for (int r = 0; r < 4; ++r) S.c[i][r] = S.c[i][r] * f;
        }
    return mul(mul(T, R), S);
}

struct Accessor {
    const uint8_t* base = nullptr;
    size_t count = 0, stride = 0;
    int component = 0, ncomp = 0;
    bool normalized = false;
};

using FileBytes = std::vector<uint8_t, DefaultInit<uint8_t>>;

struct Glb {
    m2s_json::Value doc;
    FileBytes file;
    const uint8_t* bin = nullptr;
    size_t bin_len = 0;
};

int comp_size(int t) { return t == 5120 || t == 5121 ? 1 : t == 5122 || t == 5123 ? 2 : t == 5125 || t == 5126 ? 4 : 0; }
int type_ncomp(const std::string& t) {
    return t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : t == "MAT4" ? 16 : 0;
}

bool get_accessor(const Glb& g, long long idx, Accessor& a, std::string& err) {
    const auto& acc = g.doc["accessors"][(size_t)idx];
    if (idx < 0 || !acc.is_object()) { err = "accessor index out of range"; return false; }
    const long long bvi = acc["bufferView"].int_or(-1);
    const auto& bv = g.doc["bufferViews"][(size_t)bvi];
    if (bvi < 0 || !bv.is_object()) { err = "accessor without bufferView (sparse accessors are not supported)"; return false; }
    if (bv["buffer"].int_or(0) != 0) { err = "only the GLB-embedded buffer 0 is supported"; return false; }
    a.component = (int)acc["componentType"].int_or(0);
    a.ncomp = type_ncomp(acc["type"].string_or(""));
    a.normalized = acc["normalized"].kind == m2s_json::Value::Bool && acc["normalized"].b;
    const size_t elem = (size_t)comp_size(a.component) * a.ncomp;
    if (!elem) { err = "unsupported accessor type"; return false; }
    // Every quantity below comes from the (untrusted) JSON chunk: reject negatives before any cast and bound the
    // element range without forming off + (count-1)*stride + elem, which can wrap.
    const long long count = acc["count"].int_or(0), stride = bv["byteStride"].int_or(0);
    const long long bvo = bv["byteOffset"].int_or(0), aco = acc["byteOffset"].int_or(0);
    if (acc["count"].bad_int() || bv["byteStride"].bad_int() || bv["byteOffset"].bad_int() || acc["byteOffset"].bad_int()) {
        err = "accessor field out of range"; return false;
    }
    if (count < 0 || stride < 0 || bvo < 0 || aco < 0) { err = "negative accessor field"; return false; }
    if ((unsigned long long)bvo > g.bin_len || (unsigned long long)aco > g.bin_len - (size_t)bvo) { err = "accessor exceeds the binary chunk"; return false; }
    const size_t off = (size_t)bvo + (size_t)aco;
    a.stride = stride ? (size_t)stride : elem;
    a.count = (size_t)count;
    if (a.count) {
        const size_t room = g.bin_len - off;               // bytes from the first element to the end of the chunk
        if (room < elem || (a.count - 1) > (room - elem) / a.stride) { err = "accessor exceeds the binary chunk"; return false; }
    }
    a.base = g.bin + off;
    return true;
}

// byteOffset/byteLength of a bufferView, validated against the binary chunk (both come from untrusted JSON)
static bool view_range(const m2s_json::Value& bv, size_t bin_len, size_t& off, size_t& len) {
    const long long o = bv["byteOffset"].int_or(0), l = bv["byteLength"].int_or(0);
    if (bv["byteOffset"].bad_int() || bv["byteLength"].bad_int()) return false;
    if (o < 0 || l < 0 || (unsigned long long)o > bin_len || (unsigned long long)l > bin_len - (size_t)o) return false;
    off = (size_t)o; len = (size_t)l;
    return true;
}

inline void read_floats(const Accessor& a, size_t i, int n, float* out) { std::memcpy(out, a.base + i * a.stride, sizeof(float) * n); }

void read_uv(const Accessor& a, size_t i, float out[2]) {
    const uint8_t* p = a.base + i * a.stride;
    if (a.component == 5126) { std::memcpy(out, p, 8); return; }
    if (a.component == 5121) { out[0] = p[0] / 255.0f; out[1] = p[1] / 255.0f; return; }
    if (a.component == 5123) { uint16_t v[2]; std::memcpy(v, p, 4); out[0] = v[0] / 65535.0f; out[1] = v[1] / 65535.0f; return; }
    out[0] = out[1] = 0.0f;
}

bool base64_decode(const char* s, size_t n, FileBytes& out) {
    out.clear();
    out.reserve(n / 4 * 3);
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = 0; i < n; ++i) {
        const char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else if (c == '=' || c == '\n' || c == '\r' || c == ' ') continue;
        else return false;
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}
std::string uri_decode(const std::string& u) {   // %XX escapes
    std::string r;
    for (size_t i = 0; i < u.size(); ++i) {
        auto hex = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
        if (u[i] == '%' && i + 2 < u.size() && hex(u[i + 1]) >= 0 && hex(u[i + 2]) >= 0) { r.push_back((char)(hex(u[i + 1]) * 16 + hex(u[i + 2]))); i += 2; }
        else r.push_back(u[i]);
    }
    return r;
}

}  // namespace

// Splits [0, n) into contiguous chunks, one per thread (at most 16, at least `grain` items each); f(begin, end) must only
// touch what belongs to its range.  Results do not depend on the number of threads.
static size_t host_threads() {     // hardware threads, at most 16; M2S_HOST_THREADS overrides (tests: 1 = serial reference)
    if (const char* e = std::getenv("M2S_HOST_THREADS")) { const long v = std::atol(e); if (v >= 1) return (size_t)std::min<long>(v, 64); }
    return std::min<size_t>((size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16);
}

template <class F>
static void parallel_for(size_t n, size_t grain, F f) {
    const size_t T = std::min<size_t>(host_threads(), n / std::max<size_t>(grain, 1));
    if (T <= 1) { f((size_t)0, n); return; }
    // an exception inside a worker is carried to the calling thread (first one wins) instead of std::terminate
    std::vector<std::thread> pool;
    std::exception_ptr first_error;
    std::mutex error_lock;
    auto guarded = [&](size_t b, size_t e) {
        try { f(b, e); }
        catch (...) { std::lock_guard<std::mutex> lk(error_lock); if (!first_error) first_error = std::current_exception(); }
    };
    for (size_t i = 1; i < T; ++i) pool.emplace_back([&guarded, i, n, T] { guarded(i * n / T, (i + 1) * n / T); });
    guarded((size_t)0, n / T);
    for (auto& th : pool) th.join();
    if (first_error) std::rethrow_exception(first_error);
}

// The whole file into memory: sized without a zero fill, read by several threads (each pread copies its own part out of the
// page cache).
static bool read_file(const std::string& path, FileBytes& out) {
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (::fstat(fd, &st) != 0 || st.st_size < 0) { ::close(fd); return false; }
    const size_t n = (size_t)st.st_size;
    try { out.resize(n); } catch (...) { ::close(fd); return false; }
    std::atomic<bool> ok{ true };
    parallel_for(n, (size_t)4 << 20, [&](size_t b, size_t e) {
        while (b < e) {
            const ssize_t r = ::pread(fd, out.data() + b, e - b, (off_t)b);
            if (r <= 0) { ok.store(false); return; }
            b += (size_t)r;
        }
    });
    ::close(fd);
    return ok.load();
}

bool load_glb(const std::string& path, HostScene& scene, std::string& err) {
    Glb g;
    if (!read_file(path, g.file)) { err = "cannot read " + path; return false; }
    const FileBytes& f = g.file;
    auto u32 = [&](size_t o) { uint32_t v; std::memcpy(&v, &f[o], 4); return v; };
    if (f.size() < 20 || u32(0) != 0x46546C67u) { err = "not a binary glTF (.glb) file"; return false; }
    if (u32(4) != 2) { err = "unsupported glTF container version"; return false; }
    size_t pos = 12;
    const uint8_t* json = nullptr;
    size_t json_len = 0;
    while (pos + 8 <= f.size()) {
        const uint32_t clen = u32(pos), ctype = u32(pos + 4);
        if (pos + 8 + (size_t)clen > f.size()) { err = "truncated GLB chunk"; return false; }
        if (ctype == 0x4E4F534Au && !json) { json = &f[pos + 8]; json_len = clen; }
        else if (ctype == 0x004E4942u && !g.bin) { g.bin = &f[pos + 8]; g.bin_len = clen; }
        pos += 8 + (size_t)clen;
        pos = (pos + 3) & ~(size_t)3;
    }
    if (!json) { err = "GLB without JSON chunk"; return false; }
    try { g.doc = m2s_json::parse((const char*)json, json_len); }
    catch (const std::exception& e) { err = e.what(); return false; }
    const auto& doc = g.doc;

    // ---- extensions: the reference's loader (tiny_gltf + SceneManager.cpp) knows none.  Where geometry is stored in an
    // extension's own encoding it would read garbage; refuse those files instead.  Extensions that only refine the
    // appearance are ignored, as the reference ignores them, and reported.
    {
        const auto& req = doc["extensionsRequired"];
        for (size_t i = 0; i < req.size(); ++i) {
            const std::string e = req[i].string_or("");
            if (e == "KHR_draco_mesh_compression" || e == "EXT_meshopt_compression" || e == "KHR_mesh_quantization") {
                err = "required glTF extension " + e + " is not supported (geometry is not stored as plain accessors)";
                return false;
            }
        }
        const auto& used = doc["extensionsUsed"];
        for (size_t i = 0; i < used.size(); ++i) {
            const std::string e = used[i].string_or("");
            if (!e.empty()) scene.warnings.push_back("glTF extension " + e + " is ignored (as by the reference's loader)");
        }
    }

    // ---- scene graph -> (mesh, world matrix) instances: SceneManager.cpp:211-281 ---------------------
    struct Inst { long long mesh; M4 world; };
    std::vector<Inst> insts;
    const auto& nodes = doc["nodes"];
    std::function<void(long long, const M4&, int)> walk = [&](long long ni, const M4& parent, int depth) {
        if (ni < 0 || (size_t)ni >= nodes.size() || depth > 256) return;
        const auto& node = nodes[(size_t)ni];
        const M4 world = mul(parent, node_local(node));
        const long long mi = node["mesh"].int_or(-1);
        if (mi >= 0 && (size_t)mi < doc["meshes"].size()) insts.push_back({ mi, world });
        const auto& ch = node["children"];
        for (size_t k = 0; k < ch.size(); ++k) walk(ch[k].int_or(-1), world, depth + 1);
    };
    if (doc["scenes"].size()) {
        long long si = doc["scene"].int_or(-1);
        if (si < 0 || (size_t)si >= doc["scenes"].size()) si = 0;
        const auto& roots = doc["scenes"][(size_t)si]["nodes"];
        for (size_t k = 0; k < roots.size(); ++k) walk(roots[k].int_or(-1), identity(), 0);
    }
    if (insts.empty())
        for (size_t i = 0; i < doc["meshes"].size(); ++i) insts.push_back({ (long long)i, identity() });

    // ---- images embedded in the binary chunk and referenced by a texture are decoded up front, in parallel (the decoders
    // are pure functions of their input; three 2048^2 PNGs: 370 -> 130 ms).  Results are CONSUMED below in the reference's
    // order — first use by a material — so image slots, and which error is reported first, do not depend on the threads.
    struct PreDecoded { const uint8_t* enc = nullptr; size_t len = 0; bool ok = false; Image img; std::string perr; };
    std::map<long long, PreDecoded> predecoded;
    {
        const auto& texs = doc["textures"];
        for (size_t ti = 0; ti < texs.size(); ++ti) {
            const long long src = texs[ti]["source"].int_or(-1);
            const auto& im = doc["images"][(size_t)src];
            if (src < 0 || !im.is_object() || predecoded.count(src)) continue;
            const long long bvi = im["bufferView"].int_or(-1);
            const auto& bv = doc["bufferViews"][(size_t)bvi];
            if (bvi < 0 || !bv.is_object()) continue;
            size_t off = 0, len = 0;
            if (!view_range(bv, g.bin_len, off, len)) continue;          // reported when (if) the image is used
            PreDecoded& pd = predecoded[src];
            pd.enc = g.bin + off;
            pd.len = len;
        }
    }
    // the decoders run while the geometry is de-indexed; whoever needs an image first (or the end of this function) joins them
    std::vector<PreDecoded*> decode_work;
    for (auto& kv : predecoded) decode_work.push_back(&kv.second);
    std::atomic<size_t> decode_next{ 0 };
    std::vector<std::thread> decode_pool;
    auto join_decoders = [&]() { for (auto& th : decode_pool) if (th.joinable()) th.join(); };
    struct AtExit { std::function<void()> f; ~AtExit() { f(); } } join_at_exit{ join_decoders };
    {
        auto run = [&decode_work, &decode_next]() {
            for (size_t i; (i = decode_next.fetch_add(1)) < decode_work.size();) {
                PreDecoded& pd = *decode_work[i];
                const bool is_jpeg = pd.len >= 3 && pd.enc[0] == 0xFF && pd.enc[1] == 0xD8 && pd.enc[2] == 0xFF;
                // a decoder that throws (bad_alloc on a header claiming a huge image) must not take the process down
                // from inside a worker thread: record it like any other decode failure
                try { pd.ok = is_jpeg ? decode_jpeg(pd.enc, pd.len, pd.img, pd.perr) : decode_png(pd.enc, pd.len, pd.img, pd.perr); }
                catch (const std::exception& e) { pd.ok = false; pd.perr = std::string("decoder failed: ") + e.what(); }
                catch (...) { pd.ok = false; pd.perr = "decoder failed"; }
            }
        };
        const size_t n_threads = std::min<size_t>(host_threads(), decode_work.size());
        if (host_threads() > 1) for (size_t t = 0; t < n_threads; ++t) decode_pool.emplace_back(run);
        else run();                                                      // M2S_HOST_THREADS=1: everything on this thread
    }

    // ---- images are handed out lazily, once per glTF image ---------------------------------------------
    std::map<long long, int> image_slot;
    auto load_image = [&](long long tex_index, int& out_slot) -> bool {
        out_slot = -1;
        const auto& tex = doc["textures"][(size_t)tex_index];
        if (tex_index < 0 || !tex.is_object()) return true;               // SceneManager.cpp:68-70: silently absent
        const long long src = tex["source"].int_or(-1);
        const auto& im = doc["images"][(size_t)src];
        if (src < 0 || !im.is_object()) return true;
        auto it = image_slot.find(src);
        if (it != image_slot.end()) { out_slot = it->second; return true; }
        // the encoded image: a bufferView of the binary chunk, or (like tiny_gltf) a data: URI / a file next to the .glb
        const uint8_t* enc = nullptr;
        size_t len = 0;
        FileBytes ext;
        const long long bvi = im["bufferView"].int_or(-1);
        const auto& bv = doc["bufferViews"][(size_t)bvi];
        if (bvi >= 0 && bv.is_object()) {
            size_t off = 0;
            if (!view_range(bv, g.bin_len, off, len)) { err = "image bufferView exceeds the binary chunk"; return false; }
            enc = g.bin + off;
        } else {
            const std::string uri = im["uri"].string_or("");
            if (uri.empty()) { err = "image " + std::to_string(src) + " has neither a bufferView nor a uri"; return false; }
            if (uri.compare(0, 5, "data:") == 0) {
                const size_t comma = uri.find(',');
                if (comma == std::string::npos || uri.find(";base64") == std::string::npos || uri.find(";base64") > comma) {
                    err = "image " + std::to_string(src) + ": only base64 data URIs are supported";
                    return false;
                }
                if (!base64_decode(uri.c_str() + comma + 1, uri.size() - comma - 1, ext)) { err = "image " + std::to_string(src) + ": bad base64 data"; return false; }
            } else {
                const size_t slash = path.find_last_of("/\\");
                const std::string file = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + uri_decode(uri);
                if (!read_file(file, ext)) { err = "image " + std::to_string(src) + ": cannot read " + file; return false; }
            }
            enc = ext.data();
            len = ext.size();
        }
        Image img;
        std::string perr;
        bool decoded_ok;
        join_decoders();
        auto pre = predecoded.find(src);
        if (pre != predecoded.end() && pre->second.enc == enc) {          // decoded up front
            decoded_ok = pre->second.ok;
            img = std::move(pre->second.img);
            perr = pre->second.perr;
        } else {
            // like stb_image, go by the file signature, not by the declared mimeType
            const bool is_jpeg = len >= 3 && enc[0] == 0xFF && enc[1] == 0xD8 && enc[2] == 0xFF;
            decoded_ok = is_jpeg ? decode_jpeg(enc, len, img, perr) : decode_png(enc, len, img, perr);
        }
        if (!decoded_ok) {
            err = "image " + std::to_string(src) + " (" + im["mimeType"].string_or("?") + "): " + perr;
            return false;
        }
        scene.images.push_back(std::move(img));
        out_slot = image_slot[src] = (int)scene.images.size() - 1;
        return true;
    };
    // ---- primitives -> meshes: SceneManager.cpp:283-457 ----------------------------------------------------
    int mesh_counter = 0;
    for (const Inst& inst : insts) {
        const auto& mesh = doc["meshes"][(size_t)inst.mesh];
        const M4& world = inst.world;
        const M3 world3 = upper3(world);
        const M3 normal_matrix = inverse_transpose(world3);
        const auto& prims = mesh["primitives"];
        for (size_t pi = 0; pi < prims.size(); ++pi) {
            const auto& prim = prims[pi];
            const long long mode = prim["mode"].int_or(-1);
            if (mode != 4 && mode != -1) { scene.warnings.push_back("skipping non-triangle primitive (mode=" + std::to_string(mode) + ")"); continue; }
            const auto& attrs = prim["attributes"];
            if (!attrs.has("POSITION")) { scene.warnings.push_back("primitive without POSITION skipped"); continue; }
            const std::string base = mesh["name"].string_or("").empty() ? "mesh" : mesh["name"].str;
            HostMesh hm;
            hm.name = base + "_" + std::to_string(mesh_counter++);

            Accessor pos_a;
            if (!get_accessor(g, attrs["POSITION"].int_or(-1), pos_a, err)) return false;
            if (pos_a.component != 5126 || pos_a.ncomp != 3) { err = "POSITION must be float VEC3"; return false; }
            // index list
            std::vector<uint32_t> idx;
            if (prim["indices"].is_number()) {
                Accessor ia;
                if (!get_accessor(g, prim["indices"].int_or(-1), ia, err)) return false;
                if (ia.component != 5121 && ia.component != 5123 && ia.component != 5125) {
                    scene.warnings.push_back("unsupported index component type, primitive skipped");
                    continue;
                }
                idx.resize(ia.count);
                parallel_for(ia.count, 1u << 18, [&](size_t i_begin, size_t i_end) {
                    for (size_t i = i_begin; i < i_end; ++i) {
                        const uint8_t* p = ia.base + i * ia.stride;
                        if (ia.component == 5121) idx[i] = *p;
                        else if (ia.component == 5123) { uint16_t v; std::memcpy(&v, p, 2); idx[i] = v; }
                        else { uint32_t v; std::memcpy(&v, p, 4); idx[i] = v; }
                    }
                });
            } else {
                idx.resize(pos_a.count);
                for (size_t i = 0; i < pos_a.count; ++i) idx[i] = (uint32_t)i;
            }
            if (idx.size() < 3 || idx.size() % 3 != 0) { scene.warnings.push_back("invalid index count, primitive skipped"); continue; }
            {
                std::atomic<bool> out_of_range{ false };
                parallel_for(idx.size(), 1u << 18, [&](size_t i_begin, size_t i_end) {
                    bool bad = false;
                    for (size_t i = i_begin; i < i_end; ++i) bad |= idx[i] >= pos_a.count;
                    if (bad) out_of_range.store(true);
                });
                if (out_of_range.load()) { err = "vertex index out of range in mesh " + hm.name; return false; }
            }

            Accessor nrm_a, uv_a, tan_a;
            const bool has_n = attrs.has("NORMAL"), has_uv = attrs.has("TEXCOORD_0"), has_t = attrs.has("TANGENT");
            if (has_n && (!get_accessor(g, attrs["NORMAL"].int_or(-1), nrm_a, err) || nrm_a.component != 5126 || nrm_a.ncomp != 3)) { if (err.empty()) err = "NORMAL must be float VEC3"; return false; }
            if (has_uv && (!get_accessor(g, attrs["TEXCOORD_0"].int_or(-1), uv_a, err) || uv_a.ncomp != 2)) { if (err.empty()) err = "TEXCOORD_0 must be VEC2"; return false; }
            if (has_t && (!get_accessor(g, attrs["TANGENT"].int_or(-1), tan_a, err) || tan_a.component != 5126 || tan_a.ncomp != 4)) { if (err.empty()) err = "TANGENT must be float VEC4"; return false; }
            if ((has_n && nrm_a.count < pos_a.count) || (has_uv && uv_a.count < pos_a.count) || (has_t && tan_a.count < pos_a.count)) {
                err = "attribute accessor shorter than POSITION in mesh " + hm.name; return false;
            }

            const size_t n_tri = idx.size() / 3;
            hm.vertices.resize(n_tri * 3 * 17);   // (not zero-filled: DefaultInit; every float is written below)
            // every triangle is independent and writes its own 51 floats: large meshes are de-indexed by several threads,
            // each of which also keeps the bounding box of what it wrote (min / max do not depend on the order of evaluation)
            float pmn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, pmx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            std::mutex box_lock;
            parallel_for(n_tri, 32768, [&](size_t t_begin, size_t t_end) {
            float lmn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, lmx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
            for (size_t t = t_begin; t < t_end; ++t) {
                V3 p[3], n[3];
                float uv[3][2] = { { 0, 0 }, { 0, 0 }, { 0, 0 } }, tg[3][4];
                for (int e = 0; e < 3; ++e) {
                    const uint32_t vi = idx[3 * t + e];
                    float raw[4];
                    read_floats(pos_a, vi, 3, raw);
                    p[e] = xform_point(world, { raw[0], raw[1], raw[2] });
                    if (has_uv) read_uv(uv_a, vi, uv[e]);
                    if (has_n) { read_floats(nrm_a, vi, 3, raw); n[e] = normalize(mul(normal_matrix, { raw[0], raw[1], raw[2] })); }
                }
                if (!has_n) {  // :406-413 flat face normal from the transformed positions
                    const V3 fn = normalize(cross(sub(p[1], p[0]), sub(p[2], p[0])));
                    n[0] = n[1] = n[2] = fn;
                }
                if (has_t) {
                    for (int e = 0; e < 3; ++e) {
                        float raw[4];
                        read_floats(tan_a, idx[3 * t + e], 4, raw);
                        const V3 tv = normalize(mul(world3, { raw[0], raw[1], raw[2] }));
                        tg[e][0] = tv.x; tg[e][1] = tv.y; tg[e][2] = tv.z; tg[e][3] = raw[3];
                    }
                } else {  // :421-451 tangent from the UV parametrisation, one per face
                    const V3 dp1 = sub(p[1], p[0]), dp2 = sub(p[2], p[0]);
                    const float du1 = uv[1][0] - uv[0][0], dv1 = uv[1][1] - uv[0][1], du2 = uv[2][0] - uv[0][0], dv2 = uv[2][1] - uv[0][1];
                    float det = du1 * dv2 - dv1 * du2;
                    if (std::fabs(det) < 1e-8f) det = 1.0f;
                    const float inv = 1.0f / det;
                    V3 tangent = scale(sub(scale(dp1, dv2), scale(dp2, dv1)), inv);
                    V3 bitangent = scale(sub(scale(dp2, du1), scale(dp1, du2)), inv);
                    tangent = normalize(tangent);
                    bitangent = normalize(bitangent);
                    const V3 nn = normalize(cross(dp1, dp2));
                    const float hand = dot(cross(nn, tangent), bitangent) < 0.0f ? -1.0f : 1.0f;
                    for (int e = 0; e < 3; ++e) { tg[e][0] = tangent.x; tg[e][1] = tangent.y; tg[e][2] = tangent.z; tg[e][3] = hand; }
                }
                for (int e = 0; e < 3; ++e) {  // setupMeshBuffers :483-512
                    float* v = &hm.vertices[(t * 3 + e) * 17];
                    v[0] = p[e].x; v[1] = p[e].y; v[2] = p[e].z;
                    v[3] = n[e].x; v[4] = n[e].y; v[5] = n[e].z;
                    v[6] = tg[e][0]; v[7] = tg[e][1]; v[8] = tg[e][2]; v[9] = tg[e][3];
                    v[10] = uv[e][0]; v[11] = uv[e][1];
                    v[12] = v[13] = v[14] = v[15] = v[16] = 0.0f;   // normalizedUv, scale: always 0 in the reference's VBO
                    lmn[0] = std::min(lmn[0], v[0]); lmn[1] = std::min(lmn[1], v[1]); lmn[2] = std::min(lmn[2], v[2]);
                    lmx[0] = std::max(lmx[0], v[0]); lmx[1] = std::max(lmx[1], v[1]); lmx[2] = std::max(lmx[2], v[2]);
                }
            }
            std::lock_guard<std::mutex> lk(box_lock);
            for (int k = 0; k < 3; ++k) { pmn[k] = std::min(pmn[k], lmn[k]); pmx[k] = std::max(pmx[k], lmx[k]); }
            });
            std::memcpy(hm.bbox_min, pmn, 12);   // this primitive's own box; made cumulative below
            std::memcpy(hm.bbox_max, pmx, 12);

            // material: SceneManager.cpp:99-193 (factors other than baseColorFactor are parsed but never used by the pass)
            hm.base_color[0] = hm.base_color[1] = hm.base_color[2] = hm.base_color[3] = 1.0f;
            const long long mat_i = prim["material"].int_or(-1);
            const auto& mat = doc["materials"][(size_t)mat_i];
            if (mat_i >= 0 && mat.is_object()) {
                const auto& pbr = mat["pbrMetallicRoughness"];
                const auto& bcf = pbr["baseColorFactor"];
                if (bcf.is_array() && bcf.size() == 4) for (int k = 0; k < 4; ++k) hm.base_color[k] = (float)bcf[(size_t)k].number_or(1.0);
                if (pbr["baseColorTexture"].is_object() && !load_image(pbr["baseColorTexture"]["index"].int_or(-1), hm.tex_image[0])) return false;
                if (mat["normalTexture"].is_object() && !load_image(mat["normalTexture"]["index"].int_or(-1), hm.tex_image[1])) return false;
                if (pbr["metallicRoughnessTexture"].is_object() && !load_image(pbr["metallicRoughnessTexture"]["index"].int_or(-1), hm.tex_image[2])) return false;
            }
            scene.meshes.push_back(std::move(hm));
        }
    }

    // ---- cumulative bbox (:476-477,514-527: minBB/maxBB live outside the mesh loop) + C view ----------------
    float mn[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, mx[3] = { -FLT_MAX, -FLT_MAX, -FLT_MAX };
    scene.c_meshes.clear();
    for (HostMesh& hm : scene.meshes) {   // (each mesh's own box was reduced while its vertices were written)
        for (int k = 0; k < 3; ++k) { mn[k] = std::min(mn[k], hm.bbox_min[k]); mx[k] = std::max(mx[k], hm.bbox_max[k]); }
        std::memcpy(hm.bbox_min, mn, 12);
        std::memcpy(hm.bbox_max, mx, 12);
    }
    for (HostMesh& hm : scene.meshes) {
        m2s_mesh cm{};
        cm.vertices = hm.vertices.data();
        cm.n_vertices = (uint32_t)(hm.vertices.size() / 17);
        cm.stride_floats = 17;
        std::memcpy(cm.bbox_min, hm.bbox_min, 12);
        std::memcpy(cm.bbox_max, hm.bbox_max, 12);
        std::memcpy(cm.base_color, hm.base_color, 16);
        for (int k = 0; k < 3; ++k) {
            if (hm.tex_image[k] < 0) continue;
            const Image& im = scene.images[(size_t)hm.tex_image[k]];
            cm.tex[k].rgba8 = im.rgba.data();
            cm.tex[k].width = im.width;
            cm.tex[k].height = im.height;
        }
        scene.c_meshes.push_back(cm);
    }
    return true;
}

}  // namespace m2s_host

// Test hook (not part of include/m2s.h): the loader's TRS composition and vertex transforms for one node,
// so that the glm cross-check harness (ref_glm_xform_check.cpp, test infrastructure) can compare them bit-for-bit with glm compiled from the
// reference's vendored copy.  trs = translation(3) rotation xyzw(4) scale(3); out = world(16, column-major),
// transformed point(3), transformed+normalised normal(3), transformed+normalised tangent(3).
extern "C" void m2s_debug_node_xform(const float trs[10], const float p[3], const float n[3], const float t[3], float out[25]) {
    using namespace m2s_host;
    m2s_json::Value node;
    node.kind = m2s_json::Value::Obj;
    node.obj = std::make_shared<m2s_json::Object>();
    auto arr = [](const float* v, int k) {
        m2s_json::Value a;
        a.kind = m2s_json::Value::Arr;
        a.arr = std::make_shared<m2s_json::Array>();
        for (int i = 0; i < k; ++i) { m2s_json::Value x; x.kind = m2s_json::Value::Number; x.num = v[i]; a.arr->push_back(x); }
        return a;
    };
    (*node.obj)["translation"] = arr(trs, 3);
    (*node.obj)["rotation"] = arr(trs + 3, 4);
    (*node.obj)["scale"] = arr(trs + 7, 3);
    const M4 w = mul(identity(), node_local(node));
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) out[c * 4 + r] = w.c[c][r];
    const V3 q = xform_point(w, { p[0], p[1], p[2] });
    out[16] = q.x; out[17] = q.y; out[18] = q.z;
    const V3 nn = normalize(mul(inverse_transpose(upper3(w)), { n[0], n[1], n[2] }));
    out[19] = nn.x; out[20] = nn.y; out[21] = nn.z;
    const V3 tt = normalize(mul(upper3(w), { t[0], t[1], t[2] }));
    out[22] = tt.x; out[23] = tt.y; out[24] = tt.z;
}
