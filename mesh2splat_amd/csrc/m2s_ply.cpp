// m2s_ply.cpp — binary little-endian .ply export of Gaussian records, byte-compatible with the
// reference's three writers (src/parsers/parsers.cpp: standard 431-514, PBR 232-316, compressed
// PBR 339-428; dispatch savePlyVector 631-651).  The reference issues one ofstream::write per
// field (62 per Gaussian in the standard format); here rows are encoded into a large buffer by
// a pool of threads and written with a few big fwrite calls.
#include "m2s_ply.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <future>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr float kShC0 = 0.28209479177387814f;  // params.hpp:17 SH_COEFF0

// utils.hpp:270 invSigmoid
inline float inv_sigmoid(float alpha) {
    alpha = std::clamp(alpha, 0.0f, 1.0f);
    return -std::log((1.0f / (alpha + 1e-8f)) - 1.0f);
}
// glm::clamp == min(max(x, lo), hi) with glm's comparison order (NaN passes through)
inline float glm_clamp(float x, float lo, float hi) {
    const float mx = (x < lo) ? lo : x;
    return (hi < mx) ? hi : mx;
}
// parsers.cpp:370-375
inline uint8_t to_byte(float v) { return static_cast<uint8_t>(std::round(glm_clamp(v, 0.0f, 1.0f) * 255.0f)); }

struct Format {
    size_t row_bytes;
    std::vector<std::string> props;
};

Format describe(uint32_t format) {
    Format f;
    auto fl = [&](const char* n) { f.props.push_back(std::string("property float ") + n); };
    auto u8 = [&](const char* n) { f.props.push_back(std::string("property uint8 ") + n); };
    if (format == 1) {  // parsers.cpp:240-266
        for (const char* n : { "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "metallicFactor",
                               "roughnessFactor", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2",
                               "rot_3" })
            fl(n);
        f.row_bytes = 19 * 4;
    } else if (format == 2) {  // parsers.cpp:346-365
        for (const char* n : { "x", "y", "z" }) fl(n);
        for (const char* n : { "red", "green", "blue", "opacity" }) u8(n);
        for (const char* n : { "rot_0", "rot_1", "rot_2", "rot_3", "scale_0", "scale_1", "scale_2" }) fl(n);
        for (const char* n : { "octa_nx", "octa_ny", "roughness", "metallic" }) u8(n);
        f.row_bytes = 48;
    } else {  // parsers.cpp:438-464 (also the default branch of savePlyVector)
        for (const char* n : { "x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2" }) fl(n);
        for (int i = 0; i <= 44; ++i) fl(("f_rest_" + std::to_string(i)).c_str());
        for (const char* n : { "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3" }) fl(n);
        f.row_bytes = 62 * 4;
    }
    return f;
}

inline uint8_t* put(uint8_t* p, float v) { std::memcpy(p, &v, 4); return p + 4; }

void encode_rows(const m2s_gaussian* g, size_t n, uint32_t format, float sm, uint8_t* p) {
    for (size_t i = 0; i < n; ++i, ++g) {
        if (format == 2) {
            p = put(p, g->position[0]); p = put(p, g->position[1]); p = put(p, g->position[2]);
            *p++ = to_byte(g->color[0]); *p++ = to_byte(g->color[1]); *p++ = to_byte(g->color[2]); *p++ = to_byte(g->color[3]);
            for (int k = 0; k < 4; ++k) p = put(p, g->rotation[k]);
            const float min_xy = std::min(g->scale[0], g->scale[1]);  // parsers.cpp:403
            p = put(p, std::log(g->scale[0] * sm));
            p = put(p, std::log(g->scale[1] * sm));
            p = put(p, std::log(min_xy * sm));
            // EncodeOcta, parsers.cpp:320-337 (OctWrap flips both components on a joint sign test)
            const float d = std::fabs(g->normal[0]) + std::fabs(g->normal[1]) + std::fabs(g->normal[2]) + 1e-8f;
            const float nx = g->normal[0] / d, ny = g->normal[1] / d, nz = g->normal[2] / d;
            float ex = nx, ey = ny;
            if (!(nz >= 0.0f)) {
                const float s = (nx >= 0 && ny >= 0) ? 1.0f : -1.0f;
                ex = (1.0f - std::fabs(ny)) * s;
                ey = (1.0f - std::fabs(nx)) * s;
            }
            ex = ex * 0.5f + 0.5f;
            ey = ey * 0.5f + 0.5f;
            *p++ = static_cast<uint8_t>(glm_clamp(std::round(ex * 255.0f), 0.0f, 255.0f));
            *p++ = static_cast<uint8_t>(glm_clamp(std::round(ey * 255.0f), 0.0f, 255.0f));
            *p++ = to_byte(g->pbr[1]);
            *p++ = to_byte(g->pbr[0]);
            continue;
        }
        p = put(p, g->position[0]); p = put(p, g->position[1]); p = put(p, g->position[2]);
        p = put(p, g->normal[0]); p = put(p, g->normal[1]); p = put(p, g->normal[2]);
        for (int k = 0; k < 3; ++k) p = put(p, (g->color[k] - 0.5f) / kShC0);  // utils.cpp:45-49
        if (format == 1) { p = put(p, g->pbr[0]); p = put(p, g->pbr[1]); }
        else { std::memset(p, 0, 45 * 4); p += 45 * 4; }
        p = put(p, inv_sigmoid(g->color[3]));
        for (int k = 0; k < 3; ++k) p = put(p, std::log(g->scale[k] * sm));
        for (int k = 0; k < 4; ++k) p = put(p, g->rotation[k]);  // stored (w,x,y,z)
    }
}

}  // namespace

namespace m2s_ply {

m2s_status Writer::open(const char* path, uint64_t n_total, uint32_t format, float scale_multiplier) {
    if (!path) return M2S_ERR_INVALID;
    if (format > 2) format = 0;
    f_ = std::fopen(path, "wb");
    if (!f_) return M2S_ERR_IO;
    const Format fmt = describe(format);
    format_ = format; sm_ = scale_multiplier; row_bytes_ = fmt.row_bytes; expected_ = n_total; written_ = 0; ok_ = true; cur_ = 0;
    std::string header = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(n_total) + "\n";
    for (const auto& p : fmt.props) header += p + "\n";
    header += "end_header\n";
    ok_ = std::fwrite(header.data(), 1, header.size(), f_) == header.size();
    return ok_ ? M2S_OK : M2S_ERR_IO;
}

m2s_status Writer::open_slice(const char* path, uint64_t n_total, uint32_t format, float scale_multiplier, uint64_t first_row, uint64_t n_rows) {
    if (!path || first_row > n_total || n_rows > n_total - first_row) return M2S_ERR_INVALID;
    if (format > 2) format = 0;
    const Format fmt = describe(format);
    std::string header = "ply\nformat binary_little_endian 1.0\nelement vertex " + std::to_string(n_total) + "\n";
    for (const auto& p : fmt.props) header += p + "\n";
    header += "end_header\n";
    const int fd = ::open(path, O_WRONLY | O_CREAT, 0644);
    if (fd < 0) return M2S_ERR_IO;
    format_ = format; sm_ = scale_multiplier; row_bytes_ = fmt.row_bytes; expected_ = n_rows; written_ = 0; ok_ = true; cur_ = 0;
    if (first_row == 0) {   // the header's writer also fixes the length (drops what a longer, older file held beyond it)
        const unsigned long long len = header.size() + n_total * (unsigned long long)fmt.row_bytes;
        ok_ = ::ftruncate(fd, (off_t)len) == 0;
    }
    f_ = ::fdopen(fd, "wb");
    if (!f_) { ::close(fd); return M2S_ERR_IO; }
    if (first_row == 0) ok_ = ok_ && std::fwrite(header.data(), 1, header.size(), f_) == header.size();
    else ok_ = ::fseeko(f_, (off_t)(header.size() + first_row * (unsigned long long)fmt.row_bytes), SEEK_SET) == 0;
    return ok_ ? M2S_OK : M2S_ERR_IO;
}

m2s_status Writer::append_encoded(const uint8_t* rows, size_t n_rows) {
    if (!f_) return M2S_ERR_STATE;
    if (n_rows && !rows) return M2S_ERR_INVALID;
    if (pending_.valid()) ok_ = pending_.get() && ok_;
    ok_ = ok_ && std::fwrite(rows, row_bytes_, n_rows, f_) == n_rows;
    written_ += n_rows;
    return ok_ ? M2S_OK : M2S_ERR_IO;
}

m2s_status Writer::append(const m2s_gaussian* records, size_t rows) {
    if (!f_) return M2S_ERR_STATE;
    if (rows && !records) return M2S_ERR_INVALID;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    for (size_t r0 = 0; ok_ && r0 < rows; r0 += kChunkRows) {
        const size_t n = std::min(kChunkRows, rows - r0);
        std::vector<uint8_t>& buf = buf_[cur_];
        try { if (buf.size() < n * row_bytes_) buf.resize(n * row_bytes_); } catch (...) { ok_ = false; return M2S_ERR_OOM; }
        const unsigned nt = (unsigned)std::min<size_t>(hw, (n + 4095) / 4096);
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) {
            const size_t a = n * t / nt, b = n * (t + 1) / nt;
            pool.emplace_back(encode_rows, records + r0 + a, b - a, format_, sm_, buf.data() + a * row_bytes_);
        }
        encode_rows(records + r0, n / nt, format_, sm_, buf.data());
        for (auto& th : pool) th.join();
        if (pending_.valid()) ok_ = pending_.get() && ok_;       // the OTHER buffer has reached the file
        FILE* f = f_;
        const uint8_t* data = buf.data();
        const size_t bytes = n * row_bytes_;
        pending_ = std::async(std::launch::async, [f, data, bytes] { return std::fwrite(data, 1, bytes, f) == bytes; });
        written_ += n;
        cur_ ^= 1;
    }
    return ok_ ? M2S_OK : M2S_ERR_IO;
}

m2s_status Writer::close() {
    if (!f_) return ok_ ? M2S_OK : M2S_ERR_IO;
    if (pending_.valid()) ok_ = pending_.get() && ok_;
    ok_ = (std::fclose(f_) == 0) && ok_;
    f_ = nullptr;
    if (written_ != expected_) ok_ = false;
    return ok_ ? M2S_OK : M2S_ERR_IO;
}

}  // namespace m2s_ply

extern "C" m2s_status m2s_write_ply_slice(const char* path, const m2s_gaussian* records, uint64_t n, uint32_t format, float scale_multiplier,
                                          uint64_t first_row, uint64_t total_rows) {
    if (!path || (n && !records)) return M2S_ERR_INVALID;
    m2s_ply::Writer w;
    m2s_status s = w.open_slice(path, total_rows, format, scale_multiplier, first_row, n);
    if (s == M2S_OK) s = w.append(records, (size_t)n);
    const m2s_status c = w.close();
    return s != M2S_OK ? s : c;
}

extern "C" m2s_status m2s_write_ply(const char* path, const m2s_gaussian* records, uint64_t n, uint32_t format,
                                    float scale_multiplier) {
    if (!path || (n && !records)) return M2S_ERR_INVALID;
    m2s_ply::Writer w;
    m2s_status s = w.open(path, n, format, scale_multiplier);
    if (s == M2S_OK) s = w.append(records, (size_t)n);
    const m2s_status c = w.close();
    return s != M2S_OK ? s : c;
}
