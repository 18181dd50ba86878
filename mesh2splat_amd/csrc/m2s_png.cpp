// m2s_png.cpp — PNG -> RGBA8 decoder for textures embedded in .glb files (the reference decodes them
// with stb_image through tiny_gltf, always expanding to 4 components: tiny_gltf.h:2609).  zlib does the
// inflate; filtering, de-interlacing (Adam7) and colour-type expansion are done here.
// Supported: bit depths 1/2/4/8/16, colour types 0 (grey), 2 (RGB), 3 (palette + tRNS), 4 (grey+alpha),
// 6 (RGBA).  16-bit samples keep their high byte (as stb_image does for 8-bit requests).
#include "m2s_host.h"

#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace m2s_host {

namespace {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline int paeth(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// un-filter `h` scanlines of `stride` bytes (each preceded by its filter byte) into out.  One loop per filter type and row
// (PNG spec 9.2), the first pixel of a row — which has no left neighbour — apart; the row above the first one is all zero.
bool unfilter(const uint8_t* in, size_t in_len, uint8_t* out, uint32_t h, size_t stride, int bpp_bytes) {
    if (in_len < (stride + 1) * (size_t)h) return false;
    const size_t bpp = (size_t)bpp_bytes, head = std::min(bpp, stride);
    std::vector<uint8_t> zero_row(stride, 0);
    const uint8_t* prev = zero_row.data();
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t ft = in[0];
        const uint8_t* src = in + 1;
        uint8_t* dst = out + (size_t)y * stride;
        switch (ft) {
            case 0: std::memcpy(dst, src, stride); break;
            case 1:
                for (size_t i = 0; i < head; ++i) dst[i] = src[i];
                for (size_t i = bpp; i < stride; ++i) dst[i] = (uint8_t)(src[i] + dst[i - bpp]);
                break;
            case 2:
                for (size_t i = 0; i < stride; ++i) dst[i] = (uint8_t)(src[i] + prev[i]);
                break;
            case 3:
                for (size_t i = 0; i < head; ++i) dst[i] = (uint8_t)(src[i] + (prev[i] >> 1));
                for (size_t i = bpp; i < stride; ++i) dst[i] = (uint8_t)(src[i] + ((dst[i - bpp] + prev[i]) >> 1));
                break;
            case 4:
                for (size_t i = 0; i < head; ++i) dst[i] = (uint8_t)(src[i] + prev[i]);   // paeth(0, b, 0) = b
                for (size_t i = bpp; i < stride; ++i) dst[i] = (uint8_t)(src[i] + paeth(dst[i - bpp], prev[i], prev[i - bpp]));
                break;
            default: return false;
        }
        prev = dst;
        in += stride + 1;
    }
    return true;
}
}  // namespace

bool decode_png(const uint8_t* data, size_t len, Image& img, std::string& err) {
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    if (len < 8 || memcmp(data, sig, 8) != 0) { err = "not a PNG"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    bool have_ihdr = false, done = false;
    while (!done && pos + 12 <= len) {
        const uint32_t clen = be32(data + pos);
        const uint8_t* type = data + pos + 4;
        const uint8_t* body = data + pos + 8;
        if (clen > len || pos + 12 + (size_t)clen > len) { err = "truncated PNG chunk"; return false; }
        if (!memcmp(type, "IHDR", 4)) {
            if (clen < 13) { err = "bad IHDR"; return false; }
            w = be32(body); h = be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
            if (body[10] != 0 || body[11] != 0) { err = "unsupported PNG compression/filter method"; return false; }
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + clen);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + clen);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + clen);
        else if (!memcmp(type, "IEND", 4)) done = true;
        pos += 12 + (size_t)clen;
    }
    if (!have_ihdr || w == 0 || h == 0 || w > 32768 || h > 32768) { err = "bad PNG header"; return false; }
    if ((uint64_t)w * h > (1ull << 30)) { err = "PNG too large"; return false; }   // same pixel cap as the JPEG path
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "unsupported PNG colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        err = "unsupported PNG bit depth"; return false;
    }
    if (ctype == 3 && plte.size() < 3) { err = "palette PNG without PLTE"; return false; }
    const int bits_pp = depth * channels;
    const int bpp_bytes = (bits_pp + 7) / 8;

    // inflate
    auto pass_bytes = [&](uint32_t pw, uint32_t ph) -> size_t { return pw && ph ? ((size_t)pw * bits_pp + 7) / 8 * ph + ph : 0; };
    static const int xs[7] = { 0, 4, 0, 2, 0, 1, 0 }, ys[7] = { 0, 0, 4, 0, 2, 0, 1 };
    static const int dx[7] = { 8, 8, 4, 4, 2, 2, 1 }, dy[7] = { 8, 8, 8, 4, 4, 2, 2 };
    auto pass_w = [&](int p) -> uint32_t { return w > (uint32_t)xs[p] ? (w - xs[p] + dx[p] - 1) / dx[p] : 0u; };
    auto pass_h = [&](int p) -> uint32_t { return h > (uint32_t)ys[p] ? (h - ys[p] + dy[p] - 1) / dy[p] : 0u; };
    size_t raw_len = 0;
    if (interlace == 0) raw_len = pass_bytes(w, h);
    else if (interlace == 1) {
        for (int p = 0; p < 7; ++p) raw_len += pass_bytes(pass_w(p), pass_h(p));
    } else { err = "unsupported PNG interlace method"; return false; }
    // Streamed inflate: the image needs exactly raw_len bytes; what the zlib stream holds beyond them (trailing bytes some
    // encoders leave, which stb_image ignores) is not an error, a stream that ends early is.
    // (deflate expands by at most 1032 : 1: a header must not be able to demand gigabytes that the IDAT bytes cannot fill)
    if (raw_len / 1032 > idat.size() + 1) { err = "PNG data too short for its dimensions"; return false; }
    std::vector<uint8_t, DefaultInit<uint8_t>> raw(raw_len);   // (filled by inflate; no zero fill)
    {
        z_stream zs;
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) { err = "PNG inflate failed"; return false; }
        size_t in_pos = 0;
        int zr = Z_OK;
        while (zs.total_out < raw_len && zr == Z_OK) {
            if (zs.avail_in == 0) {
                const size_t take = std::min<size_t>(idat.size() - in_pos, (size_t)1 << 30);
                if (!take) break;
                zs.next_in = idat.data() + in_pos; zs.avail_in = (uInt)take; in_pos += take;
            }
            zs.next_out = raw.data() + zs.total_out;
            zs.avail_out = (uInt)std::min<size_t>(raw_len - zs.total_out, (size_t)1 << 30);
            zr = inflate(&zs, Z_NO_FLUSH);
        }
        const bool complete = zs.total_out == raw_len && (zr == Z_OK || zr == Z_STREAM_END);
        inflateEnd(&zs);
        if (!complete) { err = "PNG inflate failed"; return false; }
    }

    img.width = w; img.height = h;
    // The common kinds — 8 bits per sample, not interlaced, no tRNS — without the per-pixel dispatch below: RGBA is un-filtered
    // straight into the image; RGB, grey and grey + alpha are un-filtered into a line buffer and widened row by row.
    if (interlace == 0 && depth == 8 && ctype != 3 && trns.empty()) {
        const size_t stride = (size_t)w * channels;
        img.rgba.resize((size_t)w * h * 4);
        if (ctype == 6) {
            if (!unfilter(raw.data(), raw.size(), img.rgba.data(), h, stride, bpp_bytes)) { err = "bad PNG filter"; return false; }
            return true;
        }
        std::vector<uint8_t, DefaultInit<uint8_t>> lines(stride * h);
        if (!unfilter(raw.data(), raw.size(), lines.data(), h, stride, bpp_bytes)) { err = "bad PNG filter"; return false; }
        const uint8_t* in = lines.data();
        uint8_t* o = img.rgba.data();
        const size_t n = (size_t)w * h;
        if (ctype == 2) for (size_t i = 0; i < n; ++i, in += 3, o += 4) { o[0] = in[0]; o[1] = in[1]; o[2] = in[2]; o[3] = 255; }
        else if (ctype == 0) for (size_t i = 0; i < n; ++i, in += 1, o += 4) { o[0] = o[1] = o[2] = in[0]; o[3] = 255; }
        else for (size_t i = 0; i < n; ++i, in += 2, o += 4) { o[0] = o[1] = o[2] = in[0]; o[3] = in[1]; }   // ctype 4
        return true;
    }
    img.rgba.assign((size_t)w * h * 4, 255);

    auto put_pixel = [&](const uint8_t* line, uint32_t x_in_line, uint32_t X, uint32_t Y) {
        uint8_t* o = &img.rgba[((size_t)Y * w + X) * 4];
        auto sample = [&](int ch) -> int {   // returns raw sample value at `depth` bits
            if (depth == 8) return line[(size_t)x_in_line * channels + ch];
            if (depth == 16) return line[((size_t)x_in_line * channels + ch) * 2];   // high byte
            const size_t bit = (size_t)x_in_line * depth;
            return (line[bit >> 3] >> (8 - depth - (bit & 7))) & ((1 << depth) - 1);
        };
        if (ctype == 3) {
            const int idx = sample(0);
            if ((size_t)idx * 3 + 2 < plte.size()) { o[0] = plte[idx * 3]; o[1] = plte[idx * 3 + 1]; o[2] = plte[idx * 3 + 2]; }
            else { o[0] = o[1] = o[2] = 0; }
            o[3] = (size_t)idx < trns.size() ? trns[idx] : 255;
        } else if (ctype == 0 || ctype == 4) {
            int g = sample(0);
            if (depth < 8) g = g * 255 / ((1 << depth) - 1);
            o[0] = o[1] = o[2] = (uint8_t)g;
            o[3] = ctype == 4 ? (uint8_t)sample(1) : 255;
            if (ctype == 0 && trns.size() >= 2) {   // single transparent grey value
                const int key = depth == 16 ? trns[0] : trns[1];
                const int raw_g = depth < 8 ? sample(0) : g;
                if (raw_g == key) o[3] = 0;
            }
        } else {
            o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2);
            o[3] = ctype == 6 ? (uint8_t)sample(3) : 255;
            if (ctype == 2 && trns.size() >= 6) {
                const int kr = depth == 16 ? trns[0] : trns[1], kg = depth == 16 ? trns[2] : trns[3], kb = depth == 16 ? trns[4] : trns[5];
                if (o[0] == kr && o[1] == kg && o[2] == kb) o[3] = 0;
            }
        }
    };

    if (interlace == 0) {
        const size_t stride = ((size_t)w * bits_pp + 7) / 8;
        std::vector<uint8_t> lines(stride * h);
        if (!unfilter(raw.data(), raw.size(), lines.data(), h, stride, bpp_bytes)) { err = "bad PNG filter"; return false; }
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) put_pixel(&lines[(size_t)y * stride], x, x, y);
    } else {
        size_t off = 0;
        for (int p = 0; p < 7; ++p) {
            const uint32_t pw = pass_w(p), ph = pass_h(p);
            if (!pw || !ph) continue;
            const size_t stride = ((size_t)pw * bits_pp + 7) / 8;
            std::vector<uint8_t> lines(stride * ph);
            if (!unfilter(raw.data() + off, raw.size() - off, lines.data(), ph, stride, bpp_bytes)) { err = "bad PNG filter"; return false; }
            off += (stride + 1) * ph;
            for (uint32_t y = 0; y < ph; ++y)
                for (uint32_t x = 0; x < pw; ++x) put_pixel(&lines[(size_t)y * stride], x, xs[p] + x * dx[p], ys[p] + y * dy[p]);
        }
    }
    return true;
}

}  // namespace m2s_host
