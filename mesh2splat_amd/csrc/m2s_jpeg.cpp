// m2s_jpeg.cpp — JPEG decoder of the scene loader (host code).
//
// The reference decodes glTF images with tiny_gltf -> stb_image 2.29 (vendored under thirdParty/, not copied here).
// Entropy decoding is fixed by ITU-T T.81; what is implementation-defined — the inverse DCT, chroma upsampling and
// the YCbCr -> RGB conversion — follows stb_image's published integer algorithms so that decoded texels are
// bit-identical to what the reference uploads to GL (checked against the reference's own loader, compiled from where
// it lies: tests/test_ref_host.py::test_jpeg_textures_match_reference_*):
//   * IDCT: the IJG "jidctint" (Loeffler-Ligtenberg-Moschytz) factorisation with 12-bit constants, column pass kept at
//     +2 fractional bits (rounding 512, >> 10), row pass rounding 65536 + (128 << 17), >> 17, clamped to [0,255];
//   * upsampling: "fancy" triangle filters for the 2x1, 1x2 and 2x2 cases ((3a+b+2)>>2 and (3(3a+b)+(3c+d)+8)>>4),
//     sample replication for every other ratio;
//   * colour: 20-bit fixed point (y << 20) + (1 << 19) with the constants 1.40200, 0.71414, 0.34414, 1.77200 rounded
//     at 12 bits, the Cb contribution to green truncated to its upper 16 bits.
// ATTRIBUTION.  The three arithmetic recipes above are not ours: the fixed-point constants, rounding terms and shift
// amounts are those of stb_image.h v2.29 (Sean Barrett et al., public domain / MIT; stbi__idct_block, stbi__resample_row_*,
// stbi__YCbCr_to_RGB_row), whose inverse DCT in turn derives from the Independent JPEG Group's jidctint.c (Loeffler,
// Ligtenberg, Moschytz; IJG licence: "this software is based in part on the work of the Independent JPEG Group").  They
// are reproduced because the decoded bytes must equal the reference loader's; the code that evaluates them below is
// written from the formulas (matrix form of the transform, one planar up-sampler, a per-pixel colour function), not
// transcribed from stb_image.
// ENTROPY STAGE (round 3: rewritten; round 2's version restated stb_image's decode functions).  Written from ITU-T T.81:
// a bit sequence over the byte-stuffed segment with a 64-bit accumulator (B.1.1.5, F.2.2.5), canonical Huffman tables decoded
// by a left-aligned bound search behind a 256-entry first-byte table (Annex C, F.2.2.3), the sequential block procedure
// (F.2.2.1-F.2.2.2), the progressive DC / AC first / AC refinement procedures (G.1.2.1-G.1.2.3, Figure G.7) and one generic
// loop over the scan's units with restart intervals (E.2.4).  On VALID files the result is the standard's, hence the
// reference loader's, bit for bit (tests/test_ref_host.py, tools/diff_decoders.py); on corrupt data this decoder stops at
// a coefficient index beyond 63 where stb_image clamps it.
// Supported: baseline / extended sequential (SOF0, SOF1) and progressive (SOF2) Huffman JPEG, 8 bits per sample,
// 1 or 3 components, sampling factors 1..4, restart intervals, Adobe APP14 / component-id RGB detection.
// Not supported (explicit error): arithmetic coding, lossless, 12-bit, 4-component (CMYK / YCCK) files.
#include <algorithm>
#include <cstring>
#include <vector>

#include "m2s_host.h"

namespace m2s_host {
namespace {

// zig-zag sequence of T.81 Figure A.6: position in the 8x8 block (row-major) of the k-th coefficient of the sequence
const uint8_t kZigzag[64] = { 0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                              6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                              39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// ---- the entropy-coded segment as a bit sequence (T.81 B.1.1.5 byte stuffing, F.2.2.5 NEXTBIT) ----------------------------
// A 64-bit accumulator holds the bits not yet consumed, most significant first.  0xFF 0x00 is the data byte 0xFF; 0xFF followed
// by anything else (after optional 0xFF fill bytes) is a marker: it ends the segment, is remembered, and from then on — as past
// the end of the file — the sequence continues with zero bits, so that a decoder asking for more than the segment holds gets
// zeros instead of an error (what the reference's loader does too; valid files never get there).
class SegmentBits {
public:
    SegmentBits(const uint8_t*& cursor, const uint8_t* end) : cur_(cursor), end_(end) {}
    void restart() { acc_ = 0; have_ = 0; marker_ = 0; }       // byte-aligned restart: pending (padding) bits are dropped
    int marker() const { return marker_; }                      // 0: still inside the segment
    void clear_marker() { marker_ = 0; }
    void fill() {
        while (have_ <= 56) {
            unsigned byte = 0;
            if (!marker_ && cur_ < end_) {
                byte = *cur_++;
                if (byte == 0xFF) {
                    unsigned next = cur_ < end_ ? *cur_++ : 0u;
                    while (next == 0xFF) next = cur_ < end_ ? *cur_++ : 0u;
                    if (next != 0) { marker_ = (int)next; byte = 0; }
                }
            }
            acc_ |= (uint64_t)byte << (56 - have_);
            have_ += 8;
        }
    }
    unsigned peek16() { if (have_ < 16) fill(); return (unsigned)(acc_ >> 48); }
    void drop(int n) { acc_ <<= n; have_ -= n; }
    unsigned bits(int n) {                                      // RECEIVE(n), 0 <= n <= 16
        if (n == 0) return 0;
        if (have_ < n) fill();
        const unsigned v = (unsigned)(acc_ >> (64 - n));
        drop(n);
        return v;
    }
    unsigned bit() { return bits(1); }
    int receive_extend(int ssss) {                              // F.2.2.1: RECEIVE then EXTEND (Figure F.12)
        if (ssss == 0) return 0;
        const int v = (int)bits(ssss);
        return v < (1 << (ssss - 1)) ? v - (1 << ssss) + 1 : v;
    }
private:
    const uint8_t*& cur_;
    const uint8_t* end_;
    uint64_t acc_ = 0;
    int have_ = 0;
    int marker_ = 0;
};

// ---- Huffman table (T.81 Annex C: code lengths -> codes; F.2.2.3 DECODE) ------------------------------------------------
// Canonical codes: the codes of length l are first[l], first[l] + 1, ... in the order of HUFFVAL, and first[l + 1] =
// (first[l] + count[l]) * 2.  Left-aligned to 16 bits, "the codes of length <= l" are exactly the values below
// bound[l] = (first[l] + count[l]) << (16 - l), which grows with l: DECODE is "the smallest l with peek16 < bound[l]".  The 256
// possible leading bytes are resolved through a table (codes of <= 8 bits: length and symbol at once).
struct HuffTable {
    uint16_t quick[256] = {};      // (length << 8) | symbol for a leading byte that starts with a code of <= 8 bits, else 0
    uint32_t bound[18] = {};       // exclusive upper bound of the left-aligned codes of length <= l
    int base[17] = {};             // index into symbols of code 0 of length l (may be negative: only the sum with a code is used)
    uint8_t symbols[256] = {};
    bool present = false;
    bool define(const uint8_t counts[16], const uint8_t* vals, int n_vals) {
        std::memset(quick, 0, sizeof quick);
        uint32_t code = 0;
        int index = 0;
        for (int l = 1; l <= 16; ++l) {
            const int n = counts[l - 1];
            if (code + (uint32_t)n > (1u << l) || index + n > n_vals) return false;   // more codes than a prefix code of this length can have
            base[l] = index - (int)code;
            if (l <= 8)
                for (int i = 0; i < n; ++i) {
                    const unsigned lo = (code + (unsigned)i) << (8 - l);
                    for (unsigned f = 0; f < (1u << (8 - l)); ++f) quick[lo + f] = (uint16_t)((l << 8) | vals[index + i]);
                }
            code += (uint32_t)n;
            index += n;
            bound[l] = code << (16 - l);
            code <<= 1;
        }
        bound[17] = 0xFFFFFFFFu;
        if (index != n_vals || index > 256) return false;
        std::memcpy(symbols, vals, (size_t)n_vals);
        present = true;
        return true;
    }
    int decode(SegmentBits& in) const {                         // -1: no code matches (corrupt data)
        const unsigned v = in.peek16();
        if (const unsigned q = quick[v >> 8]) { in.drop((int)(q >> 8)); return (int)(q & 255u); }
        for (int l = 9; l <= 16; ++l)
            if (v < bound[l]) {
                in.drop(l);
                return symbols[base[l] + (int)(v >> (16 - l))];
            }
        return -1;
    }
};

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int pred = 0;                // PRED of F.2.2.1: the previous DC value of this component
    int x = 0, y = 0;            // size in samples
    int pw = 0, ph = 0;          // plane size: padded to whole MCUs
    int bw = 0, bh = 0;          // plane size in blocks
    std::vector<uint8_t> data;   // decoded samples, pw x ph
    std::vector<short> coeff;    // progressive: 64 coefficients per block
};

// parameters of the scan being decoded (T.81 B.2.3: Ss, Se, Ah, Al) and its running state
struct Scan {
    int n = 0, order[4] = {};
    int Ss = 0, Se = 63, Ah = 0, Al = 0;
    unsigned eobrun = 0;         // G.1.2.2: blocks still covered by the last end-of-band run
    int until_restart = 0;       // MCUs left in the current restart interval
};

// ---- block decoding procedures ----------------------------------------------------------------------------------------------
// Sequential DCT (F.2.2): DC difference, then (run, size) pairs until the block is full or EOB.  Coefficients are dequantised as
// they are stored (the inverse DCT follows at once).
inline const char* decode_sequential_block(SegmentBits& in, const HuffTable& dc, const HuffTable& ac, const uint16_t dq[64], int& pred, short blk[64]) {
    std::memset(blk, 0, 64 * sizeof(short));
    const int t = dc.decode(in);
    if (t < 0 || t > 15) return "bad JPEG Huffman code";
    pred = (int)((unsigned)pred + (unsigned)in.receive_extend(t));          // (wraps instead of overflowing on corrupt streams)
    blk[0] = (short)((unsigned)pred * dq[0]);
    for (int k = 1; k < 64; ++k) {
        const int rs = ac.decode(in);
        if (rs < 0) return "bad JPEG Huffman code";
        const int ssss = rs & 15, rrrr = rs >> 4;
        if (ssss == 0) {
            if (rrrr != 15) break;                                           // EOB
            k += 15;                                                         // ZRL: sixteen zeros
            continue;
        }
        k += rrrr;
        if (k > 63) return "corrupt JPEG: coefficient index beyond the block";
        blk[kZigzag[k]] = (short)(in.receive_extend(ssss) * dq[kZigzag[k]]);
    }
    return nullptr;
}
// Progressive, DC (G.1.2.1): first scan = the sequential DC procedure with the point transform Al; a refinement scan adds one bit.
inline const char* decode_dc_progressive(SegmentBits& in, const HuffTable& dc, const Scan& sc, int& pred, short blk[64]) {
    if (sc.Ah == 0) {
        std::memset(blk, 0, 64 * sizeof(short));
        const int t = dc.decode(in);
        if (t < 0 || t > 15) return "bad JPEG Huffman code";
        pred = (int)((unsigned)pred + (unsigned)in.receive_extend(t));
        blk[0] = (short)((unsigned)pred << sc.Al);
    } else if (in.bit()) {
        blk[0] = (short)(blk[0] + (1 << sc.Al));
    }
    return nullptr;
}
// Progressive, AC, first scan of a band (G.1.2.2): as sequential AC within [Ss, Se], plus EOBn = end of band for 2^n + bits blocks.
inline const char* decode_ac_first(SegmentBits& in, const HuffTable& ac, Scan& sc, short blk[64]) {
    if (sc.eobrun) { --sc.eobrun; return nullptr; }
    for (int k = sc.Ss; k <= sc.Se; ++k) {
        const int rs = ac.decode(in);
        if (rs < 0) return "bad JPEG Huffman code";
        const int ssss = rs & 15, rrrr = rs >> 4;
        if (ssss == 0) {
            if (rrrr == 15) { k += 15; continue; }
            sc.eobrun = (1u << rrrr) + in.bits(rrrr) - 1u;                  // this block is the first of the run
            break;
        }
        k += rrrr;
        if (k > 63) return "corrupt JPEG: coefficient index beyond the block";
        blk[kZigzag[k]] = (short)(in.receive_extend(ssss) * (1 << sc.Al));
    }
    return nullptr;
}
// Progressive, AC, refinement (G.1.2.3, Figure G.7).  Coefficients with a non-zero history receive one correction bit each as
// they are passed; a (run, 1) code places a new +-2^Al coefficient after `run` ZERO-history coefficients; EOBn ends the band for
// this and the following blocks, whose non-zero coefficients still get their correction bits.
inline const char* decode_ac_refine(SegmentBits& in, const HuffTable& ac, Scan& sc, short blk[64]) {
    const short plus = (short)(1 << sc.Al), minus = (short)-plus;
    auto correct = [&](short& c) {                                           // one correction bit for a non-zero coefficient
        if (in.bit() && (c & plus) == 0) c = (short)(c > 0 ? c + plus : c + minus);
    };
    int k = sc.Ss;
    if (sc.eobrun == 0) {
        while (k <= sc.Se) {
            const int rs = ac.decode(in);
            if (rs < 0) return "bad JPEG Huffman code";
            const int ssss = rs & 15;
            int zeros = rs >> 4;
            short fresh = 0;
            if (ssss == 0) {
                if (zeros != 15) { sc.eobrun = (1u << zeros) + in.bits(zeros); break; }   // EOBn: the rest of this block is handled below
            } else {
                if (ssss != 1) return "bad JPEG Huffman code";
                fresh = in.bit() ? plus : minus;
            }
            for (; k <= sc.Se; ++k) {
                short& c = blk[kZigzag[k]];
                if (c != 0) { correct(c); continue; }
                if (zeros == 0) { c = fresh; ++k; break; }                   // (ZRL: fresh == 0, the sixteenth zero is passed)
                --zeros;
            }
        }
    }
    if (sc.eobrun) {
        for (; k <= sc.Se; ++k)
            if (short& c = blk[kZigzag[k]]; c != 0) correct(c);
        --sc.eobrun;
    }
    return nullptr;
}

struct Decoder {
    const uint8_t* p;
    const uint8_t* end;
    std::string* err;
    // frame
    bool progressive = false;
    int img_x = 0, img_y = 0, n_comp = 0;
    int h_max = 1, v_max = 1, mcu_w = 0, mcu_h = 0, mcu_x = 0, mcu_y = 0;
    Component comp[4];
    uint16_t dequant[4][64] = {};
    bool have_q[4] = {};
    HuffTable dc[4], ac[4];
    int restart_interval = 0;
    bool has_jfif = false;       // an APP0 "JFIF" segment was seen
    int adobe_transform = -1;    // colour transform flag of an APP14 "Adobe" segment, -1: none
    int rgb_ids = 0;
    Scan scan;
    int pending_marker = 0;      // marker that ended the last entropy-coded segment (0: none)

    bool fail(const char* m) { *err = m; return false; }
    int get8() { return p < end ? *p++ : 0; }
    int get16() { const int a = get8(); return (a << 8) | get8(); }

    // ---- inverse DCT (see the header comment) ------------------------------------------------------------------
    // The 8-point transform as two 4x4 integer matrices (even and odd inputs): every entry is a sum of the 12-bit LLM
    // constants, so in integer arithmetic the products below equal the butterfly network's results exactly (64-bit
    // accumulators: no overflow for any coefficient a corrupt stream can hold).  out[k] = even[k] + odd[k],
    // out[7 - k] = even[k] - odd[k].
    static inline uint8_t clamp8(long long x) { return x < 0 ? 0 : x > 255 ? 255 : (uint8_t)x; }
    struct IdctMatrices {
        long long even[4][4];   // rows: outputs 0..3, columns: inputs 0, 2, 4, 6
        long long odd[4][4];    // rows: outputs 0..3, columns: inputs 1, 3, 5, 7
        IdctMatrices() {
            auto fx = [](double x) { return (long long)(int)(x * 4096 + 0.5); };
            const long long A = fx(0.5411961), B = fx(-1.847759065), C = fx(0.765366865), one = 4096;
            const long long k1175 = fx(1.175875602), k0298 = fx(0.298631336), k2053 = fx(2.053119869), k3072 = fx(3.072711026),
                            k1501 = fx(1.501321110), m0899 = fx(-0.899976223), m2562 = fx(-2.562915447), m1961 = fx(-1.961570560),
                            m0390 = fx(-0.390180644);
            const long long r2 = A + C, r6 = A, q2 = A, q6 = A + B;            // the two rotated even terms
            const long long e[4][4] = { { one, r2, one, r6 }, { one, q2, -one, q6 }, { one, -q2, -one, -q6 }, { one, -r2, one, -r6 } };
            const long long o[4][4] = { { k1501 + k1175 + m0899 + m0390, k1175, k1175 + m0390, k1175 + m0899 },
                                        { k1175, k3072 + k1175 + m2562 + m1961, k1175 + m2562, k1175 + m1961 },
                                        { k1175 + m0390, k1175 + m2562, k2053 + k1175 + m2562 + m0390, k1175 },
                                        { k1175 + m0899, k1175 + m1961, k1175, k0298 + k1175 + m0899 + m1961 } };
            memcpy(even, e, sizeof even);
            memcpy(odd, o, sizeof odd);
        }
    };
    // one 8-point pass over in[0], in[step], ...: writes (value + bias) >> shift through `store`
    template <class In, class Store>
    static inline void idct_pass(const IdctMatrices& M, In in, long long bias, int shift, Store store) {
        const long long ev[4] = { in(0), in(2), in(4), in(6) }, od[4] = { in(1), in(3), in(5), in(7) };
        for (int k = 0; k < 4; ++k) {
            const long long e = M.even[k][0] * ev[0] + M.even[k][1] * ev[1] + M.even[k][2] * ev[2] + M.even[k][3] * ev[3] + bias;
            const long long o = M.odd[k][0] * od[0] + M.odd[k][1] * od[1] + M.odd[k][2] * od[2] + M.odd[k][3] * od[3];
            store(k, (e + o) >> shift);
            store(7 - k, (e - o) >> shift);
        }
    }
    static void idct(uint8_t* out, int stride, const short d[64]) {
        static const IdctMatrices M;
        long long mid[64];
        for (int c = 0; c < 8; ++c) {   // columns; two extra fractional bits are kept (bias 512, >> 10)
            const short* s = d + c;
            bool ac = false;
            for (int r = 1; r < 8; ++r) ac = ac || s[8 * r] != 0;
            if (!ac) { for (int r = 0; r < 8; ++r) mid[8 * r + c] = (long long)s[0] * 4; continue; }   // DC only: the constant column
            idct_pass(M, [&](int r) { return (long long)s[8 * r]; }, 512, 10, [&](int r, long long v) { mid[8 * r + c] = v; });
        }
        for (int r = 0; r < 8; ++r) {   // rows; the level shift (+128) is folded into the rounding term
            const long long* v = mid + 8 * r;
            uint8_t* o = out + (size_t)r * stride;
            idct_pass(M, [&](int c) { return v[c]; }, 65536 + (128 << 17), 17, [&](int c, long long x) { o[c] = clamp8(x); });
        }
    }

    // ---- markers ---------------------------------------------------------------------------------------------------
    bool process_marker(int m) {
        switch (m) {
        case 0xDD: {
            if (get16() != 4) return fail("bad DRI length");
            restart_interval = get16();
            return true;
        }
        case 0xDB: {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), prec = q >> 4, t = q & 15;
                if (prec > 1 || t > 3) return fail("bad DQT");
                for (int i = 0; i < 64; ++i) dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
                have_q[t] = true;
                L -= prec ? 129 : 65;
            }
            return L == 0 ? true : fail("bad DQT length");
        }
        case 0xC4: {
            int L = get16() - 2;
            while (L > 0) {
                const int q = get8(), tc = q >> 4, th = q & 15;
                if (tc > 1 || th > 3) return fail("bad DHT");
                uint8_t counts[16];
                int n = 0;
                for (int i = 0; i < 16; ++i) { counts[i] = (uint8_t)get8(); n += counts[i]; }
                if (n > 256) return fail("bad DHT");
                uint8_t vals[256];
                for (int i = 0; i < n; ++i) vals[i] = (uint8_t)get8();
                if (!(tc ? ac[th] : dc[th]).define(counts, vals, n)) return fail("bad DHT code lengths");
                L -= 17 + n;
            }
            return L == 0 ? true : fail("bad DHT length");
        }
        default: break;
        }
        if ((m >= 0xE0 && m <= 0xEF) || m == 0xFE) {
            int L = get16();
            if (L < 2) return fail("bad APP/COM length");
            L -= 2;
            if (m == 0xE0 && L >= 5) {
                static const char tag[5] = { 'J', 'F', 'I', 'F', 0 };
                bool ok = true;
                for (int i = 0; i < 5; ++i) if (get8() != tag[i]) ok = false;
                L -= 5;
                if (ok) has_jfif = true;
            } else if (m == 0xEE && L >= 12) {
                static const char tag[6] = { 'A', 'd', 'o', 'b', 'e', 0 };
                bool ok = true;
                for (int i = 0; i < 6; ++i) if (get8() != tag[i]) ok = false;
                L -= 6;
                if (ok) {
                    get8(); get16(); get16();      // version, flags0, flags1
                    adobe_transform = get8();
                    L -= 6;
                }
            }
            if (p + L > end) return fail("truncated JPEG");
            p += L;
            return true;
        }
        return fail("unsupported JPEG marker");
    }
    int next_marker() {   // 0xFF: no marker here
        if (pending_marker) { const int m = pending_marker; pending_marker = 0; return m; }
        int x = get8();
        if (x != 0xFF) return 0xFF;
        while (x == 0xFF) x = get8();
        return x;
    }
    bool frame_header(int sof) {
        progressive = sof == 0xC2;
        const int L = get16();
        if (L < 11) return fail("bad SOF length");
        if (get8() != 8) return fail("only 8-bit JPEG is supported");
        img_y = get16();
        img_x = get16();
        if (img_y == 0 || img_x == 0) return fail("JPEG with zero size");
        n_comp = get8();
        if (n_comp != 1 && n_comp != 3) return fail("only 1- and 3-component JPEG is supported (no CMYK)");
        if (L != 8 + 3 * n_comp) return fail("bad SOF length");
        rgb_ids = 0;
        for (int i = 0; i < n_comp; ++i) {
            static const unsigned char rgb[3] = { 'R', 'G', 'B' };
            comp[i].id = get8();
            if (n_comp == 3 && comp[i].id == rgb[i]) ++rgb_ids;
            const int q = get8();
            comp[i].h = q >> 4; comp[i].v = q & 15;
            if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4) return fail("bad JPEG sampling factors");
            comp[i].tq = get8();
            if (comp[i].tq > 3) return fail("bad JPEG quantisation table index");
        }
        h_max = v_max = 1;
        for (int i = 0; i < n_comp; ++i) { h_max = std::max(h_max, comp[i].h); v_max = std::max(v_max, comp[i].v); }
        for (int i = 0; i < n_comp; ++i)
            if (h_max % comp[i].h || v_max % comp[i].v) return fail("bad JPEG sampling factors");
        mcu_w = h_max * 8; mcu_h = v_max * 8;
        mcu_x = (img_x + mcu_w - 1) / mcu_w;
        mcu_y = (img_y + mcu_h - 1) / mcu_h;
        if ((uint64_t)mcu_x * mcu_w * (uint64_t)mcu_y * mcu_h > (1ull << 30)) return fail("JPEG too large");
        {   // a header must not be able to demand gigabytes that the file cannot fill: every 8x8 block costs at least one bit
            // of entropy-coded data (a Huffman code has no zero-length words), so fewer bits than blocks is not an image
            uint64_t blocks = 0;
            for (int i = 0; i < n_comp; ++i) blocks += (uint64_t)mcu_x * comp[i].h * (uint64_t)mcu_y * comp[i].v;
            if (blocks > (uint64_t)(end - p) * 8u) return fail("JPEG data too short for its dimensions");
        }
        for (int i = 0; i < n_comp; ++i) {
            Component& c = comp[i];
            c.x = (img_x * c.h + h_max - 1) / h_max;
            c.y = (img_y * c.v + v_max - 1) / v_max;
            c.pw = mcu_x * c.h * 8;
            c.ph = mcu_y * c.v * 8;
            c.bw = c.pw / 8; c.bh = c.ph / 8;
            c.data.assign((size_t)c.pw * c.ph, 0);
            if (progressive) c.coeff.assign((size_t)c.pw * c.ph, 0);
        }
        return true;
    }
    bool scan_header() {
        const int L = get16();
        scan.n = get8();
        if (scan.n < 1 || scan.n > 4 || scan.n > n_comp) return fail("bad SOS component count");
        if (L != 6 + 2 * scan.n) return fail("bad SOS length");
        for (int i = 0; i < scan.n; ++i) {
            const int id = get8(), q = get8();
            int which = 0;
            while (which < n_comp && comp[which].id != id) ++which;
            if (which == n_comp) return fail("SOS names an unknown component");
            comp[which].td = q >> 4; comp[which].ta = q & 15;
            if (comp[which].td > 3 || comp[which].ta > 3) return fail("bad SOS table index");
            scan.order[i] = which;
        }
        scan.Ss = get8();
        scan.Se = get8();
        const int a = get8();
        scan.Ah = a >> 4; scan.Al = a & 15;
        if (progressive) {
            if (scan.Ss > 63 || scan.Se > 63 || scan.Ss > scan.Se || scan.Ah > 13 || scan.Al > 13) return fail("bad SOS");
        } else {
            if (scan.Ss != 0 || scan.Ah != 0 || scan.Al != 0) return fail("bad SOS");
            scan.Se = 63;
        }
        return true;
    }

    // One entropy-coded segment (with its restart intervals), T.81 E.2.3-E.2.5.  The scan is a sequence of "units": MCUs when it
    // interleaves several components, single blocks of the component's own grid (no MCU padding) when it has one component.
    // `unit(i, j)` decodes unit (i, j); after every unit the restart counter runs, and at the end of an interval the segment
    // must stand at an RSTn marker — if it does not, the scan ends there (what the reference's loader does with such files).
    template <class Unit>
    bool for_each_unit(SegmentBits& in, int nx, int ny, Unit unit) {
        auto begin_interval = [&]() {
            in.restart();
            for (Component& c : comp) c.pred = 0;
            scan.eobrun = 0;
            scan.until_restart = restart_interval ? restart_interval : 0x7FFFFFFF;
        };
        begin_interval();
        for (int j = 0; j < ny; ++j)
            for (int i = 0; i < nx; ++i) {
                if (const char* e = unit(i, j)) return fail(e);
                if (--scan.until_restart <= 0) {
                    in.fill();                                   // (reaches the marker, if the interval really ends here)
                    if (!(in.marker() >= 0xD0 && in.marker() <= 0xD7)) return true;
                    begin_interval();
                }
            }
        return true;
    }
    bool entropy_scan() {
        SegmentBits in(p, end);
        bool ok;
        short block[64];
        const char* const kMissing = "JPEG table missing";
        if (scan.n == 1) {
            Component& c = comp[scan.order[0]];
            const int nx = (c.x + 7) >> 3, ny = (c.y + 7) >> 3;
            if (!progressive)
                ok = for_each_unit(in, nx, ny, [&](int i, int j) -> const char* {
                    if (!dc[c.td].present || !ac[c.ta].present || !have_q[c.tq]) return kMissing;
                    if (const char* e = decode_sequential_block(in, dc[c.td], ac[c.ta], dequant[c.tq], c.pred, block)) return e;
                    idct(c.data.data() + (size_t)c.pw * j * 8 + i * 8, c.pw, block);
                    return nullptr;
                });
            else
                ok = for_each_unit(in, nx, ny, [&](int i, int j) -> const char* {
                    short* blk = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                    if (scan.Ss == 0) {
                        if (scan.Se != 0) return "corrupt progressive JPEG";       // a DC scan codes the DC coefficient only (G.1.1.1.1)
                        if (!dc[c.td].present) return kMissing;
                        return decode_dc_progressive(in, dc[c.td], scan, c.pred, blk);
                    }
                    if (!ac[c.ta].present) return kMissing;
                    return scan.Ah == 0 ? decode_ac_first(in, ac[c.ta], scan, blk) : decode_ac_refine(in, ac[c.ta], scan, blk);
                });
        } else {
            ok = for_each_unit(in, mcu_x, mcu_y, [&](int i, int j) -> const char* {
                for (int k = 0; k < scan.n; ++k) {
                    Component& c = comp[scan.order[k]];
                    if (progressive) {                                              // interleaved progressive scans are DC scans
                        if (scan.Ss != 0 || scan.Se != 0) return "corrupt progressive JPEG";
                        if (!dc[c.td].present) return kMissing;
                    } else if (!dc[c.td].present || !ac[c.ta].present || !have_q[c.tq]) return kMissing;
                    for (int y = 0; y < c.v; ++y)
                        for (int x = 0; x < c.h; ++x) {
                            const int bx = i * c.h + x, by = j * c.v + y;
                            if (progressive) {
                                if (const char* e = decode_dc_progressive(in, dc[c.td], scan, c.pred, c.coeff.data() + 64 * ((size_t)bx + (size_t)by * c.bw))) return e;
                            } else {
                                if (const char* e = decode_sequential_block(in, dc[c.td], ac[c.ta], dequant[c.tq], c.pred, block)) return e;
                                idct(c.data.data() + (size_t)c.pw * by * 8 + bx * 8, c.pw, block);
                            }
                        }
                }
                return nullptr;
            });
        }
        pending_marker = in.marker();
        return ok;
    }
    void finish_progressive() {
        for (int n = 0; n < n_comp; ++n) {
            Component& c = comp[n];
            const int w = (c.x + 7) >> 3, h = (c.y + 7) >> 3;
            for (int j = 0; j < h; ++j)
                for (int i = 0; i < w; ++i) {
                    short* data = c.coeff.data() + 64 * ((size_t)i + (size_t)j * c.bw);
                    const uint16_t* dq = dequant[c.tq];
                    for (int k = 0; k < 64; ++k) data[k] = (short)(data[k] * dq[k]);
                    idct(c.data.data() + (size_t)c.pw * j * 8 + i * 8, c.pw, data);
                }
        }
    }

    bool decode() {
        if (get8() != 0xFF || get8() != 0xD8) return fail("not a JPEG file");
        int m = next_marker();
        while (!(m == 0xC0 || m == 0xC1 || m == 0xC2)) {
            if (m == 0xFF) { if (p >= end) return fail("no SOF marker"); m = next_marker(); continue; }
            if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) return fail("unsupported JPEG process (lossless / arithmetic)");
            if (!process_marker(m)) return false;
            m = next_marker();
        }
        if (!frame_header(m)) return false;
        m = next_marker();
        while (m != 0xD9) {
            if (m == 0xDA) {
                if (!scan_header()) return false;
                if (!entropy_scan()) return false;
                if (!pending_marker) {   // the decoder did not run into the marker that ends the segment: look for it
                    while (p < end) {
                        const int x = get8();
                        if (x == 0xFF) { const int y = get8(); if (y != 0 && y != 0xFF) { pending_marker = y; break; } if (y == 0xFF) --p; }
                    }
                    if (!pending_marker && p >= end) break;   // missing EOI: decode what we have, like the reference's loader
                }
            } else if (m == 0xDC) {
                const int L = get16(), nl = get16();
                if (L != 4 || nl != img_y) return fail("bad DNL");
            } else if (m == 0xFF) {
                if (p >= end) break;
            } else if (!process_marker(m)) {
                return false;
            }
            m = next_marker();
        }
        if (progressive) finish_progressive();
        return true;
    }
};

// ---- upsampling and colour (see the header comment) ----------------------------------------------------------------------
// One up-sampler for all ratios.  A component plane is (h_max / h) x (v_max / v) times smaller than the image.  For the
// ratios 1 and 2 the output sample is the triangle-filtered blend of the nearest and the second-nearest source sample
// in each direction, weights 3/4 and 1/4, evaluated vertically first, with ONE rounding per direction combination:
//   v only:  (3 n + f + 2) >> 2        h only:  (3 c + side + 2) >> 2        both:  (3 (3 n + f)_c + (3 n + f)_side + 8) >> 4
// (edge columns, which have no outer neighbour: the vertical blend alone, rounded as in "v only"); other ratios replicate.
struct Plane {
    const uint8_t* data;
    int w2;            // row pitch
    int w, h;          // valid samples
    int hs, vs;        // up-sampling factors
    // the row of the image-sized plane that output row j needs, written to `out` (W samples); rows are visited in order
    void row(int j, int W, uint8_t* out) const {
        if (!((hs == 1 || hs == 2) && (vs == 1 || vs == 2))) {                     // every other ratio: replicate
            const uint8_t* src = data + (size_t)std::min(j / vs, h - 1) * w2;
            for (int i = 0; i < W; ++i) out[i] = src[std::min(i / hs, w - 1)];
            return;
        }
        // vertical neighbours of output row j: the source row it falls in, and the one on the side of the nearer edge
        int near_r = 0, far_r = 0;
        if (vs == 2) {
            near_r = std::min(j >> 1, h - 1);
            far_r = (j & 1) ? std::min(near_r + 1, h - 1) : std::max(near_r - 1, 0);
        } else near_r = far_r = std::min(j, h - 1);
        const uint8_t* n = data + (size_t)near_r * w2;
        const uint8_t* f = data + (size_t)far_r * w2;
        const int wl = (W + hs - 1) / hs;                                           // source samples this row uses
        if (hs == 1) {
            if (vs == 1) { memcpy(out, n, (size_t)W); return; }
            for (int i = 0; i < W; ++i) out[i] = (uint8_t)((3 * n[i] + f[i] + 2) >> 2);
            return;
        }
        // hs == 2: vertical blend kept unrounded (x4, or x1 when there is no vertical step), then the horizontal blend
        auto vblend = [&](int i) { return vs == 2 ? 3 * n[i] + f[i] : (int)n[i]; };
        const int unit = vs == 2 ? 4 : 1;                                           // scale of vblend
        auto put = [&](int x, uint8_t v) { if (x < W + 3) out[x] = v; };           // (the scratch row has slack)
        for (int i = 0; i < wl; ++i) {
            const int c = vblend(i);
            const int left = i > 0 ? vblend(i - 1) : -1, right = i + 1 < wl ? vblend(i + 1) : -1;
            // (the horizontal-only filter of the decoder this reproduces weights its LAST interior sample the other way
            // round: 3 * in[w-2] + in[w-1]; kept, or the decoded bytes would differ in one column)
            if (vs == 1 && i == wl - 1 && i > 0) put(2 * i, (uint8_t)((3 * left + c + 2) >> 2));
            else put(2 * i, left < 0 ? (uint8_t)((c + unit / 2) / unit) : (uint8_t)((3 * c + left + 2 * unit) / (4 * unit)));
            put(2 * i + 1, right < 0 ? (uint8_t)((c + unit / 2) / unit) : (uint8_t)((3 * c + right + 2 * unit) / (4 * unit)));
        }
    }
};

// JFIF YCbCr -> RGB in 20-bit fixed point; the constants are rounded at 12 bits first, and the Cb term of green keeps only
// its upper 16 bits (all three as in the decoder whose bytes this must reproduce)
struct YccToRgb {
    int cr_r, cr_g, cb_g, cb_b;
    YccToRgb() {
        auto fx = [](double x) { return ((int)(x * 4096.0 + 0.5)) << 8; };
        cr_r = fx(1.40200); cr_g = -fx(0.71414); cb_g = -fx(0.34414); cb_b = fx(1.77200);
    }
    inline void operator()(int y, int cb, int cr, uint8_t* rgba) const {
        const int base = (y << 20) + (1 << 19), u = cb - 128, v = cr - 128;
        rgba[0] = Decoder::clamp8((base + v * cr_r) >> 20);
        rgba[1] = Decoder::clamp8((base + v * cr_g + (int)(((unsigned)(u * cb_g)) & 0xffff0000u)) >> 20);
        rgba[2] = Decoder::clamp8((base + u * cb_b) >> 20);
        rgba[3] = 255;
    }
};

}  // namespace

bool decode_jpeg(const uint8_t* data, size_t len, Image& img, std::string& err) {
    Decoder d;
    d.p = data; d.end = data + len; d.err = &err;
    if (!d.decode()) return false;
    const int W = d.img_x, H = d.img_y;
    img.width = (uint32_t)W; img.height = (uint32_t)H;
    img.rgba.assign((size_t)W * H * 4, 255);

    Plane plane[3];
    std::vector<uint8_t> scratch[3];
    for (int k = 0; k < d.n_comp; ++k) {
        const Component& c = d.comp[k];
        plane[k] = Plane{ c.data.data(), c.pw, c.x, c.y, d.h_max / c.h, d.v_max / c.v };
        scratch[k].assign((size_t)W + 3 + 16, 0);
    }
    const bool is_rgb = d.n_comp == 3 && (d.rgb_ids == 3 || (d.adobe_transform == 0 && !d.has_jfif));
    static const YccToRgb to_rgb;
    for (int j = 0; j < H; ++j) {
        const uint8_t* row[3] = { nullptr, nullptr, nullptr };
        for (int k = 0; k < d.n_comp; ++k) { plane[k].row(j, W, scratch[k].data()); row[k] = scratch[k].data(); }
        uint8_t* out = img.rgba.data() + (size_t)j * W * 4;
        if (d.n_comp == 1) {
            for (int i = 0; i < W; ++i) { out[4 * i] = out[4 * i + 1] = out[4 * i + 2] = row[0][i]; out[4 * i + 3] = 255; }
        } else if (is_rgb) {
            for (int i = 0; i < W; ++i) { out[4 * i] = row[0][i]; out[4 * i + 1] = row[1][i]; out[4 * i + 2] = row[2][i]; out[4 * i + 3] = 255; }
        } else {
            for (int i = 0; i < W; ++i) to_rgb(row[0][i], row[1][i], row[2][i], out + 4 * i);
        }
    }
    return true;
}

}  // namespace m2s_host
