// fuzz_host.cpp — mutation fuzzer for the host-side readers of untrusted files (.glb container + JSON, PNG, JPEG, .ply).
// Built by tools/fuzz_host.sh with -fsanitize=address,undefined from the product sources; no GPU, no HIP.
//
//   fuzz_host <iterations> <seed> <file>...
//
// Every iteration takes one seed file, applies a few mutations (bit flips, byte splats, truncation, 16/32-bit length
// fields set to extreme values, chunk duplication, a block copied from another seed) and hands the result to the reader for
// its kind: *.png / *.jpg straight to decode_png / decode_jpeg, *.glb to m2s_load_glb, *.ply to m2s_read_ply.  The readers
// may accept or reject the file; the run fails only when a sanitizer reports, a reader crashes, or an accepted image has
// an inconsistent size.  A line of statistics is printed at the end.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../mesh2splat_amd/csrc/m2s_host.h"

namespace {

uint64_t g_state;
uint64_t rnd() {   // splitmix64
    uint64_t z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
size_t below(size_t n) { return n ? (size_t)(rnd() % n) : 0; }

struct Seed {
    std::string path;
    int kind;   // 0 png, 1 jpeg, 2 glb, 3 ply
    std::vector<uint8_t> bytes;
};

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

void mutate(std::vector<uint8_t>& b, const std::vector<Seed>& seeds) {
    static const uint32_t extremes[] = { 0u, 1u, 0x7Fu, 0x80u, 0xFFu, 0x100u, 0x7FFFu, 0x8000u, 0xFFFFu, 0x10000u, 0x7FFFFFFFu, 0x80000000u, 0xFFFFFFFFu };
    const int n_mut = 1 + (int)below(6);
    for (int m = 0; m < n_mut && !b.empty(); ++m) {
        switch (below(9)) {
        case 0: b[below(b.size())] ^= (uint8_t)(1u << below(8)); break;
        case 1: b[below(b.size())] = (uint8_t)rnd(); break;
        case 2: {   // splat a run
            const size_t at = below(b.size()), len = 1 + below(16);
            const uint8_t v = (uint8_t)rnd();
            for (size_t i = at; i < b.size() && i < at + len; ++i) b[i] = v;
            break;
        }
        case 3: b.resize(below(b.size()) + 1); break;   // truncate
        case 4: {   // 16-bit field (either endianness)
            if (b.size() < 2) break;
            const size_t at = below(b.size() - 1);
            const uint32_t v = extremes[below(sizeof extremes / sizeof *extremes)];
            if (rnd() & 1) { b[at] = (uint8_t)(v >> 8); b[at + 1] = (uint8_t)v; } else { b[at] = (uint8_t)v; b[at + 1] = (uint8_t)(v >> 8); }
            break;
        }
        case 5: {   // 32-bit field
            if (b.size() < 4) break;
            const size_t at = below(b.size() - 3);
            const uint32_t v = extremes[below(sizeof extremes / sizeof *extremes)];
            const bool be = rnd() & 1;
            for (int k = 0; k < 4; ++k) b[at + k] = (uint8_t)(v >> (be ? 24 - 8 * k : 8 * k));
            break;
        }
        case 6: {   // duplicate a block in place (chunks / segments twice)
            const size_t at = below(b.size()), len = 1 + below(std::min<size_t>(512, b.size() - at));
            std::vector<uint8_t> blk(b.begin() + at, b.begin() + at + len);
            b.insert(b.begin() + at, blk.begin(), blk.end());
            break;
        }
        case 7: {   // a block from another seed over this one
            const Seed& o = seeds[below(seeds.size())];
            if (o.bytes.empty()) break;
            const size_t from = below(o.bytes.size()), len = 1 + below(std::min<size_t>(256, o.bytes.size() - from)), at = below(b.size());
            for (size_t i = 0; i < len && at + i < b.size(); ++i) b[at + i] = o.bytes[from + i];
            break;
        }
        default: {  // delete a block
            const size_t at = below(b.size()), len = 1 + below(std::min<size_t>(64, b.size() - at));
            b.erase(b.begin() + at, b.begin() + at + len);
            break;
        }
        }
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: fuzz_host <iterations> <seed> <file>...\n"); return 2; }
    const long iters = std::atol(argv[1]);
    g_state = std::strtoull(argv[2], nullptr, 0);
    std::vector<Seed> seeds;
    for (int i = 3; i < argc; ++i) {
        Seed s;
        s.path = argv[i];
        s.kind = ends_with(s.path, ".png") ? 0 : (ends_with(s.path, ".jpg") || ends_with(s.path, ".jpeg")) ? 1 : ends_with(s.path, ".glb") ? 2 : ends_with(s.path, ".ply") ? 3 : -1;
        if (s.kind < 0) continue;
        FILE* f = std::fopen(argv[i], "rb");
        if (!f) { std::fprintf(stderr, "cannot read %s\n", argv[i]); return 2; }
        std::fseek(f, 0, SEEK_END);
        const long n = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        s.bytes.resize((size_t)n);
        if (n && std::fread(s.bytes.data(), 1, (size_t)n, f) != (size_t)n) return 2;
        std::fclose(f);
        seeds.push_back(std::move(s));
    }
    if (seeds.empty()) return 2;
    const std::string tmp = std::string("/tmp/fuzz_host_") + std::to_string((unsigned long long)g_state) + ".bin";
    long accepted[4] = { 0, 0, 0, 0 }, tried[4] = { 0, 0, 0, 0 };
    for (long it = 0; it < iters; ++it) {
        const Seed& s = seeds[below(seeds.size())];
        std::vector<uint8_t> b = s.bytes;
        if (it >= (long)seeds.size()) mutate(b, seeds);   // the first pass runs every seed unmodified
        else b = seeds[(size_t)it].bytes;
        const int kind = it >= (long)seeds.size() ? s.kind : seeds[(size_t)it].kind;
        ++tried[kind];
        if (kind <= 1) {
            m2s_host::Image img;
            std::string err;
            bool ok = false;
            try {   // (the loader calls the decoders inside the same try / catch: a failed allocation is a rejected file)
                ok = kind == 0 ? m2s_host::decode_png(b.data(), b.size(), img, err) : m2s_host::decode_jpeg(b.data(), b.size(), img, err);
            } catch (const std::exception&) { ok = false; }
            if (ok) {
                ++accepted[kind];
                if (img.rgba.size() != (size_t)img.width * img.height * 4 || img.width == 0 || img.height == 0) {
                    std::fprintf(stderr, "iteration %ld: accepted image with inconsistent size %ux%u / %zu bytes\n", it, img.width, img.height, img.rgba.size());
                    return 1;
                }
            }
        } else {
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f) return 2;
            if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f);
            std::fclose(f);
            if (kind == 2) {
                m2s_host_scene* sc = nullptr;
                if (m2s_load_glb(tmp.c_str(), &sc) == M2S_OK) {
                    ++accepted[kind];
                    // touch what a caller would read
                    const uint32_t nm = m2s_host_scene_num_meshes(sc);
                    const m2s_mesh* ms = m2s_host_scene_meshes(sc);
                    volatile float sink = 0;
                    for (uint32_t i = 0; i < nm; ++i) {
                        if (ms[i].n_vertices) sink = sink + ms[i].vertices[(size_t)ms[i].n_vertices * ms[i].stride_floats - 1];
                        for (int k = 0; k < 3; ++k)
                            if (ms[i].tex[k].rgba8) sink = sink + ms[i].tex[k].rgba8[(size_t)ms[i].tex[k].width * ms[i].tex[k].height * 4 - 1];
                        (void)m2s_host_scene_mesh_name(sc, i);
                    }
                    (void)m2s_host_scene_warnings(sc);
                    m2s_free_host_scene(sc);
                }
            } else {
                m2s_gaussian* rec = nullptr;
                uint64_t n = 0;
                int pbr = 0;
                if (m2s_read_ply(tmp.c_str(), &rec, &n, &pbr) == M2S_OK) {
                    ++accepted[kind];
                    volatile float sink = 0;
                    if (n) sink = sink + rec[n - 1].pbr[3];
                    m2s_free_records(rec);
                }
            }
        }
    }
    std::remove(tmp.c_str());
    std::printf("fuzz_host: %ld iterations, no sanitizer report; accepted/tried png %ld/%ld jpeg %ld/%ld glb %ld/%ld ply %ld/%ld\n", iters, accepted[0], tried[0],
                accepted[1], tried[1], accepted[2], tried[2], accepted[3], tried[3]);
    return 0;
}
