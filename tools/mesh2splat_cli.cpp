// mesh2splat — headless command-line converter.  The reference has no CLI (src/utils/argparser.hpp is
// dead code and main() ignores argv); its only workflow is the GUI's LoadModel -> RunConversion ->
// SavePLY event sequence (src/renderer/guiRendererConcreteMediator.cpp:11-29,51-57,111-115 and the
// batch state machine :134-251).  This tool runs exactly that sequence through the C ABI (include/m2s.h).
//
//   mesh2splat in.glb out.ply [--density R | --quality q [--max-res 1024|2048|4096]]
//              [--std s] [--format 0|1|2] [--device d] [--cap n] [--pipeline auto|multipass] [--timing]
//
// Defaults are the GUI's: quality 0.5 with max-res 1024 -> R = int(16 + q*(maxRes-16)) = 520
// (ImGuiUI.cpp:512, main.cpp:26), gaussian std 0.65 (main.cpp:26), format 0 (standard 3DGS .ply).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../include/m2s.h"

static void usage() {
    std::fprintf(stderr,
                 "usage: mesh2splat in.glb out.ply [--density R | --quality q [--max-res M]] [--std s] [--format 0|1|2]\n"
                 "                  [--device d] [--cap n (0 = unlimited, default: reference formula)] [--pipeline auto|multipass] [--timing]\n");
}

int main(int argc, char** argv) {
    if (argc < 3) { usage(); return 2; }
    const std::string in = argv[1], out = argv[2];
    double quality = 0.5, std_dev = 0.65;
    long max_res = 1024, density = -1, device = 0, cap = -1, format = 0;
    int pipeline = M2S_PIPELINE_AUTO;
    bool timing = false;
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(2); } return argv[++i]; };
        if (a == "--density") density = std::atol(next());
        else if (a == "--quality") quality = std::atof(next());
        else if (a == "--max-res") max_res = std::atol(next());
        else if (a == "--std") std_dev = std::atof(next());
        else if (a == "--format") format = std::atol(next());
        else if (a == "--device") device = std::atol(next());
        else if (a == "--cap") cap = std::atol(next());
        else if (a == "--pipeline") pipeline = std::string(next()) == "multipass" ? M2S_PIPELINE_MULTIPASS : M2S_PIPELINE_AUTO;
        else if (a == "--timing") timing = true;
        else { usage(); return 2; }
    }
    const uint32_t R = density > 0 ? (uint32_t)density : (uint32_t)(int)(16 + quality * (double)(max_res - 16));  // ImGuiUI.cpp:512
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };

    const auto t0 = now();
    m2s_host_scene* scene = nullptr;
    if (m2s_load_glb(in.c_str(), &scene) != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_io_last_error()); return 1; }
    if (*m2s_host_scene_warnings(scene)) std::fprintf(stderr, "%s", m2s_host_scene_warnings(scene));
    const auto t1 = now();

    m2s_ctx* ctx = nullptr;
    if (m2s_create((int)device, &ctx) != M2S_OK) { std::fprintf(stderr, "%s\n", m2s_last_error(nullptr)); return 1; }
    auto die = [&](const char* what) { std::fprintf(stderr, "%s: %s\n", what, m2s_last_error(ctx)); m2s_destroy(ctx); m2s_free_host_scene(scene); return 1; };
    if (m2s_set_pipeline(ctx, pipeline) != M2S_OK) return die("set_pipeline");
    if (m2s_set_max_gaussians(ctx, cap) != M2S_OK) return die("set_max_gaussians");
    if (m2s_upload_scene(ctx, m2s_host_scene_meshes(scene), m2s_host_scene_num_meshes(scene)) != M2S_OK) return die("upload");
    const auto t2 = now();
    uint64_t total = 0;
    if (m2s_convert(ctx, R, &total) != M2S_OK) return die("convert");
    const auto t3 = now();
    if (m2s_export_ply(ctx, out.c_str(), (uint32_t)format, (float)std_dev) != M2S_OK) return die("export");
    const auto t4 = now();

    std::printf("%s: %u mesh(es), %llu triangles, density %u -> %llu Gaussians (%llu stored) -> %s (format %ld)\n", in.c_str(),
                m2s_host_scene_num_meshes(scene), (unsigned long long)m2s_num_triangles(ctx), R, (unsigned long long)total,
                (unsigned long long)m2s_num_stored(ctx), out.c_str(), format);
    if (timing)
        std::printf("load %.2f ms | upload %.2f ms | convert %.3f ms (first call, incl. buffer allocation) | export %.2f ms\n", ms(t0, t1), ms(t1, t2),
                    ms(t2, t3), ms(t3, t4));
    m2s_destroy(ctx);
    m2s_free_host_scene(scene);
    return 0;
}
