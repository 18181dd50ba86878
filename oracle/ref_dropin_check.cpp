// ref_dropin_check.cpp — TEST INFRASTRUCTURE.  The drop-in seen from the REFERENCE's side, compiled and run:
//
//     SceneManager::loadModel      the reference's own (SceneManager.cpp, tiny_gltf, stb_image), from where it lies
//  -> ConversionPass::execute      NOT the reference's ConversionPass.cpp but oracle/ref_dropin/ConversionPassHip.cpp — the
//                                  replacement body INTEGRATION.md shows — linked against libm2s_hip.so: the conversion runs
//                                  on the GPU through the C ABI (include/m2s.h)
//  -> SceneManager::exportPly      the reference's own again: it reads renderContext.numberOfGaussians records back from
//                                  renderContext.gaussianBuffer, which the replacement body filled
//
// on the same minimal software GL as ref_pipeline_check (ref_swgl.h; no draw call is ever issued here).  The output file has
// ref_pipeline_check's layout, so the all-reference run and this run of the same .glb compare directly
// (tests/test_gpu_dropin.py).  Needs a GPU at run time (libm2s_hip.so has no CPU fallback).
//
//   ref_dropin_check in.glb R out_records.bin [out.ply format gaussianStd]
//   M2S_DROPIN_LOAD_FIRST=other.glb : SceneManager::loadModel + execute on other.glb first, then everything above on in.glb with
//   the SAME SceneManager / RenderContext / pass — a second model with as many meshes re-uses dataMeshAndGlMesh's allocation
//   (clear + reserve + push_back), which an address-keyed "is the scene resident" test would mistake for the first model
#include "utils/SceneManager.hpp"
#include "renderer/renderPasses/ConversionPass.hpp"

#include <chrono>
#include <fstream>
#include <thread>

#include "ref_swgl.h"

extern "C" void GLAPIENTRY glDrawArrays(GLenum, GLint, GLsizei) {
    fprintf(stderr, "ref_dropin_check: glDrawArrays called — the conversion is supposed to run through libm2s_hip.so\n");
    exit(73);
}

int main(int argc, char** argv) {
    if (argc != 4 && argc != 7) {
        fprintf(stderr, "usage: ref_dropin_check in.glb R out_records.bin [out.ply format gaussianStd]\n");
        return 64;
    }
    RenderContext rc;
    // the two buffers Renderer::initialize creates for this pass (renderer.cpp:48-50,75-77)
    glGenBuffers(1, &rc.gaussianBuffer);
    glBindBuffer(GL_SHADER_STORAGE_BUFFER, rc.gaussianBuffer);
    glBufferData(GL_SHADER_STORAGE_BUFFER, 0, nullptr, GL_DYNAMIC_DRAW);
    glGenBuffers(1, &rc.atomicCounterBufferConversionPass);
    glBindBuffer(GL_ATOMIC_COUNTER_BUFFER, rc.atomicCounterBufferConversionPass);
    glBufferData(GL_ATOMIC_COUNTER_BUFFER, sizeof(uint32_t), nullptr, GL_DYNAMIC_DRAW);
    rc.resolutionTarget = (unsigned)atoi(argv[2]);
    int rcode = 0;
    try {
        SceneManager sm(rc);
        ConversionPass pass;
        if (const char* first = getenv("M2S_DROPIN_LOAD_FIRST")) {
            if (!sm.loadModel(first, "")) return 2;
            pass.execute(rc);
        }
        if (!sm.loadModel(argv[1], "")) return 2;
        pass.execute(rc);                                         // first call: uploads the scene, converts
        const auto t0 = std::chrono::steady_clock::now();
        pass.execute(rc);                                         // the scene is resident: what a slider move costs
        const double exec_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();

        const uint32_t counter = (uint32_t)rc.numberOfGaussians;
        const std::vector<uint8_t>& ssbo = swgl::buffers[rc.gaussianBuffer];
        const uint64_t ssbo_bytes = ssbo.size();
        const uint32_t cap = (uint32_t)(ssbo_bytes / 96);         // the body sizes the buffer as the reference does: cap * 96
        const uint64_t stored = counter < cap ? counter : cap;
        std::ofstream f(std::string(argv[3]) == "-" ? "/dev/null" : argv[3], std::ios::binary);
        f.write(reinterpret_cast<const char*>(&counter), 4);
        f.write(reinterpret_cast<const char*>(&cap), 4);
        f.write(reinterpret_cast<const char*>(&ssbo_bytes), 8);
        f.write(reinterpret_cast<const char*>(ssbo.data()), (std::streamsize)(stored * 96));
        fprintf(stdout, "{\"counter\": %u, \"max_gaussians\": %u, \"ssbo_bytes\": %llu, \"meshes\": %zu, \"execute_ms\": %.3f}\n",
                counter, cap, (unsigned long long)ssbo_bytes, rc.dataMeshAndGlMesh.size(), exec_ms);

        if (argc == 7) {
            const unsigned fmt = (unsigned)atoi(argv[5]);
            rc.gaussianStd = (float)atof(argv[6]);
            std::remove(argv[4]);
            sm.exportPly(argv[4], fmt);   // writes from a detached thread (SceneManager.cpp:671-676): wait for the file
            const uint64_t row = fmt == 1 ? 76 : fmt == 2 ? 48 : 248;
            rcode = 5;
            for (int i = 0; i < 6000 && rcode; ++i) {
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
                std::ifstream p(argv[4], std::ios::binary | std::ios::ate);
                if (!p) continue;
                const uint64_t sz = (uint64_t)p.tellg();
                if (sz > (uint64_t)counter * row) {   // header + all rows; let the writer close the stream
                    std::this_thread::sleep_for(std::chrono::milliseconds(200));
                    rcode = 0;
                }
            }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "ref_dropin_check: %s\n", e.what());
        return 6;
    }
    return rcode;
}
