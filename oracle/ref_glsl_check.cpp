// ref_glsl_check.cpp — TEST INFRASTRUCTURE.  Executes the REFERENCE'S OWN conversion shaders on the CPU and
// compares them, stage by stage, with the oracle's restatement (m2s_oracle.c).
//
// The three shader sources (src/shaders/conversion/converter{VS,GS,FS}.glsl) are translated at build time by
// oracle/glsl2cpp.py — a purely syntactic rewrite into oracle/_ref/gen/*.inc — and compiled here against glm,
// which the reference vendors and which mirrors GLSL's vector/matrix types and built-in functions.  What GLSL
// leaves to fixed-function hardware is supplied by this harness and is therefore NOT checked here:
//   * primitive assembly / rasterisation / varying interpolation — the harness feeds both sides the same
//     barycentric sample points and interpolates with the oracle's pinned formula;
//   * texture(): bound to the oracle's sampler (orc_debug_sample) with the same explicit level of detail;
//   * the atomic counter and the SSBO: a plain counter and a one-element array.
// Everything the shaders themselves compute — longest-edge swap, triplanar axis choice, bbox-normalised
// orthogonal UVs (gl_Position), the UV->3D Jacobian scale, quat_cast, the TBN normal, colour x factor,
// metallic/roughness selection, the record layout — is the reference's code, run as written.
//
//   ref_glsl_check <scene.bin> <R> <samples_per_triangle> [dump.bin] -> one JSON line with per-field statistics
//
// dump.bin (optional; becomes a committed golden fixture, tests/golden/make_ref_golden.py): what the REFERENCE
// shaders produced — per triangle 13 floats (gl_Position.xy x3, Scale, Quaternion), then per sample 39 floats
// (the 12 varyings and 3 levels of detail fed in, the 24-float record the FS wrote).
//
// scene.bin (written by tests/test_ref_glsl.py): u32 n_meshes; per mesh: u32 n_vertices, float bmin[3], bmax[3],
// color[4], n_vertices x 12 floats (pos, normal, tangent, uv), 3 x {u32 w, u32 h, w*h*4 bytes}.
#include "ref_glsl_env.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {
#include "m2s_oracle.h"
}

// texture(): bound to the oracle's sampler with an explicit level of detail per texture slot
static const orc_scene* g_scene = nullptr;
static uint32_t g_mesh = 0;
static float g_lambda[3] = { 0, 0, 0 };
static glm::vec4 fetch_texel(int slot, glm::vec2 uv) {
    float o[4];
    orc_debug_sample(g_scene, g_mesh, slot, uv.x, uv.y, g_lambda[slot], o);
    return glm::vec4(o[0], o[1], o[2], o[3]);
}

// ---- statistics -------------------------------------------------------------------------------------------------
struct Stat {
    double max_abs = 0, max_rel = 0;
    uint64_t n = 0, exact = 0;
    void add(float ref, float got, float scale_ref) {   // scale_ref: magnitude the error is judged against
        ++n;
        if (std::memcmp(&ref, &got, 4) == 0 || ref == got) { ++exact; return; }
        if (std::isnan(ref) && std::isnan(got)) { ++exact; return; }
        const double d = std::fabs((double)ref - (double)got);
        if (!(d <= max_abs)) max_abs = d;
        const double r = d / (std::fabs((double)scale_ref) + 1e-30);
        if (!(r <= max_rel)) max_rel = r;
    }
};
static void print_stat(const char* name, const Stat& s, bool last = false) {
    printf("\"%s\": {\"n\": %llu, \"exact\": %llu, \"max_abs\": %.3e, \"max_rel\": %.3e}%s", name, (unsigned long long)s.n,
           (unsigned long long)s.exact, s.max_abs, s.max_rel, last ? "" : ", ");
}
static float vmax3(const float* v) { return std::fmax(std::fabs(v[0]), std::fmax(std::fabs(v[1]), std::fabs(v[2]))); }

struct MeshIn {
    std::vector<float> verts;   // 12 floats per vertex
    float bmin[3], bmax[3], color[4];
    std::vector<uint8_t> tex[3];
    uint32_t tw[3], th[3];
};

int main(int argc, char** argv) {
    if (argc != 4 && argc != 5) { fprintf(stderr, "usage: ref_glsl_check scene.bin R samples_per_triangle [dump.bin]\n"); return 64; }
    FILE* dump = argc == 5 ? fopen(argv[4], "wb") : nullptr;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const uint32_t R = (uint32_t)atoi(argv[2]);
    const int n_samples = atoi(argv[3]);
    uint32_t n_meshes = 0;
    if (fread(&n_meshes, 4, 1, f) != 1) return 2;
    std::vector<MeshIn> meshes(n_meshes);
    std::vector<orc_mesh> om(n_meshes);
    for (uint32_t m = 0; m < n_meshes; ++m) {
        MeshIn& mi = meshes[m];
        uint32_t nv = 0;
        if (fread(&nv, 4, 1, f) != 1) return 2;
        if (fread(mi.bmin, 4, 3, f) != 3 || fread(mi.bmax, 4, 3, f) != 3 || fread(mi.color, 4, 4, f) != 4) return 2;
        mi.verts.resize((size_t)nv * 12);
        if (nv && fread(mi.verts.data(), 4, mi.verts.size(), f) != mi.verts.size()) return 2;
        std::memset(&om[m], 0, sizeof(orc_mesh));
        for (int k = 0; k < 3; ++k) {
            if (fread(&mi.tw[k], 4, 1, f) != 1 || fread(&mi.th[k], 4, 1, f) != 1) return 2;
            mi.tex[k].resize((size_t)mi.tw[k] * mi.th[k] * 4);
            if (!mi.tex[k].empty() && fread(mi.tex[k].data(), 1, mi.tex[k].size(), f) != mi.tex[k].size()) return 2;
            om[m].tex[k].rgba8 = mi.tex[k].empty() ? nullptr : mi.tex[k].data();
            om[m].tex[k].width = mi.tw[k];
            om[m].tex[k].height = mi.th[k];
        }
        om[m].vertices = mi.verts.data();
        om[m].n_vertices = nv;
        om[m].stride_floats = 12;
        std::memcpy(om[m].bbox_min, mi.bmin, 12);
        std::memcpy(om[m].bbox_max, mi.bmax, 12);
        std::memcpy(om[m].base_color, mi.color, 16);
    }
    fclose(f);
    orc_scene* sc = orc_scene_create(om.data(), n_meshes);
    g_scene = sc;
    glsl_env::g_texture = fetch_texel;

    Stat st_ndc, st_scale, st_quat, st_pos, st_col, st_nrm, st_pbr, st_const;
    uint64_t n_tri = 0, n_frag = 0, n_quat_sign = 0, n_degenerate = 0;
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (float)((rng >> 40) & 0xFFFFFF) / 16777216.0f; };

    for (uint32_t m = 0; m < n_meshes; ++m) {
        const MeshIn& mi = meshes[m];
        g_mesh = m;
        // uniforms (ConversionPass.cpp:100-112)
        ref_gs::u_bboxMin = glm::vec3(mi.bmin[0], mi.bmin[1], mi.bmin[2]);
        ref_gs::u_bboxMax = glm::vec3(mi.bmax[0], mi.bmax[1], mi.bmax[2]);
        ref_fs::albedoTexture.unit = 0; ref_fs::normalTexture.unit = 1; ref_fs::metallicRoughnessTexture.unit = 2;
        ref_fs::hasAlbedoMap = mi.tex[0].empty() ? 0 : 1;
        ref_fs::hasNormalMap = mi.tex[1].empty() ? 0 : 1;
        ref_fs::hasMetallicRoughnessMap = mi.tex[2].empty() ? 0 : 1;
        ref_fs::u_materialFactor = glm::vec4(mi.color[0], mi.color[1], mi.color[2], mi.color[3]);
        ref_fs::u_maxGaussians = 1 << 30;
        const size_t nt = mi.verts.size() / 36;
        for (size_t t = 0; t < nt; ++t) {
            const float* v[3] = { &mi.verts[(t * 3 + 0) * 12], &mi.verts[(t * 3 + 1) * 12], &mi.verts[(t * 3 + 2) * 12] };
            // ---- VS (reference) x3 -> GS inputs
            for (int i = 0; i < 3; ++i) {
                ref_vs::position = glm::vec3(v[i][0], v[i][1], v[i][2]);
                ref_vs::normal = glm::vec3(v[i][3], v[i][4], v[i][5]);
                ref_vs::tangent = glm::vec4(v[i][6], v[i][7], v[i][8], v[i][9]);
                ref_vs::uv = glm::vec2(v[i][10], v[i][11]);
                ref_vs::normalizedUv = glm::vec2(0);
                ref_vs::scale = glm::vec3(0);
                ref_vs::main_();
                ref_gs::gs_in[i].position = ref_vs::vs_out.position;
                ref_gs::gs_in[i].normal = ref_vs::vs_out.normal;
                ref_gs::gs_in[i].tangent = ref_vs::vs_out.tangent;
                ref_gs::gs_in[i].uv = ref_vs::vs_out.uv;
                ref_gs::gs_in[i].normalizedUv = ref_vs::vs_out.normalizedUv;
                ref_gs::gs_in[i].scale = ref_vs::vs_out.scale;
            }
            // ---- GS (reference)
            glsl_env::g_emitted.clear();
            ref_gs::main_();
            if (glsl_env::g_emitted.size() != 3) { fprintf(stderr, "GS emitted %zu vertices\n", glsl_env::g_emitted.size()); return 3; }
            const auto& E = glsl_env::g_emitted;
            // ---- oracle GS
            float ndc[6], scl[3], rot[4];
            orc_debug_gs(v[0], v[1], v[2], mi.bmin, mi.bmax, R, ndc, scl, rot);
            ++n_tri;
            bool finite = true;
            for (int i = 0; i < 3; ++i) finite = finite && std::isfinite(E[i].gl_Position.x) && std::isfinite(E[i].gl_Position.y);
            for (int k = 0; k < 2; ++k) finite = finite && std::isfinite(E[0].Scale[k]);
            for (int k = 0; k < 4; ++k) finite = finite && std::isfinite(E[0].Quaternion[k]);
            if (dump) {
                for (int i = 0; i < 3; ++i) { fwrite(&E[i].gl_Position.x, 4, 1, dump); fwrite(&E[i].gl_Position.y, 4, 1, dump); }
                fwrite(&E[2].Scale, 4, 3, dump);
                fwrite(&E[2].Quaternion, 4, 4, dump);
            }
            if (!finite && !dump) { ++n_degenerate; continue; }   // zero-area / zero-range input: both sides produce NaN/inf
            if (!finite) ++n_degenerate;
            for (int i = 0; i < 3; ++i) {
                st_ndc.add(E[i].gl_Position.x, ndc[2 * i], 1.0f);
                st_ndc.add(E[i].gl_Position.y, ndc[2 * i + 1], 1.0f);
                // the GS must forward the vertex attributes unchanged, in input order
                for (int k = 0; k < 3; ++k) st_const.add(E[i].Position[k], v[i][k], 1.0f);
                for (int k = 0; k < 3; ++k) st_const.add(E[i].Normal[k], v[i][3 + k], 1.0f);
                for (int k = 0; k < 4; ++k) st_const.add(E[i].Tangent[k], v[i][6 + k], 1.0f);
                for (int k = 0; k < 2; ++k) st_const.add(E[i].UV[k], v[i][10 + k], 1.0f);
            }
            for (int k = 0; k < 3; ++k) st_scale.add(E[0].Scale[k], scl[k], E[0].Scale[k]);
            // q and -q are the same rotation, but the reference stores one of them: demand the same one
            float qd = 0;
            for (int k = 0; k < 4; ++k) qd += E[0].Quaternion[k] * rot[k];
            if (qd < 0) ++n_quat_sign;
            for (int k = 0; k < 4; ++k) st_quat.add(E[0].Quaternion[k], rot[k], 1.0f);

            // ---- FS at sample points: centroid, then pseudo-random interior points
            for (int s = 0; s < n_samples; ++s) {
                float l1 = 1.0f / 3.0f, l2 = 1.0f / 3.0f;
                if (s > 0) { l1 = rnd(); l2 = rnd(); if (l1 + l2 > 1.0f) { l1 = 1.0f - l1; l2 = 1.0f - l2; } }
                float vary[12];
                for (int k = 0; k < 12; ++k) vary[k] = (v[0][k] + l1 * (v[1][k] - v[0][k])) + l2 * (v[2][k] - v[0][k]);
                // level of detail: an input to both sides (fixed-function in GL); exercise magnification,
                // fractional levels and the clamp at the last level
                const float lam_choices[5] = { -1.5f, 0.0f, 0.37f, 1.62f, 9.0f };
                for (int k = 0; k < 3; ++k) g_lambda[k] = lam_choices[(s + k + (int)t) % 5];
                // reference FS, flat inputs from the reference GS (provoking vertex = last, all three equal here)
                ref_fs::Position = glm::vec3(vary[0], vary[1], vary[2]);
                ref_fs::Normal = glm::vec3(vary[3], vary[4], vary[5]);
                ref_fs::Tangent = glm::vec4(vary[6], vary[7], vary[8], vary[9]);
                ref_fs::UV = glm::vec2(vary[10], vary[11]);
                ref_fs::Scale = E[2].Scale;
                ref_fs::Quaternion = E[2].Quaternion;
                ref_fs::GaussianVertex slot;
                std::memset(&slot, 0, sizeof slot);
                ref_fs::gaussianBuffer.vertices = &slot;
                ref_fs::g_validCounter.v = 0;
                ref_fs::main_();
                const float* ref = reinterpret_cast<const float*>(&slot);
                if (dump) { fwrite(vary, 4, 12, dump); fwrite(g_lambda, 4, 3, dump); fwrite(ref, 4, 24, dump); }
                // oracle FS, flat inputs from the oracle GS
                float rec[24];
                orc_debug_fs(sc, m, vary, g_lambda, scl, rot, rec);
                ++n_frag;
                const float pm = vmax3(ref);
                for (int k = 0; k < 3; ++k) st_pos.add(ref[k], rec[k], pm);
                st_const.add(ref[3], rec[3], 1.0f);
                for (int k = 4; k < 8; ++k) st_col.add(ref[k], rec[k], ref[k]);
                for (int k = 8; k < 11; ++k) st_scale.add(ref[k], rec[k], ref[k]);
                st_const.add(ref[11], rec[11], 1.0f);
                const float nm = vmax3(ref + 12);
                for (int k = 12; k < 15; ++k) st_nrm.add(ref[k], rec[k], nm);
                st_const.add(ref[15], rec[15], 1.0f);
                for (int k = 16; k < 20; ++k) st_quat.add(ref[k], rec[k], 1.0f);
                for (int k = 20; k < 22; ++k) st_pbr.add(ref[k], rec[k], ref[k]);
                st_const.add(ref[22], rec[22], 1.0f);
                st_const.add(ref[23], rec[23], 1.0f);
            }
        }
    }
    printf("{\"triangles\": %llu, \"fragments\": %llu, \"degenerate\": %llu, \"quat_sign_flips\": %llu, ", (unsigned long long)n_tri,
           (unsigned long long)n_frag, (unsigned long long)n_degenerate, (unsigned long long)n_quat_sign);
    print_stat("ndc", st_ndc); print_stat("scale", st_scale); print_stat("quaternion", st_quat); print_stat("position", st_pos);
    print_stat("color", st_col); print_stat("normal", st_nrm); print_stat("pbr", st_pbr); print_stat("passthrough", st_const, true);
    printf("}\n");
    if (dump) fclose(dump);
    orc_scene_destroy(sc);
    return 0;
}
