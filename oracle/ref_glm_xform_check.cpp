// oracle/_ref cross-check #2: the .glb loader's glm-free transform code (mesh2splat_amd/csrc/m2s_gltf.cpp)
// against glm compiled from the reference's vendored copy, on the exact expressions of
// SceneManager.cpp:224-257 (T*R*S, parent*local) and :285,394-420 (point, normal matrix, tangent).
// Build: oracle/Makefile target `ref` (outputs only into oracle/_ref/).  Exit 0 = bitwise equal.
#include <glm/glm.hpp>
#include <glm/gtc/matrix_transform.hpp>
#include <glm/gtc/quaternion.hpp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" void m2s_debug_node_xform(const float trs[10], const float p[3], const float n[3], const float t[3], float out[25]);

static uint64_t st = 0x4D32535F5345454Full;
static float urand() {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0)) * 2.0f - 1.0f;
}

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 100000;
    for (long i = 0; i < n; i++) {
        float trs[10], p[3], nr[3], tg[3];
        for (auto& v : trs) v = urand() * 3.0f;
        float ql = sqrtf(trs[3] * trs[3] + trs[4] * trs[4] + trs[5] * trs[5] + trs[6] * trs[6]);
        for (int k = 3; k < 7; k++) trs[k] /= ql;
        for (int k = 7; k < 10; k++) trs[k] = 0.25f + fabsf(trs[k]);
        for (int k = 0; k < 3; k++) { p[k] = urand() * 5; nr[k] = urand(); tg[k] = urand(); }
        float out[25];
        m2s_debug_node_xform(trs, p, nr, tg, out);
        glm::mat4 T = glm::translate(glm::mat4(1.0f), glm::vec3(trs[0], trs[1], trs[2]));
        glm::quat q(trs[6], trs[3], trs[4], trs[5]);
        glm::mat4 R = glm::mat4_cast(q);
        glm::mat4 S = glm::scale(glm::mat4(1.0f), glm::vec3(trs[7], trs[8], trs[9]));
        glm::mat4 local = T * R * S;
        glm::mat4 world = glm::mat4(1.0f) * local;
        glm::vec3 wp = glm::vec3(world * glm::vec4(p[0], p[1], p[2], 1.0f));
        glm::mat3 nm = glm::transpose(glm::inverse(glm::mat3(world)));
        glm::vec3 wn = glm::normalize(nm * glm::vec3(nr[0], nr[1], nr[2]));
        glm::vec3 wt = glm::normalize(glm::mat3(world) * glm::vec3(tg[0], tg[1], tg[2]));
        float ref[25];
        for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) ref[c * 4 + r] = world[c][r];
        ref[16] = wp.x; ref[17] = wp.y; ref[18] = wp.z; ref[19] = wn.x; ref[20] = wn.y; ref[21] = wn.z; ref[22] = wt.x; ref[23] = wt.y; ref[24] = wt.z;
        if (memcmp(out, ref, sizeof ref) != 0) {
            for (int k = 0; k < 25; k++) if (memcmp(&out[k], &ref[k], 4)) printf("MISMATCH at %ld [%d]: loader %.9g glm %.9g\n", i, k, out[k], ref[k]);
            return 1;
        }
    }
    printf("OK %ld\n", n);
    return 0;
}
