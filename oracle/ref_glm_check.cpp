// oracle/_ref cross-check: the oracle's quat_cast restatement (converterGS.glsl:131-183)
// against glm::quat_cast compiled from the reference's vendored glm (thirdParty/glm).
// Build recipe: oracle/Makefile target `ref` (outputs only into oracle/_ref/).
// Usage: glm_check [n] -> prints "OK n" or the first mismatch; exit code 0/1.
#include <glm/glm.hpp>
#include <glm/gtc/quaternion.hpp>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

extern "C" void orc_quat_cast(const float m9_colmajor[9], float q_xyzw[4]);

static uint64_t sm_state = 0x4D32535F5345454Full;
static uint64_t splitmix() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float urand() { return (float)((splitmix() >> 40) * (1.0 / 16777216.0)) * 2.0f - 1.0f; }

int main(int argc, char** argv) {
    long n = argc > 1 ? atol(argv[1]) : 200000;
    for (long i = 0; i < n; i++) {
        // random orthonormal frame built the way the GS builds it (x, normalize(cross(n,x)), n)
        glm::vec3 a(urand(), urand(), urand()), b(urand(), urand(), urand());
        glm::vec3 x = a / glm::length(a);
        glm::vec3 nn = glm::cross(x, b);
        nn = nn / glm::length(nn);
        glm::vec3 y = glm::cross(nn, x);
        y = y / glm::length(y);
        glm::mat3 M(x, y, nn);
        glm::quat q = glm::quat_cast(M);
        float m9[9];
        for (int c = 0; c < 3; c++) for (int r = 0; r < 3; r++) m9[c * 3 + r] = M[c][r];
        float o[4];
        orc_quat_cast(m9, o);
        float g[4] = { q.x, q.y, q.z, q.w };
        if (memcmp(o, g, 16) != 0) {
            printf("MISMATCH at %ld: oracle (%.9g %.9g %.9g %.9g) glm (%.9g %.9g %.9g %.9g)\n", i, o[0], o[1],
                   o[2], o[3], g[0], g[1], g[2], g[3]);
            return 1;
        }
    }
    printf("OK %ld\n", n);
    return 0;
}
