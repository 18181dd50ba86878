"""CPU: the reference's conversion path — its own host code and its UNMODIFIED shaders — on a real OpenGL implementation
(Mesa llvmpipe, brought up without X through the DRI software-rasteriser interface: oracle/ref_gl_boot.c), against the oracle.

What GL's fixed function leaves implementation-defined (VERDICT r1 "what's missing" 3) is measured here, not assumed:
  * rasterisation — fragment count AND the set of covered pixels per triangle: identical to the pinned integer rasteriser on
    every scene below (and the counter of the full C3 workload, 2 738 368, is the same number: bench.py runs that as
    cpu_baseline, tools/ref_gl_report.py records it);
  * attribute interpolation — positions agree to ~4e-5 of the coordinate range (llvmpipe interpolates with plane equations);
  * flat outputs of the geometry shader — Scale bit-identical, quaternion within 1 ulp;
  * texture filtering — llvmpipe's default sampler blends RGBA8 texels with 8-bit weights; with its fp32 path
    (GALLIVM_PERF=no_aos_sampling) a magnified texture agrees with the pinned bilinear filter to 2e-7;
  * glGenerateMipmap — Mesa rounds the 2x2 average differently in ties: ~25 % of the level-1 texels differ by 1/255 from the
    pinned round-half-up rule, so blended mip levels differ by up to ~2e-2 on high-contrast noise.  Implementation-defined.
Skipped when oracle/_ref/ref_gl_check is absent (no /root/reference at build time) or no GL context can be created."""
import numpy as np
import pytest

import refgl
from mesh2splat_amd import synth

pytestmark = pytest.mark.skipif(not refgl.available(), reason="oracle/_ref/ref_gl_check not built")


def _cmp(scene, R, oracle, **kw):
    r = refgl.compare(scene, R, oracle, **kw)
    if r is None:
        pytest.skip("no software GL context on this machine")
    return r


@pytest.mark.parametrize("name,R", [("quad", 64), ("right_triangle", 64), ("sphere", 128), ("grid", 96), ("soup", 200), ("soup_fine", 1000)])
def test_rasteriser_matches_llvmpipe_pixel_for_pixel(oracle, name, R):
    scene = {"quad": lambda: synth.unit_quad(),
             "right_triangle": lambda: _one_triangle(),
             "sphere": lambda: synth.cube_sphere(6, tex_size=64),
             "grid": lambda: synth.sphere_grid(2, n=4, tex_size=16),
             "soup": lambda: synth.random_soup(300, seed=5),
             "soup_fine": lambda: synth.random_soup(2000, seed=11, tri_size=0.02)}[name]()
    r = _cmp(scene, R, oracle)
    assert r["gl_counter"] == r["oracle_counter"], r
    assert r["pixels_only_gl"] == 0 and r["pixels_only_oracle"] == 0, r
    if name == "quad":
        assert r["gl_counter"] == 4096                    # K-1
    assert r["scale"]["max_abs"] <= 1e-6                  # flat GS output: bit-identical on most scenes, 1 ulp at worst
    assert r["rotation"]["max_abs"] <= 2.5e-7
    assert r["position"]["max_abs"] <= 2e-4


def _one_triangle():
    q = synth.unit_quad()
    m = q.meshes[0]
    m.vertices = m.vertices[:3].copy()
    return q


def test_bilinear_filter_matches_llvmpipe_fp32_sampler(oracle):
    """Magnification = level 0 only (no mip generation, no LOD blend involved): the pinned bilinear filter against llvmpipe's
    fp32 texture path."""
    r = _cmp(synth.unit_quad(synth.procedural_textures(64)), 256, oracle, float_sampler=True)
    assert r["gl_counter"] == r["oracle_counter"] == 65536
    assert r["color"]["max_abs"] <= 5e-7 and r["pbr"]["max_abs"] <= 5e-7 and r["normal"]["max_abs"] <= 2e-6, r


def test_default_sampler_and_mipmaps_differ_within_known_bounds(oracle):
    """llvmpipe's default (8-bit) sampler and Mesa's glGenerateMipmap: the measured deviation stays within what those two
    known differences explain (<= 8/255 on hash-noise textures)."""
    r = _cmp(synth.cube_sphere(6, tex_size=64), 40, oracle)
    assert r["gl_counter"] == r["oracle_counter"]
    assert r["color"]["max_abs"] <= 8 / 255 and r["color"]["mean_abs"] <= 1e-2, r
    g = refgl.run(synth.unit_quad(synth.procedural_textures(64)), 16, want_mips=True)
    chain, offs, n = oracle.build_mips(synth.procedural_textures(64)["baseColorTexture"])
    assert g["mips"] is not None and len(g["mips"]) == n == 5
    for l, m in enumerate(g["mips"]):
        o = chain[int(offs[l]): int(offs[l]) + m.shape[0] * m.shape[1]].reshape(m.shape)
        d = np.abs(m.astype(int) - o.astype(int))
        assert d.max() <= (0 if l == 0 else 2), (l, d.max())
    lvl1 = np.abs(g["mips"][1].astype(int) - chain[int(offs[1]): int(offs[1]) + 32 * 32].reshape(32, 32, 4).astype(int))
    assert 0.15 < (lvl1 > 0).mean() < 0.35               # the tie cases of the 2x2 average


def test_c4_standin_counter_and_coverage_on_llvmpipe(oracle):
    """BASELINE config 4's stand-in at its stated size (64 meshes, 248 832 triangles, R = 1024): the reference's host code and
    unmodified shaders on a real GL give the oracle's counter (6 612 408 < the 7 M cap) and the oracle's covered pixels."""
    scene = synth.sponza_standin(32)
    g = refgl.run(scene, 1024)
    if g is None:
        pytest.skip("no software GL context on this machine")
    total, _, keys = oracle.convert(scene, 1024, cap=0, want_keys=True, n_threads=8)
    assert g["counter"] == total == 6_612_408 and g["cap"] == 7_000_000
    assert np.array_equal(np.sort(refgl.coverage_keys(scene, g["coverage"])), np.sort(keys))


def test_minification_matches_llvmpipe_once_its_lod_approximation_is_applied(oracle):
    """Full-record agreement on the MINIFICATION path (tools/ref_gl_decompose.py, profiles/r03/ref_gl_decomposition.json).
    The 2e-2 between the pinned trilinear filter and llvmpipe on noise textures decomposes into (a) llvmpipe's level of detail
    lambda = 0.5 fast_log2(rho^2) with a piecewise-LINEAR log2 (up to 0.043 below log2 rho), (b) its 8-bit filter weights,
    (c) glGenerateMipmap's rounding of ties, (d) its plane-equation interpolation of the texture coordinates.  Remove (b) with
    llvmpipe's fp32 sampler, (c) with textures whose 2x2 averages are exact integers through level 3, (d) with a quad whose
    coordinates are exact, and apply (a) to the oracle through its DIAGNOSTIC switch: what is left is 3e-7 — the pinned level
    selection, texel addressing, bilinear weights and level blend ARE what a real GL computes; only its log2 is approximate."""
    base = synth.procedural_textures(64)
    tie_free = {k: (v & np.uint8(0xC0)) for k, v in base.items()}          # multiples of 64: levels 1..3 are exact
    L = oracle.lib()
    try:
        L.orc_debug_set_lod_mode(1)
        for R in (24, 12):                                                 # lambda 1.4 (levels 1+2), 2.4 (levels 2+3)
            r = _cmp(synth.unit_quad(tie_free), R, oracle, float_sampler=True)
            assert r["gl_counter"] == r["oracle_counter"] == R * R
            assert r["color"]["max_abs"] <= 2e-6 and r["pbr"]["max_abs"] <= 2e-6 and r["normal"]["max_abs"] <= 4e-6, r
        L.orc_debug_set_lod_mode(0)
        r = _cmp(synth.unit_quad(tie_free), 24, oracle, float_sampler=True)
        assert 1e-3 < r["color"]["max_abs"] < 2e-2                         # the pinned log2 against llvmpipe's: the LOD term alone
    finally:
        L.orc_debug_set_lod_mode(0)
