"""-m gpu: the short correctly rounded sequences of mesh2splat_amd/csrc/m2s_exact.h against the compiler's IEEE expansions, by
exhaustion on the GPU the suite runs on (tests/exact_math/exact_math_check.hip includes the shipped header).

  rcp_rn   every |x| in [2^-64, 2^64], both signs      sqrt_rn   every x in [2^-96, 2^100] (+ rcp_rn(sqrt_rn(x)) vs 1.0f / sqrtf(x))
  div_rn   divisor significands: both ends of [1, 2) and random slices, each against ALL 2^23 dividend significands (1.2e11 pairs),
           + 2^24 random pairs over the whole guarded exponent range; M2S_TEST_DIVALL=1: all 2^23 divisors (7.0e13 pairs, ~45 s —
           the run recorded in profiles/r06/exact_math_exhaustive.jsonl)
The reciprocal and the square root start from hardware seeds (v_rcp_f32, v_rsq_f32), so this is the proof of those two, and it is
re-run wherever the suite runs; the division identity is seed-free (tests/test_round6_math.py has it on the CPU as well)."""
import json
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "exact_math", "_build", "exact_math_check")


def run(*args, timeout=600):
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(EXE))], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def shipped(lines):
    return [ln for ln in lines if "candidate" in ln and "(shipped)" in ln["candidate"]]


def test_reciprocal_every_operand_of_its_range():
    lines = run("rcp")
    s = shipped(lines)
    assert len(s) == 1 and s[0]["mismatches"] == 0, s
    alone = [ln for ln in lines if ln["candidate"] == "v_rcp_f32 alone"][0]
    assert alone["mismatches"] > 0          # (the seed alone is NOT correctly rounded: the check can tell the difference)


def test_square_root_every_operand_of_its_range_and_the_reciprocal_length():
    lines = run("sqrt")
    s = shipped(lines)
    assert len(s) == 1 and s[0]["mismatches"] == 0, s
    comp = [ln for ln in lines if ln["candidate"].startswith("rcp_rn(sqrt_rn(x))")][0]
    assert comp["mismatches"] == 0, comp
    assert [ln for ln in lines if ln["candidate"] == "v_sqrt_f32 alone"][0]["mismatches"] > 0


def test_division_divisor_slices_against_all_dividends():
    if os.environ.get("M2S_TEST_DIVALL") == "1":
        lines = run("divall", 900, timeout=1200)
        assert lines[-1]["complete"] is True and lines[-1]["divisors_done"] == 1 << 23
        assert all(ln["mismatches"] == 0 for ln in shipped(lines)), lines
        return
    rng = random.Random(0xD1F)
    starts = [0, (1 << 23) - 4096] + [rng.randrange(0, (1 << 23) - 1024) for _ in range(6)]
    for k, m0 in enumerate(starts):
        n = 4096 if k < 2 else 1024
        lines = run("div", m0, n)
        assert lines[-1]["complete"] is True and lines[-1]["divisors_done"] == n
        s = shipped(lines)
        assert len(s) == 2 and all(ln["mismatches"] == 0 for ln in s), (m0, s)


# ---- the guards of m2s_exact.h on whole scenes ------------------------------------------------------------------------------------
def _scaled(scene, k):
    """The scene with every position (and so every bounding box) multiplied by 2^k: exact in fp32, and every quantity the conversion derives
    from positions scales exactly with it as long as nothing under- or overflows — the bounding-box-normalised coordinates, hence the
    rasterisation, do not change at all."""
    import copy
    import numpy as np
    from mesh2splat_amd.scene import Mesh, Scene
    f = np.float32(2.0) ** np.float32(k)
    meshes = []
    for m in scene.meshes:
        v = m.vertices.copy()
        v[:, 0:3] *= f
        meshes.append(Mesh(m.name, v, m.base_color, copy.deepcopy(m.textures)))
    return Scene(meshes)


def test_power_of_two_scaling_fast_sequences_and_ieee_fallback_give_the_same_bits(hiplib, oracle):
    """A scene scaled by 2^k converts to the same records with positions and Scale.xy scaled by 2^k — bit for bit, because scaling by a
    power of two commutes with every rounding (while nothing under- or overflows).  At k = -45 every squared edge length (2^-123 ..
    2^-100) lies below sqrt_rn's range and the whole scene takes the compiler's IEEE square roots and reciprocals; at k = 0 and
    +-20 it takes the short sequences: equal bits across the scales = the two paths agree on every triangle of a real scene.
    Further out exact scaling ends (squares in the denormals at 2^-65, the Jacobian's squares beyond 2^127 at 2^58) and so do
    div_rn's and sqrt_rn's upper ranges: those two scenes are checked against the oracle instead."""
    import numpy as np
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    from mesh2splat_amd.scene import Scene
    from parity import assert_records_match
    a = synth.random_soup(6000, seed=77, tri_size=0.03, textures=synth.procedural_textures(64, seed=5), name="a")
    b = synth.cube_sphere(24, tex_size=64, seed=9)
    base = _scaled(Scene(a.meshes + b.meshes), 0)        # (fresh meshes: the cumulative bounding boxes computed the same way at every scale)
    R = 300
    c = Converter(0)
    c.set_max_gaussians(0)
    c.upload_scene(base)
    total0 = c.convert(R)
    rec0 = c.download().copy()
    assert total0 == oracle.convert(base, R, cap=0, count_only=True)[0] and total0 > 20000
    for k in (20, -20, -45):
        c.upload_scene(_scaled(base, k))
        assert c.convert(R) == total0, k
        rec = c.download()
        want = rec0.copy()
        f = np.float32(2.0) ** np.float32(k)
        want[:, 0:3] *= f            # position
        want[:, 8:10] *= f           # Scale.xy (the UV -> 3D Jacobian's column lengths)
        assert np.array_equal(rec.view(np.uint32), want.view(np.uint32)), (k, int((rec.view(np.uint32) != want.view(np.uint32)).sum()))
    for k in (-65, 58):
        far = _scaled(base, k)
        ototal, orec, _ = oracle.convert(far, R, cap=0)
        c.upload_scene(far)
        assert c.convert(R) == ototal, k
        assert_records_match(c.download(), orec, f"scene scaled by 2^{k}")
    c.close()


def test_vertices_on_the_bounding_box_planes_and_zero_length_edges(hiplib, oracle):
    """Operands the short sequences do not take: a bounding-box-relative coordinate that is exactly 0 (every vertex of an axis-aligned
    box lies on bounding-box planes), zero-length edges and zero-area triangles (lengths 0, a 0/0 normal).  Counts and records as
    the oracle's."""
    import numpy as np
    from mesh2splat_amd.converter import Converter
    from mesh2splat_amd.scene import Mesh, Scene
    from parity import assert_records_match
    rng = np.random.default_rng(5)
    n = 1200
    g = rng.integers(0, 9, (n, 3, 3)).astype(np.float32) / np.float32(8)        # lattice points of the unit cube: most triangles touch a bbox plane
    g[::7, 1] = g[::7, 0]                                                          # every 7th triangle: a zero-length edge
    g[::11, 2] = (g[::11, 0] + g[::11, 1]) * np.float32(0.5)                       # every 11th: three collinear vertices
    v = np.zeros((3 * n, 12), np.float32)
    v[:, 0:3] = g.reshape(-1, 3)
    v[:, 3:6] = (0.0, 0.0, 1.0)
    v[:, 6:10] = (1.0, 0.0, 0.0, 1.0)
    v[:, 10:12] = rng.random((3 * n, 2), dtype=np.float32)
    scene = Scene([Mesh("lattice", v)])
    c = Converter(0)
    c.set_max_gaussians(0)
    for R in (64, 129):
        ototal, orec, _ = oracle.convert(scene, R, cap=0)
        for pipe in ("auto", "multipass", "team"):
            c.set_pipeline(pipe)
            c.upload_scene(scene)
            assert c.convert(R) == ototal, (R, pipe)
            assert_records_match(c.download(), orec, f"lattice R={R} {pipe}")
    c.close()
