"""CPU: the oracle against the REFERENCE'S OWN conversion shaders.

oracle/ref_glsl_check.cpp compiles converter{VS,GS,FS}.glsl — rewritten only syntactically by oracle/glsl2cpp.py —
against the reference's vendored glm and executes them; fixed-function GL (rasterisation, varying interpolation,
texture filtering, LOD selection) is supplied by the harness identically to both sides.  What is compared is what
the shaders compute: gl_Position (triplanar bbox-normalised UVs), Scale (UV->3D Jacobian), Quaternion (longest-
edge frame through quat_cast), and the whole 24-float record of the fragment shader.

  * golden — the reference shaders' outputs committed under tests/golden/ref_host/glsl_* ; always runs.
  * live   — larger scenes through the binary; skipped where oracle/_ref was not built.
Bar: bit-exact on every field (the oracle is an operation-for-operation restatement in IEEE fp32)."""
import json
import os

import numpy as np
import pytest

import refhost
from mesh2splat_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")
FIELDS = ("ndc", "scale", "quaternion", "position", "color", "normal", "pbr", "passthrough")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["sphere", "soup", "soup_untextured"])
def test_oracle_matches_reference_shaders_golden(oracle, name):
    with open(os.path.join(GOLD, f"glsl_{name}.report.json")) as f:
        rep = json.load(f)
    with open(os.path.join(GOLD, f"glsl_{name}.scene.bin"), "rb") as f:
        scene = refhost.parse_glsl_scene(f.read())
    T, S, R = scene.n_triangles, rep["samples"], rep["R"]
    assert rep["triangles"] == T
    with open(os.path.join(GOLD, f"glsl_{name}.dump.bin"), "rb") as f:
        gs, fs = refhost.parse_glsl_dump(f.read(), T, S)
    prep = oracle.PreparedScene(scene)
    t = 0
    for mi, m in enumerate(scene.meshes):
        v = m.vertices.reshape(-1, 3, 12)
        for k in range(v.shape[0]):
            ok, ndc, scl, rot = oracle.debug_gs(v[k, 0], v[k, 1], v[k, 2], m.bbox_min, m.bbox_max, R)
            assert np.array_equal(bits(ndc.reshape(-1)), bits(gs[t, 0:6])), (name, t, "gl_Position")
            assert np.array_equal(bits(scl), bits(gs[t, 6:9])), (name, t, "Scale")
            assert np.array_equal(bits(rot), bits(gs[t, 9:13])), (name, t, "Quaternion")
            for s in range(S):
                rec = prep.debug_fs(mi, fs[t, s, 0:12], fs[t, s, 12:15], scl[:2], rot)
                assert np.array_equal(bits(rec), bits(fs[t, s, 15:39])), (name, t, s, rec, fs[t, s, 15:39])
            t += 1
    assert t == T
    # and the report the harness wrote at generation time says the same
    assert rep["quat_sign_flips"] == 0
    for k in FIELDS:
        assert rep[k]["n"] == rep[k]["exact"] > 0, k


def live_cases():
    yield "sphere_tex", synth.cube_sphere(16, tex_size=64), 256
    yield "grid", synth.sphere_grid(2, n=4, tex_size=16), 128
    yield "soup_tex", synth.random_soup(4000, seed=1, textures=synth.procedural_textures(32, 3)), 512
    yield "soup_plain", synth.random_soup(3000, seed=2), 64
    yield "colocated", synth.colocated_spheres(4, n=4, tex_size=8), 100
    yield "quad", synth.unit_quad(), 16


@pytest.mark.skipif(not refhost.glsl_available(), reason="oracle/_ref/ref_glsl_check not built (needs /root/reference)")
@pytest.mark.parametrize("case", list(live_cases()), ids=lambda c: c[0])
def test_oracle_matches_reference_shaders_live(tmp_path, case):
    name, scene, R = case
    rep = refhost.run_glsl_check(scene, R, 4, str(tmp_path))
    assert rep["triangles"] == scene.n_triangles and rep["quat_sign_flips"] == 0
    for k in FIELDS:
        assert rep[k]["n"] == rep[k]["exact"] > 0, (k, rep[k])
