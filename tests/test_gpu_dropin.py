"""-m gpu: the drop-in seen from the REFERENCE's side, compiled and run.

oracle/_ref/ref_dropin_check = the reference's own SceneManager::loadModel and SceneManager::exportPly (compiled from
/root/reference, exactly as ref_pipeline_check links them) around oracle/ref_dropin/ConversionPassHip.cpp — the replacement body
of ConversionPass::execute that INTEGRATION.md shows — linked against libm2s_hip.so.  Its records and its .ply are compared with
the ALL-reference run of the same .glb (ref_pipeline_check: the reference's ConversionPass.cpp + shaders on the software GL):
counter, cap and SSBO size exact, records and rows within the parity tolerance."""
import os

import numpy as np
import pytest

import refhost
from mesh2splat_amd import gltf_io, synth
from parity import assert_ply_rows_match, assert_records_match

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (refhost.dropin_available() and refhost.pipeline_available()),
                                 reason="oracle/_ref/ref_dropin_check / ref_pipeline_check not built (need /root/reference at build time)")]
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")


def read_ply_rows(path):
    """format-1 .ply -> (header bytes, (n, 19) float32 rows)"""
    a = open(path, "rb").read()
    h = a.index(b"end_header\n") + 11
    return a[:h], np.frombuffer(a[h:], np.float32).reshape(-1, 19)


def both(glb, R, tmp_path, fmt):
    d_ref, d_hip = str(tmp_path / "ref"), str(tmp_path / "hip")
    os.makedirs(d_ref); os.makedirs(d_hip)
    ref = refhost.run_pipeline(glb, R, d_ref, ply_path=os.path.join(d_ref, "out.ply"), fmt=fmt, std=0.65)
    hip = refhost.run_dropin(glb, R, d_hip, ply_path=os.path.join(d_hip, "out.ply"), fmt=fmt, std=0.65)
    assert hip["counter"] == ref["counter"]
    assert hip["max_gaussians"] == ref["max_gaussians"] and hip["ssbo_bytes"] == ref["ssbo_bytes"]
    assert_records_match(hip["records"], ref["records"], f"{os.path.basename(glb)}: drop-in vs all-reference records")
    return ref, hip, os.path.join(d_ref, "out.ply"), os.path.join(d_hip, "out.ply")


@pytest.mark.parametrize("name,R", [("pipe_soup", 32), ("pipe_mixed_trs", 16)])
def test_dropin_on_the_committed_reference_scenes(tmp_path, name, R):
    """The two scenes whose all-reference outputs are committed (tests/golden/ref_host/pipe_*): textures, several meshes with
    node transforms, missing maps."""
    ref, hip, ply_ref, ply_hip = both(os.path.join(GOLD, name + ".glb"), R, tmp_path, fmt=1)
    (ha, a), (hb, b) = read_ply_rows(ply_hip), read_ply_rows(ply_ref)
    assert ha == hb
    assert_ply_rows_match(a, b, name + " .ply written by the reference's exportPly from the drop-in's records")
    with open(os.path.join(GOLD, f"{name}_R{R}.records.bin"), "rb") as f:      # and against the committed reference output
        gold = refhost.parse_pipeline_dump(f.read())
    assert gold["counter"] == hip["counter"]
    assert_records_match(hip["records"], gold["records"], name + ": drop-in vs committed reference records")


def test_dropin_follows_a_model_switch(tmp_path):
    """two different ONE-mesh models loaded in a row by the same SceneManager.  loadModel's clear + reserve + push_back
    keeps dataMeshAndGlMesh's allocation, so pointer and size are those of the first model; the body must still notice the switch
    (it keys the resident scene on a signature of the loaded model) and convert the SECOND one."""
    a, b = str(tmp_path / "a.glb"), str(tmp_path / "b.glb")
    gltf_io.write_glb(synth.cube_sphere(6, tex_size=16), a, indexed=False)
    gltf_io.write_glb(synth.cube_sphere(9, tex_size=32), b, indexed=False)
    d_ref, d_hip = str(tmp_path / "ref"), str(tmp_path / "hip")
    os.makedirs(d_ref); os.makedirs(d_hip)
    ref_b = refhost.run_pipeline(b, 48, d_ref)
    ref_a = refhost.run_pipeline(a, 48, d_ref, out_path=os.path.join(d_ref, "a.bin"))
    hip = refhost.run_dropin(b, 48, d_hip, load_first=a)
    assert ref_a["counter"] != ref_b["counter"]          # (otherwise the test could not tell the models apart)
    assert hip["counter"] == ref_b["counter"]
    assert_records_match(hip["records"], ref_b["records"], "second model after a model switch")


def test_dropin_on_the_headline_workload(tmp_path):
    """BASELINE config 3 (1 002 252 triangles, 3 x 2048^2 maps, R = 1024) through reference-loader -> HIP -> reference writer."""
    glb = str(tmp_path / "c3.glb")
    gltf_io.write_glb(synth.cube_sphere(289, tex_size=2048), glb, indexed=False)
    ref, hip, ply_ref, ply_hip = both(glb, 1024, tmp_path, fmt=1)
    assert hip["counter"] == 2_738_368
    (ha, a), (hb, b) = read_ply_rows(ply_hip), read_ply_rows(ply_ref)
    assert ha == hb
    assert_ply_rows_match(a, b, "C3 .ply")
