"""-m gpu: the HIP path against what the REFERENCE produced.

tests/golden/ref_host/pipe_* are outputs of the reference's own host code + shaders run on a software GL
(oracle/ref_pipeline_check.cpp; generator tests/golden/make_ref_golden.py).  The whole product path —
m2s_load_glb -> m2s_upload_scene -> m2s_convert -> m2s_download / m2s_export_ply — must reproduce them:
counter and record count exactly, floats within the parity tolerance, .ply rows within the same tolerance."""
import os

import numpy as np
import pytest

import refhost
from mesh2splat_amd import gltf_io
from mesh2splat_amd.converter import Converter
from mesh2splat_amd.scene import reference_cap
from parity import assert_ply_rows_match, assert_records_match

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")


@pytest.mark.parametrize("pipeline", ["auto", "multipass", "team"])
@pytest.mark.parametrize("name,R", [("mixed_trs", 16), ("soup", 32)])
def test_hip_matches_reference_pipeline_golden(tmp_path, hiplib, name, R, pipeline):
    with open(os.path.join(GOLD, f"pipe_{name}_R{R}.records.bin"), "rb") as f:
        ref = refhost.parse_pipeline_dump(f.read())
    scene = gltf_io.load_glb(os.path.join(GOLD, f"pipe_{name}.glb"))
    c = Converter(0)
    c.set_pipeline(pipeline)
    c.upload_scene(scene)
    c.set_max_gaussians(-1)                      # the reference's cap formula (ConversionPass.cpp:21-24)
    total = c.convert(R)
    rec = c.download()
    assert total == ref["counter"] and reference_cap(R, scene.n_meshes) == ref["max_gaussians"]
    assert_records_match(rec, ref["records"], f"{name} R={R} vs reference pipeline")
    ply = os.path.join(GOLD, f"pipe_{name}_R{R}.ply")
    if os.path.exists(ply):
        mine = str(tmp_path / "m.ply")
        c.export_ply(mine, 1, 0.65)
        a, b = open(mine, "rb").read(), open(ply, "rb").read()
        ha, hb = a.index(b"end_header\n") + 11, b.index(b"end_header\n") + 11
        assert a[:ha] == b[:hb] and len(a) == len(b)
        ra, rb = np.frombuffer(a[ha:], np.float32).reshape(-1, 19), np.frombuffer(b[hb:], np.float32).reshape(-1, 19)
        assert_ply_rows_match(ra, rb, f"{name} R={R} .ply vs the reference's file")    # the 1e-4 bar on the records, propagated through the row formulas
    c.close()
