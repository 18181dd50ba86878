"""CPU: the loader (m2s_gltf.cpp, following SceneManager.cpp:195-459) on files shaped like the two assets BASELINE.json names —
SciFiHelmet.glb (configs[1]) and Sponza.glb (configs[3]) —, authored by the REFERENCE's own tiny_gltf + stb_image_write
(tests/assets.py -> oracle/_ref/ref_host_check glbwrite2) and loaded by the reference's own SceneManager::loadModel: same bytes.
Small maps here (the CPU suite's time budget); tests/test_gpu_assets.py runs the full-size files through the command line and the
oracle on the GPU box.  VERDICT r5 item 7 / row g."""
import json
import os

import numpy as np
import pytest

import assets
import refhost
from mesh2splat_amd import gltf_io
from test_ref_host import assert_scene_equal

pytestmark = pytest.mark.skipif(not assets.available(), reason="oracle/_ref/ref_host_check not built (needs /root/reference)")
HASHES = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "asset_hashes.json")))


@pytest.mark.parametrize("name, make", [("helmet_like_256", lambda: assets.helmet_like(256)), ("sponza_like_0.25", lambda: assets.sponza_like(0.25))])
def test_loader_on_files_shaped_like_the_named_assets(tmp_path, hiplib, name, make):
    spec = make()
    glb = str(tmp_path / (name + ".glb"))
    sha = assets.author(spec, glb, str(tmp_path))
    assert sha == HASHES[name]["sha256"] and assets.n_triangles(spec) == HASHES[name]["triangles"], "generator or writer changed: regenerate tests/golden/asset_hashes.json"
    ref = refhost.load_scene(glb, str(tmp_path))
    mine = gltf_io.load_glb(glb)
    assert_scene_equal(ref, mine)
    if name.startswith("helmet"):
        assert mine.n_meshes == 1 and mine.n_triangles == 70074                       # the real file's triangle count, one primitive
        uv = mine.meshes[0].vertices[:, 10:12]
        assert uv[:, 0].max() > 1.0 and uv[:, 1].min() < 0.0                           # charts outside [0, 1]: REPEAT
        assert set(mine.meshes[0].textures) == {"baseColorTexture", "normalTexture", "metallicRoughnessTexture"}   # occlusion is ignored, as by the reference
        assert len(np.unique(mine.meshes[0].vertices[:, 9])) == 2                      # tangents of both handednesses
    else:
        assert mine.n_meshes == 103 and [m.name for m in mine.meshes[:2]] == ["mesh_0", "mesh_1"]   # "<mesh name>_<counter>", SceneManager.cpp
        kinds = [len(m.textures) for m in mine.meshes]
        assert 0 in kinds and 1 in kinds and 3 in kinds
        assert min(m.n_triangles for m in mine.meshes) == 2 and max(m.n_triangles for m in mine.meshes) == 2 * 96 * 96
        for a, b in zip(mine.meshes[:-1], mine.meshes[1:]):                                       # cumulative bounding boxes (Q1)
            assert np.all(np.asarray(b.bbox_min) <= np.asarray(a.bbox_min)) and np.all(np.asarray(b.bbox_max) >= np.asarray(a.bbox_max))


def test_one_interleaved_view_gives_the_same_scene_as_separate_views(tmp_path, hiplib):
    """flags 2: ONE vertex view of byteStride 48.  The reference's getBufferData (SceneManager.cpp:50-61) ignores byteStride and reads
    such a file as if it were tightly packed; this loader follows the FILE (tests/test_ref_host.py::test_interleaved_views_...), so the
    check is against the same scene written with separate views — which IS what the reference's loader returns for that file."""
    spec = assets.helmet_like(64)
    a, b = str(tmp_path / "separate.glb"), str(tmp_path / "interleaved.glb")
    assets.author(spec, a, str(tmp_path))
    assets.author(dict(spec, flags=2), b, str(tmp_path))
    sa, sb = gltf_io.load_glb(a), gltf_io.load_glb(b)
    assert np.array_equal(np.ascontiguousarray(sa.meshes[0].vertices).view(np.uint32), np.ascontiguousarray(sb.meshes[0].vertices).view(np.uint32))
    assert_scene_equal(refhost.load_scene(a, str(tmp_path)), sb)
