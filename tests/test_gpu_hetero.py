"""-m gpu: the heterogeneous scene (synth.sponza_like: 64 meshes, 266 840 triangles whose pixel areas span six decades, materials
with three maps / one map / none) at full size against the oracle, under AUTO and under every forced pipeline setting — the
scene class BASELINE config 4 (Sponza) stands for, which no cube-sphere workload covers: 2-triangle planes of half a million
fragments next to sub-pixel foliage in ONE conversion."""
import os

import numpy as np
import pytest

from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
from parity import assert_achieved, assert_records_match

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(HERE)), "gpurun_out")


@pytest.fixture(scope="module")
def hetero():
    return synth.sponza_like()


def test_sponza_like_full_size_against_the_oracle(hiplib, oracle, hetero):
    R = 1024
    ototal, orec, _ = oracle.convert(hetero, R, n_threads=os.cpu_count() or 1)
    assert 4_000_000 < ototal < 7_000_000                      # under the reference's cap: parity is defined
    cnt = oracle.count_per_triangle(hetero, R)
    assert cnt.max() > 500_000 and (cnt == 0).sum() > 50_000 and ((cnt >= 64) & (cnt < 1024)).sum() > 10_000   # the spread this test is for
    conv = Converter(0)
    conv.upload_scene(hetero)
    total = conv.convert(R)
    rec = conv.download()
    assert total == ototal == rec.shape[0]
    frac = assert_records_match(rec, orec, "sponza_like, AUTO (%s)" % conv.last_pipeline)
    assert frac > 0.6
    assert_achieved(rec, orec, "sponza_like", os.path.join(OUT_DIR, "parity_hetero.json"), config="hetero")
    assert np.array_equal(conv.download_triangle_counts(), cnt.astype(np.uint32))
    # every setting: the same bytes (forced single-pass kernels defer the planes to k_emit_big or hand the scene to the multi-pass pipeline)
    for name in ("multipass", "team", "sparse", "lean"):
        c2 = Converter(0)
        c2.set_pipeline(name)
        c2.upload_scene(hetero)
        for _ in range(2):
            assert c2.convert(R) == total
        assert np.array_equal(c2.download().view(np.uint32), rec.view(np.uint32)), (name, c2.last_pipeline)
        c2.close()
    conv.close()


@pytest.mark.parametrize("R", [256, 2048])
def test_sponza_like_other_densities_and_the_cap(hiplib, oracle, hetero, R):
    """R = 2048: 17 M fragments, the reference's 7 M cap cuts the output (the first 7 M in canonical order here); R = 256:
    most cloth and all foliage triangles cover no pixel centre."""
    conv = Converter(0)
    conv.upload_scene(hetero)
    total = conv.convert(R)
    ototal, orec, _ = oracle.convert(hetero, R, n_threads=os.cpu_count() or 1)
    assert total == ototal
    rec = conv.download()
    assert rec.shape[0] == orec.shape[0] == min(total, 7_000_000)
    assert_records_match(rec, orec, "sponza_like R=%d (%s)" % (R, conv.last_pipeline))
    conv.close()


def test_sponza_like_with_three_maps_everywhere_runs_lean_kernels_too(hiplib, oracle):
    """combo_only: every textured material has all three maps, so the lean single-pass kernel is allowed — forced, it shades the
    triangles of at most 8 x 8 pixels itself and defers everything else."""
    scene = synth.sponza_like(combo_only=True, tex_scale=0.25)
    R = 512
    ototal, orec, _ = oracle.convert(scene, R, n_threads=os.cpu_count() or 1)
    ref = None
    for name in ("auto", "lean", "team", "multipass"):
        c = Converter(0)
        c.set_pipeline(name)
        c.upload_scene(scene)
        assert c.convert(R) == ototal
        rec = c.download()
        if ref is None:
            ref = rec
            assert_records_match(rec, orec, "sponza_like combo_only, %s" % name)
        else:
            assert np.array_equal(rec.view(np.uint32), ref.view(np.uint32)), (name, c.last_pipeline)
        c.close()


def test_auto_runs_the_team_kernel_in_small_batches_at_the_lower_edge_of_the_multipass_range(hiplib, oracle):
    """AUTO between 11 and 13 fragments per triangle (run_pass: decide): a uniform mesh is converted by k_fused2 in batches of 40
    triangles (a workgroup's 160 triangles fit its LDS stream) instead of the multi-pass pipeline; same bytes either way.  A scene
    whose mean lies there only because it mixes planes with foliage is not, and neither is a mesh of 14 fragments per triangle."""
    scene = synth.cube_sphere(140, tex_size=256)           # 235 200 triangles, 11.6 fragments each at R = 1024
    R = 1024
    ototal, orec, _ = oracle.convert(scene, R, n_threads=os.cpu_count() or 1)
    assert 11 * scene.n_triangles <= ototal < 13 * scene.n_triangles
    c = Converter(0)
    c.set_resolution_hint(R)
    c.upload_scene(scene)
    for _ in range(2):
        assert c.convert(R) == ototal
        assert c.last_pipeline == "team"
    rec = c.download()
    assert_records_match(rec, orec, "11.6 fragments per triangle, team kernel in batches of 40")
    for _ in range(3):                                      # the asynchronous path takes the same kernel
        c.submit(R)
    for _ in range(3):
        assert c.wait() == ototal
    assert np.array_equal(c.download().view(np.uint32), rec.view(np.uint32))
    c.set_pipeline("multipass")
    assert c.convert(R) == ototal
    assert np.array_equal(c.download().view(np.uint32), rec.view(np.uint32))
    c.close()
    for other in (synth.cube_sphere(127, tex_size=64), synth.sponza_like(tex_scale=0.125)):   # 14.1 per triangle; mixed sizes
        h = Converter(0)
        h.set_resolution_hint(R)
        h.upload_scene(other)
        t = h.convert(R)
        assert h.last_pipeline == "multipass"
        assert t == oracle.convert(other, R, count_only=True)[0]
        h.close()


@pytest.mark.parametrize("world", [3, 8])
def test_sponza_like_shards_concatenate_to_the_unsharded_output(hiplib, hetero, world):
    """The multi-GPU contract on the heterogeneous scene: fragment-balanced triangle ranges (m2s_dist_shard_ranges: the floor's two
    triangles weigh as much as a cloth mesh), each converted by its own upload — ranges that start and end inside meshes, blocks of
    256 counted from the range's first triangle, fine and dense blocks in every shard — concatenated in rank order give the bytes of
    the one-GPU conversion."""
    from mesh2splat_amd import dist as m2d
    R = 1024
    whole = Converter(0)
    whole.set_max_gaussians(0)
    whole.upload_scene(hetero)
    total = whole.convert(R)
    rec = whole.download()
    whole.close()
    plan = m2d.shard_ranges_native(hetero, R, world)
    assert sum(c for _, c in plan) == hetero.n_triangles and len({c for _, c in plan}) > 1     # not an even split
    parts, totals = [], []
    for first, count in plan:
        c = Converter(0)
        c.set_triangle_range(first, count)
        c.upload_scene(hetero)
        c.set_max_gaussians(0)
        totals.append(c.convert(R))
        parts.append(c.download())
        c.close()
    assert sum(totals) == total
    assert max(totals) < 2.5 * total / world                     # fragment-balanced, not triangle-balanced
    assert np.array_equal(np.concatenate(parts).view(np.uint32), rec.view(np.uint32))
