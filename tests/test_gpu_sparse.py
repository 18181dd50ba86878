"""-m gpu: the sparse form of the single-pass kernel (k_sparse, m2s_sparse.hip) — meshes with more triangles than fragments.
Its tier-1 test may only drop triangles that the exact arithmetic leaves without a fragment, so everything here is compared
with the oracle (counter, records) and, bit for bit, with the workgroup-cooperative kernel on the same scene."""
import numpy as np
import pytest

from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
from mesh2splat_amd.scene import Mesh, Scene
from parity import assert_records_match

pytestmark = pytest.mark.gpu


def convert_with(pipeline, scene, R, cap=0, tri_range=None):
    c = Converter(0)
    c.set_pipeline(pipeline)
    if tri_range:
        c.set_triangle_range(*tri_range)
    c.upload_scene(scene)
    c.set_max_gaussians(cap)
    total = c.convert(R)
    total2 = c.convert(R)             # the second conversion reads the XCD band bases the first one left behind
    rec = c.download()
    ran = c.last_pipeline
    c.close()
    assert total == total2
    return total, rec, ran


def check(scene, R, oracle, what, cap=0, expect="sparse"):
    total, rec, ran = convert_with("sparse", scene, R, cap)
    assert ran == expect, (what, ran)
    ototal, orec, _ = oracle.convert(scene, R, cap=cap, n_threads=8)
    assert total == ototal, (what, total, ototal)
    assert_records_match(rec, orec, what)
    t2, rec2, ran2 = convert_with("team", scene, R, cap)
    assert t2 == total and np.array_equal(rec.view(np.uint32), rec2.view(np.uint32)), f"{what}: sparse and team kernels differ"
    return total


@pytest.mark.parametrize("n,R,tex", [(128, 256, 256), (128, 97, 64), (150, 450, 0)])
def test_sub_pixel_sphere(hiplib, oracle, n, R, tex):
    """0.87 / 0.12 / 2.0 fragments per triangle: most, few and nearly none of the triangles are dropped by tier 1."""
    check(synth.cube_sphere(n, tex_size=tex), R, oracle, f"sphere n={n} R={R}")


@pytest.mark.parametrize("seed,R,size", [(1, 333, 0.004), (2, 1024, 0.002), (3, 64, 0.02)])
def test_sub_pixel_soup(hiplib, oracle, seed, R, size):
    """Random triangles: every projection axis, both windings, slivers, uv outside [0, 1]."""
    scene = synth.random_soup(200_000, seed=seed, tri_size=size, textures=synth.procedural_textures(64, seed))
    check(scene, R, oracle, f"soup seed={seed} R={R}")


def test_flattened_soup_near_axis_ties(hiplib, oracle):
    """Near-degenerate triangles and near-ties of the projection axis: tier 1 must keep what it cannot decide."""
    scene = synth.random_soup(200_000, seed=12, tri_size=0.01)
    scene.meshes[0].vertices[:, 2] *= np.float32(1e-3)
    v = scene.meshes[0].vertices.reshape(-1, 3, 12)
    v[::7, :, 2] = v[::7, :, 0] - v[::7, :, 1]          # normals with |nx| = |ny| = |nz| up to rounding
    m = scene.meshes[0]
    scene = Scene([Mesh(name=m.name, vertices=m.vertices, base_color=m.base_color, textures=m.textures)])   # new bounding box
    check(scene, 512, oracle, "flattened soup")


def test_several_meshes_cumulative_bbox(hiplib, oracle):
    """Workgroups that straddle a mesh boundary skip tier 1; later meshes see a larger cumulative bounding box."""
    meshes = []
    for k in range(3):
        v = synth.cube_sphere_vertices(80, radius=1.0, center=(2.5 * k, 0.0, 0.0))
        meshes.append(Mesh(name=f"s{k}", vertices=v, base_color=(1.0, 0.9, 0.8, 1.0), textures=synth.procedural_textures(64, 5 + k)))
    check(Scene(meshes), 160, oracle, "three spheres")


def test_big_triangles_among_sub_pixel_ones(hiplib, oracle):
    """Triangles beyond an 8 x 8 pixel box are only counted by k_sparse and emitted by the second stage; order preserved."""
    soup = synth.random_soup(190_000, seed=21, tri_size=0.003).meshes[0].vertices
    quad = synth.unit_quad(stride=12).meshes[0].vertices.copy()
    quad[:, 0:2] = quad[:, 0:2] * 0.3 + 0.2
    mid = synth.random_soup(200, seed=22, tri_size=0.08).meshes[0].vertices      # (at most 256 deferred triangles: more would send AUTO to the multi-pass pipeline)
    v = np.concatenate([soup[:300_000], quad, soup[300_000:], mid], 0)
    scene = Scene([Mesh(name="mix", vertices=v, base_color=(1, 1, 1, 1), textures=synth.procedural_textures(32, 3))])
    check(scene, 400, oracle, "mixed sizes")


def test_cap_and_triangle_range(hiplib, oracle):
    scene = synth.cube_sphere(128, tex_size=32)
    total, rec, ran = convert_with("sparse", scene, 256, cap=50_000)
    ototal, orec, _ = oracle.convert(scene, 256, cap=50_000, n_threads=8)
    assert ran == "sparse" and total == ototal > 50_000 and len(rec) == 50_000
    assert_records_match(rec, orec, "cap")
    # a shard (multi-GPU path): triangles [10 000, 10 000 + 180 000) of the 196 608
    total, rec, ran = convert_with("sparse", scene, 256, cap=0, tri_range=(10_000, 180_000))
    ototal, orec, _ = oracle.convert(scene, 256, cap=0, tri_first=10_000, tri_count=180_000, n_threads=8)
    assert ran == "sparse" and total == ototal
    assert_records_match(rec, orec, "shard")


def test_falls_back_when_a_workgroup_overflows(hiplib, oracle):
    """A dense scene forced through k_sparse: every triangle is larger than an 8 x 8 pixel box, hence deferred; the conversion
    falls through to another pipeline (and the decision is remembered)."""
    scene = synth.cube_sphere(128, tex_size=32)
    total, rec, ran = convert_with("sparse", scene, 2048, cap=0)
    ototal = oracle.convert(scene, 2048, cap=0, count_only=True, n_threads=8)[0]
    assert ran in ("team", "multipass") and total == ototal


def tessellated_plane(n: int) -> Scene:
    """n x n cells of two right triangles in the plane z = 0, uv = xy: at R = 5 n every triangle has a 5 x 5 pixel box and 10 or
    15 fragments — small enough for k_sparse's 8 x 8 masks, and 512 of them hold ~6400 entries: more than its stream of 2560."""
    g = np.arange(n, dtype=np.float64) / n
    x0, y0 = np.meshgrid(g, g, indexing="xy")
    x0, y0 = x0.ravel(), y0.ravel()
    x1, y1 = x0 + 1.0 / n, y0 + 1.0 / n
    P = np.stack([np.stack([x0, y0], -1), np.stack([x1, y0], -1), np.stack([x1, y1], -1),
                  np.stack([x0, y0], -1), np.stack([x1, y1], -1), np.stack([x0, y1], -1)], 1)       # (cells, 6, 2)
    v = np.zeros((P.shape[0] * 6, 12), np.float32)
    v[:, 0:2] = P.reshape(-1, 2)
    v[:, 5] = 1.0
    v[:, 6] = 1.0
    v[:, 9] = 1.0
    v[:, 10:12] = P.reshape(-1, 2)
    return Scene([Mesh(name="plane", vertices=v, base_color=(1, 1, 1, 1), textures=synth.procedural_textures(64, 5))])


def test_entry_stream_overflow_falls_back_to_team_at_once(hiplib, oracle):
    """error 2 of k_sparse (a workgroup's entries do not fit its LDS stream) in the SECOND round of every workgroup —
    while its waves hold rounds they claimed ahead.  The abandoned rounds are never counted; sibling waves and successor
    workgroups must not wait for them (they used to spin for seconds), the host repeats the conversion with k_fused2, whose
    stream holds these workgroups' ~3200 entries, and remembers it for this and every larger R."""
    import time
    scene = tessellated_plane(300)                      # 180 000 triangles: 64-triangle batches, k_sparse is available
    R = 1500
    c = Converter(0)
    c.set_pipeline("sparse")
    c.set_resolution_hint(64)                           # (the upload's own preparation stays out of the way)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    t0 = time.perf_counter()
    total = c.convert(R)
    dt = time.perf_counter() - t0
    assert c.last_pipeline == "team", c.last_pipeline
    assert dt < 0.25, f"the fallback took {dt:.3f} s: somebody waited out a spin limit"
    rec = c.download().copy()
    t0 = time.perf_counter()
    assert c.convert(R + 4) > total and c.last_pipeline == "team"      # remembered for the scene: no second discovery
    assert time.perf_counter() - t0 < 0.25
    c.close()
    ototal, orec, _ = oracle.convert(scene, R, cap=0, n_threads=8)
    assert total == ototal
    assert_records_match(rec, orec, "plane")
    t2, rec2, ran2 = convert_with("team", scene, R, 0)
    assert ran2 == "team" and t2 == total and np.array_equal(rec.view(np.uint32), rec2.view(np.uint32))


def test_auto_picks_sparse_and_async_submissions(hiplib, oracle):
    scene = synth.cube_sphere(128, tex_size=64)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    total = c.convert(140)             # 0.26 fragments per triangle (AUTO hands a mesh of this size to k_sparse below 0.5)
    assert c.last_pipeline == "sparse"
    want = c.download()
    ototal, orec, _ = oracle.convert(scene, 140, cap=0, n_threads=8)
    assert total == ototal
    assert_records_match(want, orec, "auto")
    for depth in (1, 3):
        for _ in range(depth):
            c.submit(140)
        for _ in range(depth):
            assert c.wait() == total
    assert c.last_pipeline == "sparse"
    assert np.array_equal(c.download().view(np.uint32), want.view(np.uint32))
    assert c.convert(1024) == oracle.convert(scene, 1024, cap=0, count_only=True, n_threads=8)[0]      # 14 fragments per triangle
    assert c.last_pipeline != "sparse"
    c.close()


@pytest.mark.parametrize("pipeline", ["team", "sparse"])
def test_bands_of_equal_work_on_uneven_density(hiplib, oracle, pipeline):
    """The second conversion at an R runs in XCD bands cut (k_pick_bands) from what the first one recorded: eight runs of
    workgroups of equal estimated work.  Here the fragments sit in the first third of the triangle list (the cuts are far from
    equal lengths: the widest band is ~1.5x the narrowest, and the launch is as wide as the widest) — and, turned around, in the last
    third."""
    # (k_sparse: a workgroup's 512 triangles must stay within its LDS stream of 2560 entries)
    dense = synth.random_soup(80_000, seed=31, tri_size=0.02 if pipeline == "team" else 0.012).meshes[0].vertices
    thin = synth.random_soup(170_000, seed=32, tri_size=0.0015).meshes[0].vertices
    for order in ((dense, thin), (thin, dense)):
        v = np.concatenate(order, 0)
        scene = Scene([Mesh(name="uneven", vertices=v, base_color=(1, 1, 1, 1), textures=synth.procedural_textures(64, 9))])
        c = Converter(0)
        c.set_pipeline(pipeline)
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        first = c.convert(300)
        rec1 = c.download().copy()
        assert c.last_pipeline == pipeline
        assert c.convert(300) == first                      # in bands
        rec2 = c.download()
        assert np.array_equal(rec1.view(np.uint32), rec2.view(np.uint32)), "banded and unbanded launches differ"
        for _ in range(3):
            c.submit(300)
        for _ in range(3):
            assert c.wait() == first
        assert np.array_equal(c.download().view(np.uint32), rec1.view(np.uint32))
        c.close()
        ototal, orec, _ = oracle.convert(scene, 300, cap=0, n_threads=8)
        assert first == ototal
        assert_records_match(rec1, orec, f"uneven density, {pipeline}")


def test_positions_left_behind_for_the_depth_sort(hiplib, oracle):
    """m2s_set_keep_positions (BASELINE config 5: conversion, then "final radix sort of the merged splat buffer"): k_sparse also writes the
    records' positions as a 16-byte plane, the first m2s_sort_by_depth builds its keys from it — same records, same sorted order as
    without; a capped conversion keeps the first `cap` positions; a conversion with deferred (big) triangles does not claim the plane."""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import camera
    view = camera.look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1))

    def run(scene, R, cap, keep):
        c = Converter(0)
        c.set_pipeline("sparse")
        c.set_keep_positions(keep)
        c.upload_scene(scene)
        c.set_max_gaussians(cap)
        total = c.convert(R)
        ready = c.positions_ready
        rec = c.download()
        srt = c.sort_by_depth(view)
        after = c.positions_ready
        t2 = c.convert(R - 6)
        ready2 = c.positions_ready
        ran = c.last_pipeline
        c.close()
        return total, rec, srt, ready, after, ready2, ran, t2

    scene = synth.cube_sphere(128, tex_size=64)
    for cap in (0, 50_000):
        t0, rec0, srt0, ready0, after0, ready0b, ran0, _ = run(scene, 256, cap, False)
        t1, rec1, srt1, ready1, after1, ready1b, ran1, _ = run(scene, 256, cap, True)
        assert ran0 == ran1 == "sparse" and t0 == t1 and (cap == 0 or len(rec1) == cap)
        assert not ready0 and after0 and not ready0b        # without: the plane is the first SORT's by-product and a new conversion invalidates it
        assert ready1 and after1 and ready1b                # with: every conversion leaves it behind
        assert np.array_equal(rec0.view(np.uint32), rec1.view(np.uint32)) and np.array_equal(srt0.view(np.uint32), srt1.view(np.uint32))
    # deferred triangles go through k_emit_big, which does not write the plane: not claimed (the sort builds it as before)
    soup = synth.random_soup(190_000, seed=21, tri_size=0.003).meshes[0].vertices
    quad = synth.unit_quad(stride=12).meshes[0].vertices.copy()
    quad[:, 0:2] = quad[:, 0:2] * 0.3 + 0.2
    mix = Scene([Mesh(name="mix", vertices=np.concatenate([soup[:300_000], quad, soup[300_000:]], 0), base_color=(1, 1, 1, 1), textures={})])
    t0, rec0, srt0, ready0, _, _, ran0, _ = run(mix, 400, 0, False)
    t1, rec1, srt1, ready1, after1, _, ran1, _ = run(mix, 400, 0, True)
    assert ran1 == "sparse" and not ready1 and after1 and t0 == t1
    assert np.array_equal(rec0.view(np.uint32), rec1.view(np.uint32)) and np.array_equal(srt0.view(np.uint32), srt1.view(np.uint32))
