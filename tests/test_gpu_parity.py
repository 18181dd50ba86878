"""-m gpu: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import os
import pytest

from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
from mesh2splat_amd.scene import Mesh, Scene
from parity import assert_records_match

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "multipass", "team"])
def conv(hiplib, request):
    """Every parity test runs through both pipelines: the fused single-pass kernel (with its multi-pass
    fallback for big triangles) and the forced multi-pass pipeline."""
    c = Converter(0)
    c.set_pipeline(request.param)
    yield c
    c.close()


def test_pipelines_bit_identical(hiplib):
    """fused and multi-pass pipelines must produce the same bytes (they share the per-fragment code)."""
    a, b = Converter(0), Converter(0)
    b.set_pipeline("multipass")
    for scene, R in [(synth.cube_sphere(30, tex_size=64), 256), (synth.random_soup(4000, seed=4), 300),
                     (synth.sphere_grid(2, n=7, tex_size=32), 200)]:
        outs = []
        for c in (a, b):
            c.upload_scene(scene)
            c.set_max_gaussians(0)
            n = c.convert(R)
            outs.append((n, c.download()))
        assert outs[0][0] == outs[1][0]
        assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    a.close(); b.close()


def test_prepare_loads_the_kernels_and_changes_nothing(hiplib):
    """m2s_prepare(M2S_PREPARE_KERNELS) (code objects loaded ahead of the first conversion of the process: what the command line
    does on its second thread) is idempotent, needs no scene, and a conversion after it gives the bytes of one without it."""
    a, b = Converter(0), Converter(0)
    a.prepare(Converter.PREPARE_KERNELS)
    a.prepare(Converter.PREPARE_UPLOAD | Converter.PREPARE_EXPORT | Converter.PREPARE_KERNELS)
    scene = synth.cube_sphere(24, tex_size=64)
    outs = []
    for c in (a, b):
        c.set_resolution_hint(200)
        c.upload_scene(scene)
        outs.append((c.convert(200), c.download()))
    assert outs[0][0] == outs[1][0] > 0
    assert np.array_equal(outs[0][1].view(np.uint32), outs[1][1].view(np.uint32))
    a.close(); b.close()


def run_both(conv, oracle, scene, R, cap=None):
    conv.set_triangle_range(0, None)
    conv.upload_scene(scene)
    conv.set_max_gaussians(-1 if cap is None else cap)
    total = conv.convert(R)
    rec = conv.download()
    ototal, orec, _ = oracle.convert(scene, R, cap=cap)
    return total, rec, ototal, orec


def test_unit_quad_c1(conv, oracle):
    """BASELINE config C1 / KAT K-1: unit quad, no textures, R=64 -> exactly 4096 Gaussians."""
    total, rec, ototal, orec = run_both(conv, oracle, synth.unit_quad(), 64)
    assert total == ototal == 4096
    assert_records_match(rec, orec, "quad R=64")
    assert np.all(rec[:, 8:11] == np.float32([1, 1, 1e-7]))


@pytest.mark.parametrize("R", [16, 64, 333, 1024])
def test_unit_quad_textured(conv, oracle, R):
    """Large triangles (wave-cooperative rasteriser) + all three maps, mag and min filtering."""
    scene = synth.unit_quad(synth.procedural_textures(256))
    total, rec, ototal, orec = run_both(conv, oracle, scene, R)
    assert total == ototal == R * R
    assert_records_match(rec, orec, f"textured quad R={R}")


@pytest.mark.parametrize("n,R,tex", [(8, 64, 0), (24, 256, 128), (76, 512, 2048)])      # the last one: BASELINE config 2's stand-in (I-2) at its stated size
def test_cube_sphere(conv, oracle, n, R, tex):
    scene = synth.cube_sphere(n, tex_size=tex)
    total, rec, ototal, orec = run_both(conv, oracle, scene, R)
    assert total == ototal
    assert_records_match(rec, orec, f"sphere n={n} R={R}")


@pytest.mark.parametrize("seed,R", [(1, 64), (2, 257), (3, 1024)])
def test_random_soup(conv, oracle, seed, R):
    """All projection axes, both windings, all longest-edge cases, uv outside [0,1] (REPEAT)."""
    scene = synth.random_soup(3000, seed=seed, textures=synth.procedural_textures(64, seed))
    total, rec, ototal, orec = run_both(conv, oracle, scene, R, cap=0)
    assert total == ototal
    assert_records_match(rec, orec, f"soup seed={seed} R={R}")


def test_multi_mesh_cumulative_bbox(conv, oracle):
    """I-4 style scene: 8 meshes, distinct materials; mesh k's bbox is the AABB of meshes 0..k."""
    scene = synth.sphere_grid(2, n=6, tex_size=64)
    total, rec, ototal, orec = run_both(conv, oracle, scene, 128, cap=0)
    assert total == ototal
    assert_records_match(rec, orec, "sphere grid")


def test_cap_semantics(conv, oracle):
    """Counter keeps counting past the cap (ConversionPass.cpp:56-59); only cap records are stored."""
    scene = synth.cube_sphere(8)
    total, rec, ototal, orec = run_both(conv, oracle, scene, 64, cap=1000)
    assert total == ototal > 1000
    assert rec.shape[0] == 1000 == conv.num_stored
    assert_records_match(rec, orec, "cap")


def test_reference_cap_formula(conv, oracle):
    conv.upload_scene(synth.unit_quad())
    conv.set_max_gaussians(-1)
    assert conv.convert(16) == 256
    assert oracle.reference_cap(16, 1) == 16 * 16 * 6


def test_empty_and_degenerate(conv, oracle):
    """Empty mesh list entry, zero-area and collinear triangles, flat bbox (range 0 -> NaN uv)."""
    v = np.zeros((9, 12), np.float32)
    v[0:3, 0:3] = [[0, 0, 0], [0, 0, 0], [0, 0, 0]]           # zero area
    v[3:6, 0:3] = [[0, 0, 0], [1, 1, 0], [2, 2, 0]]           # collinear
    v[6:9, 0:3] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]           # a real one
    v[:, 3:6] = [0, 0, 1]
    v[:, 6:10] = [1, 0, 0, 1]
    empty = Mesh("empty", np.zeros((0, 12), np.float32))
    scene = Scene([empty, Mesh("m", v)])
    total, rec, ototal, orec = run_both(conv, oracle, scene, 32)
    assert total == ototal > 0
    assert_records_match(rec, orec, "degenerate")
    flat = Mesh("flat", v[6:9].copy(), bbox_min=np.zeros(3, np.float32), bbox_max=np.zeros(3, np.float32))
    total, rec, ototal, orec = run_both(conv, oracle, Scene([flat]), 32)
    assert total == ototal == 0 and rec.shape[0] == 0


def test_triangle_range_shards_concatenate(conv, oracle):
    """Multi-GPU contract: converting triangle ranges independently and concatenating in rank order
    reproduces the unsharded result exactly."""
    scene = synth.sphere_grid(2, n=5, tex_size=32)
    conv.set_triangle_range(0, None)
    conv.upload_scene(scene)
    conv.set_max_gaussians(0)
    conv.convert(96)
    full = conv.download()
    T = scene.n_triangles
    cuts = [0, T // 3, T // 3 + 7, T]
    parts = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        conv.set_triangle_range(a, b - a)
        conv.upload_scene(scene)
        conv.convert(96)
        parts.append(conv.download())
    conv.set_triangle_range(0, None)
    cat = np.concatenate(parts, 0)
    assert cat.shape == full.shape
    assert np.array_equal(cat.view(np.uint32), full.view(np.uint32))


def test_triangle_counts_match_oracle(conv, oracle):
    scene = synth.random_soup(5000, seed=9)
    conv.set_triangle_range(0, None)
    conv.upload_scene(scene)
    conv.set_max_gaussians(0)
    conv.convert(512)
    assert np.array_equal(conv.download_triangle_counts(), oracle.count_per_triangle(scene, 512))


def test_depth_sort_matches_stable_argsort(hiplib, oracle):
    """f-2 / RadixSortPass: key = floatBitsToUint(view-space z), ascending on the raw bits, stable."""
    scene = synth.sphere_grid(2, n=6, tex_size=16)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    n = c.convert(160)
    rec = c.download()
    ang = 0.7
    view = np.array([[np.cos(ang), 0, np.sin(ang), -0.3], [0, 1, 0, -0.2], [-np.sin(ang), 0, np.cos(ang), -4.0], [0, 0, 0, 1]], np.float32)
    got = c.sort_by_depth(view)
    assert got.shape == rec.shape == (n, 24)
    f = np.float32
    z = (f(view[2, 0]) * rec[:, 0] + f(view[2, 1]) * rec[:, 1]) + f(view[2, 2]) * rec[:, 2]
    z = (z + f(view[2, 3])).astype(np.float32)
    order = np.argsort(z.view(np.uint32), kind="stable")
    assert np.array_equal(got.view(np.uint32), rec[order].view(np.uint32))
    assert np.all(z < 0) and np.all(np.diff(-z[order]) >= 0)      # in front of the camera: front to back
    c.close()


def test_depth_sort_position_plane_follows_the_records(hiplib, oracle):
    """The first sort after the records changed reads the positions out of the records and leaves them behind as a compact plane;
    later sorts (another camera every frame) take their keys from it.  The plane must be dropped whenever the records may have
    changed: another conversion into the same buffer (another density: other records at the same address), uploaded records,
    adopted records."""
    f = np.float32

    def want(rec, view):
        z = (f(view[2, 0]) * rec[:, 0] + f(view[2, 1]) * rec[:, 1]) + f(view[2, 2]) * rec[:, 2]
        z = (z + f(view[2, 3])).astype(np.float32)
        return rec[np.argsort(z.view(np.uint32), kind="stable")]

    def views():
        for k, ang in enumerate((0.3, 1.1, 2.0)):
            yield np.array([[np.cos(ang), 0, np.sin(ang), 0.1 * k], [0, 1, 0, -0.2], [-np.sin(ang), 0, np.cos(ang), -4.0 - k], [0, 0, 0, 1]], np.float32)

    scene = synth.sphere_grid(2, n=5, tex_size=16)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    for R in (150, 150, 90, 151):            # same density twice (same records), then fewer records, then about as many again
        c.convert(R)
        rec = c.download()
        for view in views():                 # first view: from the records; the others: from the plane
            assert np.array_equal(c.sort_by_depth(view).view(np.uint32), want(rec, view).view(np.uint32)), R
    loaded = np.ascontiguousarray(rec[::-1][:4000])
    c.upload_records(loaded)                 # other records, possibly at an address seen before
    for view in views():
        assert np.array_equal(c.sort_by_depth(view).view(np.uint32), want(loaded, view).view(np.uint32))
    c.close()


@pytest.mark.parametrize("R,tri_size,n", [(1024, 0.08, 3000), (2048, 0.05, 2500), (512, 0.5, 600), (4096, 0.02, 2000)])
def test_row_walker_counts_mid_size_triangles(conv, oracle, R, tri_size, n):
    """Triangles of 10..300 pixel rows: the division-free row walker (sequential loops), the closed-form spans
    (wave-cooperative paths) and the coverage masks must all give the oracle's per-triangle counts, in both pipelines."""
    scene = synth.random_soup(n, seed=R + n, tri_size=tri_size)
    conv.set_triangle_range(0, None)
    conv.upload_scene(scene)
    conv.set_max_gaussians(0)
    total = conv.convert(R)
    want = oracle.count_per_triangle(scene, R)
    assert total == int(want.sum())
    got = conv.download_triangle_counts()
    assert np.array_equal(got, want)
    rows_hint = np.sqrt(want.max())
    assert rows_hint > 20            # the case really contains mid-size / big triangles


@pytest.mark.parametrize("n", [148, 156, 157])
def test_count_scan_extra_blocks_by_ticket(hiplib, oracle, n):
    """k_count_scan launches the resident workgroups only when a scene has a few more blocks of 256 triangles than the GPU holds
    workgroups (1024 on an MI355X) and hands the extra blocks out by ticket (m2s_emit2.hip, count_block_a / count_block_b): cube-spheres
    of 262 848 / 292 032 / 295 788 triangles = 3 and 117 extra blocks, and 132 (over the limit of an eighth: one workgroup per block, as
    before).  Same bytes as the single-pass kernel, the oracle's count per triangle; twice, so that the tickets were reset."""
    scene = synth.cube_sphere(n, tex_size=64)
    R = 192
    want = oracle.count_per_triangle(scene, R)
    a, b = Converter(0), Converter(0)
    a.set_pipeline("team")
    b.set_pipeline("multipass")
    outs = []
    for c in (a, b):
        c.upload_scene(scene)
        c.set_max_gaussians(0)
    for c in (a, b, b):
        total = c.convert(R)
        assert total == int(want.sum())
        outs.append(c.download())
    assert b.last_pipeline == "multipass"
    assert np.array_equal(b.download_triangle_counts(), want)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    assert np.array_equal(outs[1].view(np.uint32), outs[2].view(np.uint32))
    a.close(); b.close()


def test_many_wave_counted_triangles_per_wave(hiplib, oracle):
    """A low-polygon scene at a high density: every triangle of twelve tilted 6 x 6 patches is taller than 128 pixel rows at R = 1536 —
    k_count_scan's (triangle, chunk) tasks with dozens of wave-counted triangles per wave, several rounds, chunks cut by the patches'
    exactly horizontal edges.  The oracle's count per triangle, the team kernel's bytes."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from tall_probe import scene_of
    scene = scene_of(12)
    R = 1536
    want = oracle.count_per_triangle(scene, R)
    a, b = Converter(0), Converter(0)
    a.set_pipeline("team")
    b.set_pipeline("multipass")
    outs = []
    for c in (a, b):
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        assert c.convert(R) == int(want.sum())
        outs.append(c.download())
    assert b.last_pipeline == "multipass"
    assert np.array_equal(b.download_triangle_counts(), want)
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    a.close(); b.close()
