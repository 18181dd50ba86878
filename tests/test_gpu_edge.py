"""-m gpu: edge cases of the C ABI and of the pinned rasteriser on the device."""
import ctypes as C

import numpy as np
import pytest

from mesh2splat_amd import _lib, synth
from mesh2splat_amd.converter import Converter
from mesh2splat_amd.scene import Mesh, Scene
from parity import assert_records_match

pytestmark = pytest.mark.gpu


def both(oracle, scene, R, cap=0, pipeline="auto"):
    c = Converter(0)
    c.set_pipeline(pipeline)
    c.upload_scene(scene)
    c.set_max_gaussians(cap)
    total = c.convert(R)
    rec = c.download()
    c.close()
    ototal, orec, _ = oracle.convert(scene, R, cap=cap)
    assert total == ototal, (total, ototal)
    assert_records_match(rec, orec, f"R={R}")
    return total


@pytest.mark.parametrize("R", [1, 2, 3, 4095, 4096])
def test_extreme_resolutions(hiplib, oracle, R):
    """R = 1 .. 4096 (the UI's maximum, ImGuiUi.hpp:116-118): quad covers exactly R*R pixels."""
    scene = synth.unit_quad() if R > 100 else synth.cube_sphere(3)
    n = both(oracle, scene, R)
    if R > 100:
        assert n == R * R


def test_nonfinite_and_out_of_range_geometry(hiplib, oracle):
    """NaN / inf positions and positions far outside a caller-supplied bbox: dropped identically by both sides
    (non-finite or |window coordinate| >= 16384 px), the rest of the mesh is unaffected."""
    base = synth.cube_sphere(3).meshes[0].vertices.copy()
    v = base.copy()
    v[0, 0] = np.nan
    v[4, 1] = np.inf
    v[6:9, 0:3] *= 1e6          # far outside the bbox -> guard band
    v[9:12, 0:3] += 3.0         # outside the bbox but inside the guard band: clipped to the viewport
    scene = Scene([Mesh("m", v, bbox_min=np.float32([-1, -1, -1]), bbox_max=np.float32([1, 1, 1]))])
    for pipe in ("auto", "multipass", "team"):
        assert both(oracle, scene, 64, pipeline=pipe) > 0


def test_degenerate_uvs_and_tiny_textures(hiplib, oracle):
    """1x1 and 2x1 textures (a single mip level / NPOT chain), constant UVs (zero derivatives -> lambda = -inf)."""
    tex = {"baseColorTexture": np.full((1, 1, 4), 200, np.uint8), "normalTexture": np.array([[[128, 128, 255, 255], [255, 128, 128, 255]]], np.uint8),
           "metallicRoughnessTexture": np.array([[[0, 64, 192, 255]], [[0, 200, 10, 255]], [[0, 1, 2, 255]]], np.uint8)}
    scene = synth.unit_quad(tex)
    scene.meshes[0].vertices[:3, 10:12] = 0.25     # first triangle: constant uv
    both(oracle, scene, 48)


def test_many_small_meshes_straddling_waves(hiplib, oracle):
    """Mesh boundaries inside a 64-triangle wave tile (per-lane mesh lookup path) and empty meshes in between."""
    rng = np.random.default_rng(3)
    meshes = []
    for k in range(40):
        nt = int(rng.integers(0, 30))
        s = synth.random_soup(max(nt, 1), seed=100 + k, textures=synth.procedural_textures(8, k) if k % 3 == 0 else None)
        m = s.meshes[0]
        m.name = f"m_{k}"
        if nt == 0:
            m = Mesh(f"m_{k}", np.zeros((0, 12), np.float32))
        m.bbox_min = m.bbox_max = None
        m.base_color = (0.2 + 0.02 * k, 0.5, 1.0 - 0.02 * k, 1.0)
        meshes.append(m)
    scene = Scene(meshes)
    for pipe in ("auto", "multipass", "team"):
        both(oracle, scene, 200, pipeline=pipe)


def test_api_errors(hiplib):
    L = _lib.load()
    c = Converter(0)
    with pytest.raises(_lib.M2SError, match="M2S_ERR_STATE"):
        c.convert(64)                                  # convert before upload
    c.upload_scene(synth.unit_quad())
    with pytest.raises(_lib.M2SError, match="M2S_ERR_INVALID"):
        c.convert(0)
    with pytest.raises(_lib.M2SError, match="M2S_ERR_INVALID"):
        c.convert(5000)
    with pytest.raises(_lib.M2SError, match="M2S_ERR_INVALID"):
        c.set_max_gaussians(-5)
    bad = (_lib.MeshC * 1)()
    bad[0].n_vertices = 4
    bad[0].stride_floats = 12
    assert L.m2s_upload_scene(c._h, bad, 1) == 1       # not a multiple of 3
    bad[0].n_vertices = 3
    bad[0].stride_floats = 11
    assert L.m2s_upload_scene(c._h, bad, 1) == 1       # stride too small
    c.upload_scene(synth.unit_quad())
    assert c.convert(8) == 64
    small = np.zeros((10, 24), np.float32)
    assert L.m2s_download(c._h, small.ctypes.data, 10) == 5    # M2S_ERR_CAPACITY
    assert b"fewer records" in L.m2s_last_error(c._h)
    with pytest.raises(_lib.M2SError, match="M2S_ERR_IO"):
        c.export_ply("/nonexistent_dir/x.ply")
    c.close()


def test_convert_into_user_buffer_and_capacity(hiplib, oracle):
    """m2s_convert_into: records land in caller-owned device memory; capacity bounds what is stored, the counter
    is unaffected."""
    torch = pytest.importorskip("torch")
    scene = synth.cube_sphere(10, tex_size=32)
    total, orec, _ = oracle.convert(scene, 100, cap=0)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    buf = torch.zeros((total, 24), dtype=torch.float32, device="cuda")
    n = c.convert_into(100, buf.data_ptr(), total, torch.cuda.current_stream().cuda_stream)
    assert n == total and c.num_stored == total
    assert_records_match(buf.cpu().numpy(), orec, "convert_into")
    half = torch.full((total // 2 + 7, 24), -1.0, dtype=torch.float32, device="cuda")
    n = c.convert_into(100, half.data_ptr(), total // 2, torch.cuda.current_stream().cuda_stream)
    assert n == total and c.num_stored == total // 2
    h = half.cpu().numpy()
    assert_records_match(h[: total // 2], orec[: total // 2], "capacity")
    assert np.all(h[total // 2:] == -1.0)              # nothing written past the capacity
    c.close()


def test_team_kernel_falls_back_when_a_workgroup_overflows_its_stream(hiplib, oracle):
    """k_fused2 keeps a workgroup's fragments in a 4096-entry LDS stream.  A scene whose AVERAGE is small (AUTO picks the
    single-pass kernel) but that has a cluster of 256 consecutive triangles with ~80 fragments each overflows it: the
    host must repeat the conversion with the multi-pass pipeline, return the right answer and remember the decision."""
    import numpy as np
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    from mesh2splat_amd.scene import Mesh, Scene
    from parity import assert_records_match
    base = synth.cube_sphere(130, tex_size=64)              # 202 800 small triangles: fills the GPU with 64-triangle batches
    v = base.meshes[0].vertices.copy().reshape(-1, 3, base.meshes[0].vertices.shape[1])
    # blow up 256 consecutive triangles (4 batches = one workgroup) around their centroids
    sel = slice(64 * 400, 64 * 404)
    cen = v[sel, :, 0:3].mean(axis=1, keepdims=True)
    v[sel, :, 0:3] = cen + (v[sel, :, 0:3] - cen) * 4.0
    m = base.meshes[0]
    scene = Scene([Mesh(name=m.name, vertices=v.reshape(-1, v.shape[2]), base_color=m.base_color, textures=m.textures)])
    R = 512
    ototal, orec, _ = oracle.convert(scene, R, cap=0)
    assert ototal < 11 * scene.n_triangles                   # AUTO -> single pass
    cnt = oracle.count_per_triangle(scene, R)
    assert cnt[sel].sum() > 4096                              # ... but this workgroup does not fit
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    # AUTO takes the lean form for a scene of this size (round 5): ITS workgroups only keep triangles of at most 8 x 8 pixels, the
    # cluster is deferred to k_emit_big and nothing overflows
    assert c.convert(R) == ototal
    assert c.last_pipeline == "lean"
    assert_records_match(c.download(), orec, "lean form, cluster deferred")
    c.set_pipeline("team")                                    # k_fused2 expands the cluster in the workgroup: that overflows
    for _ in range(2):
        assert c.convert(R) == ototal
        assert c.last_pipeline == "multipass"                 # the team form gave up, the multi-pass pipeline answered (and is remembered)
    assert_records_match(c.download(), orec, "fallback to the multi-pass pipeline")
    c.set_pipeline("auto")
    # an ordinary scene runs the team form, several meshes included
    grid = synth.colocated_spheres(3, 150, 64)
    c.upload_scene(grid)
    c.set_max_gaussians(0)
    total = c.convert(512)
    assert c.last_pipeline == "lean"                      # (AUTO: the team kernel in its lean form, k_fused3 — combo textures, > 172 k triangles)
    ototal, orec, _ = oracle.convert(grid, 512, cap=0)
    assert total == ototal
    assert_records_match(c.download(), orec, "team form, three meshes")
    c.close()


def test_chain_tag_wrap_between_the_two_single_pass_forms(hiplib, oracle):
    """Look-back chain words are tagged with the low 16 bits of a launch counter instead of being cleared.  The team kernel
    (one word per 64 triangles) uses more chain words than k_count_scan (one per 256), so words it wrote keep their tag while only the multi-pass pipeline runs; when the
    counter comes round to the same tag 65 536 launches later they must not read as fresh.  Walk the counter across
    the wrap with the test hook and alternate the forms around it."""
    scene = synth.cube_sphere(60, tex_size=32)
    R = 256
    ototal, orec, _ = oracle.convert(scene, R, cap=0)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    L = hiplib

    def run(form):
        c.set_pipeline(form)
        assert c.convert(R) == ototal
        assert c.last_pipeline == form
        assert_records_match(c.download(), orec, "tag wrap, %s" % form)

    for first, second in (("team", "multipass"), ("multipass", "team")):
        assert L.m2s_debug_set_launch_counter(c._h, 0x10000 - 2) == 0
        run(first)                    # tag 0xFFFF
        run(second)                   # tag 0x0000: chains cleared
        run(second)                   # 0x0001
        run(first)                    # 0x0002 on every word this form uses
        assert L.m2s_debug_set_launch_counter(c._h, 0x20000 - 2) == 0
        for _ in range(3):
            run(second)               # 0xFFFF, 0x0000 (cleared), 0x0001: overwrites only the words the other form uses
        run(first)                    # 0x0002 again: without the clear it would meet its own words of one period ago
    c.close()
