"""CPU: scene I/O — the re-hosted .glb loader (SceneManager::loadModel minus GL), its PNG decoder and the
.ply reader (parsers::loadPlyFile).  No GPU needed: these are host functions of the C-ABI library."""
import io
import subprocess

import numpy as np
import pytest

from mesh2splat_amd import gltf_io, synth
from mesh2splat_amd._lib import M2SError
from mesh2splat_amd.converter import write_ply
from mesh2splat_amd.scene import Mesh, Scene

f32 = np.float32


def test_roundtrip_identity(tmp_path, hiplib):
    """write_glb -> loader reproduces vertices, names, factors, textures and the cumulative bboxes exactly."""
    scene = synth.sphere_grid(2, n=3, tex_size=8)
    p = str(tmp_path / "s.glb")
    gltf_io.write_glb(scene, p)
    got = gltf_io.load_glb(p)
    assert got.n_meshes == scene.n_meshes and not got.warnings
    for a, b in zip(got.meshes, scene.meshes):
        assert a.name == b.name and a.stride == 17
        assert np.array_equal(a.vertices[:, 0:3], b.vertices[:, 0:3])          # identity transform is exact
        assert np.allclose(a.vertices[:, 3:6], b.vertices[:, 3:6], atol=1e-6)  # normalize(normalMatrix * n)
        assert np.allclose(a.vertices[:, 6:9], b.vertices[:, 6:9], atol=1e-6)
        assert np.array_equal(a.vertices[:, 9:12], b.vertices[:, 9:12])        # tangent.w, uv untouched
        assert np.all(a.vertices[:, 12:17] == 0)                               # normalizedUv, scale: always 0 (dead)
        assert np.allclose(a.base_color, b.base_color)
        assert np.array_equal(a.bbox_min, b.bbox_min) and np.array_equal(a.bbox_max, b.bbox_max)   # cumulative (Q1)
        for k in b.textures:
            assert np.array_equal(a.textures[k], b.textures[k])


@pytest.mark.parametrize("indexed,index_type", [(True, "u8"), (True, "u16"), (True, "u32"), (False, "auto")])
def test_index_types_and_nonindexed(tmp_path, hiplib, indexed, index_type):
    scene = synth.cube_sphere(2)     # 48 triangles, < 255 unique vertices
    p = str(tmp_path / "i.glb")
    gltf_io.write_glb(scene, p, indexed=indexed, index_type=index_type)
    got = gltf_io.load_glb(p)
    assert np.array_equal(got.meshes[0].vertices[:, 0:3], scene.meshes[0].vertices[:, 0:3])


def test_node_transforms(tmp_path, hiplib):
    """TRS and matrix nodes, nested under a parent: world = parent * T * R * S (SceneManager.cpp:224-257);
    positions by the world matrix, normals by transpose(inverse(mat3)), tangents by mat3 (:394-420)."""
    scene = synth.sphere_grid(2, n=2)
    q = np.array([0.1, 0.5, -0.2, 0.8]); q /= np.linalg.norm(q)
    trs = [dict(translation=(1, 2, 3), rotation=q, scale=(2, 0.5, 1.5)), dict(scale=(1, -1, 1)), {},
           dict(matrix=[1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 5, 6, 7, 1])] + [{}] * 4
    p = str(tmp_path / "t.glb")
    gltf_io.write_glb(scene, p, node_trs=trs, nested=True)
    got = gltf_io.load_glb(p)

    def mat_of(t):
        if "matrix" in t:
            return np.array(t["matrix"], np.float64).reshape(4, 4).T
        M = np.eye(4)
        if "scale" in t:
            M = np.diag(list(t["scale"]) + [1.0]) @ M
        if "rotation" in t:
            x, y, z, w = t["rotation"]
            Rm = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                           [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                           [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
            R4 = np.eye(4); R4[:3, :3] = Rm
            M = R4 @ M
        if "translation" in t:
            T4 = np.eye(4); T4[:3, 3] = t["translation"]
            M = T4 @ M
        return M

    mn, mx = np.full(3, np.inf), np.full(3, -np.inf)
    for i, (a, b) in enumerate(zip(got.meshes, scene.meshes)):
        M = mat_of(trs[i])
        pos = (M[:3, :3] @ b.vertices[:, 0:3].astype(np.float64).T).T + M[:3, 3]
        assert np.allclose(a.vertices[:, 0:3], pos, rtol=1e-6, atol=1e-6)
        nm = np.linalg.inv(M[:3, :3]).T
        nrm = (nm @ b.vertices[:, 3:6].astype(np.float64).T).T
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        assert np.allclose(a.vertices[:, 3:6], nrm, atol=2e-6)
        tan = (M[:3, :3] @ b.vertices[:, 6:9].astype(np.float64).T).T
        tan /= np.linalg.norm(tan, axis=1, keepdims=True)
        assert np.allclose(a.vertices[:, 6:9], tan, atol=2e-6)
        assert np.array_equal(a.vertices[:, 9], b.vertices[:, 9])
        mn = np.minimum(mn, a.vertices[:, 0:3].min(0)); mx = np.maximum(mx, a.vertices[:, 0:3].max(0))
        assert np.array_equal(a.bbox_min, mn.astype(f32)) and np.array_equal(a.bbox_max, mx.astype(f32))


def test_fallback_normals_and_tangents(tmp_path, hiplib):
    """No NORMAL -> flat face normal (:406-413); no TANGENT -> per-face tangent from the UVs (:421-451)."""
    scene = synth.cube_sphere(2)
    p = str(tmp_path / "f.glb")
    gltf_io.write_glb(scene, p, with_normals=False, with_tangents=False, indexed=False)
    got = gltf_io.load_glb(p).meshes[0].vertices.reshape(-1, 3, 17)
    src = scene.meshes[0].vertices.reshape(-1, 3, 12)
    for t in range(src.shape[0]):
        P, UV = src[t, :, 0:3].astype(f32), src[t, :, 10:12].astype(f32)
        dp1, dp2 = P[1] - P[0], P[2] - P[0]
        fn = np.cross(dp1, dp2); fn = fn / np.linalg.norm(fn)
        assert np.allclose(got[t, :, 3:6], fn, atol=1e-6)
        duv1, duv2 = UV[1] - UV[0], UV[2] - UV[0]
        det = duv1[0] * duv2[1] - duv1[1] * duv2[0]
        if abs(det) < 1e-8:
            det = 1.0
        tg = (dp1 * duv2[1] - dp2 * duv1[1]) / det
        bt = (dp2 * duv1[0] - dp1 * duv2[0]) / det
        tg, bt = tg / np.linalg.norm(tg), bt / np.linalg.norm(bt)
        hand = -1.0 if np.dot(np.cross(fn, tg), bt) < 0 else 1.0
        assert np.allclose(got[t, :, 6:9], tg, atol=1e-5) and np.all(got[t, :, 9] == hand)
    # without TEXCOORD_0 the uv stays (0,0) (value-initialised Face, :384) and the tangent degenerates like the reference's
    gltf_io.write_glb(scene, p, with_uvs=False)
    assert np.all(gltf_io.load_glb(p).meshes[0].vertices[:, 10:12] == 0)


def test_png_decoder_against_pil(tmp_path, hiplib):
    """Colour types 0/2/3/4/6, bit depths 1-16, Adam7 interlace: decoded RGBA8 == PIL's RGBA conversion."""
    PIL = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    cases = []
    rgba = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    cases.append(("rgba", PIL.fromarray(rgba, "RGBA"), {}))
    cases.append(("rgb", PIL.fromarray(rgba[..., :3], "RGB"), {}))
    cases.append(("grey", PIL.fromarray(rgba[..., 0], "L"), {}))
    cases.append(("grey_alpha", PIL.fromarray(rgba[..., :2], "LA"), {}))
    cases.append(("palette", PIL.fromarray(rgba[..., :3], "RGB").quantize(16), {}))
    cases.append(("bilevel", PIL.fromarray((rgba[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1"), {}))
    cases.append(("interlaced", PIL.fromarray(rgba, "RGBA"), {"interlace": True}))
    cases.append(("grey16", PIL.fromarray((rgba[..., 0].astype(np.uint16) << 8) | 7, "I;16"), {}))
    quad = synth.unit_quad()
    for name, im, kw in cases:
        buf = io.BytesIO()
        if kw.get("interlace"):
            pytest.importorskip("PIL.PngImagePlugin")
            try:
                im.save(buf, "PNG", interlace=1)
            except Exception:
                continue
            if b"IHDR" not in buf.getvalue() or buf.getvalue()[28] != 1:   # PIL cannot write Adam7: skip the case
                continue
        else:
            im.save(buf, "PNG")
        ref = np.asarray(im.convert("RGBA")) if name != "grey16" else None
        scene = Scene([Mesh("q_0", quad.meshes[0].vertices, textures={"baseColorTexture": np.zeros((1, 1, 4), np.uint8)})])
        p = str(tmp_path / f"{name}.glb")
        gltf_io.write_glb(scene, p, png_override={"baseColorTexture": buf.getvalue()})
        tex = gltf_io.load_glb(p).meshes[0].textures["baseColorTexture"]
        if name == "grey16":
            assert np.all(tex[..., 0] == rgba[..., 0]) and np.all(tex[..., 3] == 255)   # high byte, like stb_image
        else:
            assert np.array_equal(tex, ref), name


def test_own_png_encoder_roundtrip(tmp_path, hiplib):
    tex = synth.procedural_textures(32)
    scene = synth.unit_quad(tex)
    p = str(tmp_path / "q.glb")
    gltf_io.write_glb(scene, p)
    got = gltf_io.load_glb(p)
    for k, v in tex.items():
        assert np.array_equal(got.meshes[0].textures[k], v)


def test_loader_errors(tmp_path, hiplib):
    with pytest.raises(M2SError, match="Failed to parse GLTF file"):
        gltf_io.load_glb(str(tmp_path / "missing.glb"))
    bad = tmp_path / "bad.glb"
    bad.write_bytes(b"not a glb at all, definitely")
    with pytest.raises(M2SError, match="not a binary glTF"):
        gltf_io.load_glb(str(bad))
    # broken image payloads -> explicit error naming the image (no silent texture loss)
    scene = synth.unit_quad({"baseColorTexture": np.zeros((2, 2, 4), np.uint8)})
    p = str(tmp_path / "jpg.glb")
    gltf_io.write_glb(scene, p, png_override={"baseColorTexture": b"\xff\xd8\xff\xe0" + b"\0" * 64})
    with pytest.raises(M2SError, match="image 0 .image/jpeg."):
        gltf_io.load_glb(p)
    gltf_io.write_glb(scene, p, png_override={"baseColorTexture": b"GIF89a" + b"\0" * 64})
    with pytest.raises(M2SError, match="not a PNG"):
        gltf_io.load_glb(p)


@pytest.mark.parametrize("fmt", [0, 1])
def test_ply_reader_roundtrip(tmp_path, hiplib, oracle, fmt):
    """parsers::loadPlyFile semantics: exp(scale), sigmoid(opacity), SH -> RGB, normalised quaternion."""
    scene = synth.cube_sphere(4, tex_size=16)
    _, rec, _ = oracle.convert(scene, 32, cap=0)
    rec[:, 7] = np.linspace(0.05, 0.95, len(rec), dtype=f32)          # finite opacities
    sm = f32(0.65) / f32(32)
    p = str(tmp_path / "r.ply")
    write_ply(p, rec, fmt, sm)
    got, has_pbr = gltf_io.read_ply(p)
    assert got.shape == rec.shape and has_pbr == (fmt == 1)
    assert np.array_equal(got[:, 0:3], rec[:, 0:3]) and np.all(got[:, 3] == 1)
    assert np.allclose(got[:, 4:7], rec[:, 4:7], atol=1e-6)
    assert np.allclose(got[:, 7], rec[:, 7], atol=1e-6)
    assert np.allclose(got[:, 8:11], rec[:, 8:11] * sm, rtol=1e-5) and np.all(got[:, 11] == 1)
    q = rec[:, 16:20] / np.linalg.norm(rec[:, 16:20], axis=1, keepdims=True)
    assert np.allclose(got[:, 16:20], q, atol=1e-6)
    if fmt == 1:
        assert np.array_equal(got[:, 12:15], rec[:, 12:15]) and np.array_equal(got[:, 20:22], rec[:, 20:22])
    else:
        assert np.all(got[:, 12:16] == 0) and np.all(got[:, 20:24] == 0)
    with pytest.raises(M2SError):
        write_ply(p, rec, 2, sm)
        gltf_io.read_ply(p)          # compressed format has no f_dc_*: the reference's reader cannot load it either


def test_cli_usage(hiplib):
    import os
    from mesh2splat_amd import _lib
    exe = os.path.join(os.path.dirname(_lib.LIB_PATH), "mesh2splat")
    assert os.path.exists(exe), "CLI not built"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage: mesh2splat" in r.stderr
    r = subprocess.run([exe, "/nonexistent.glb", "/tmp/x.ply"], capture_output=True, text=True)
    assert r.returncode == 1 and "Failed to parse GLTF file" in r.stderr


def test_loader_transforms_match_glm():
    """oracle/_ref/glm_xform_check: the loader's glm-free TRS / normal-matrix code == glm (vendored by the
    reference) bit for bit, on the expressions of SceneManager.cpp:224-257,285,394-420."""
    import os
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "glm_xform_check")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref not built (reference tree absent on this machine)")
    r = subprocess.run([exe, "20000"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout


def _patch_glb_json(src: str, dst: str, **extra):
    """Rewrite a .glb with extra top-level JSON members (test helper)."""
    import json
    import struct
    blob = open(src, "rb").read()
    jlen = struct.unpack_from("<I", blob, 12)[0]
    doc = json.loads(blob[20:20 + jlen])
    doc.update(extra)
    j = json.dumps(doc).encode()
    j += b" " * (-len(j) % 4)
    rest = blob[20 + jlen:]
    out = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(j) + len(rest)) + struct.pack("<II", len(j), 0x4E4F534A) + j + rest
    open(dst, "wb").write(out)


def test_gltf_extensions(tmp_path, hiplib):
    """Geometry-encoding extensions the reference's loader would misread are refused with a clear message; appearance
    extensions are ignored (as the reference ignores them) and reported."""
    from mesh2splat_amd import _lib
    scene = synth.cube_sphere(2, tex_size=8)
    base = str(tmp_path / "base.glb")
    gltf_io.write_glb(scene, base)
    used = str(tmp_path / "used.glb")
    _patch_glb_json(base, used, extensionsUsed=["KHR_texture_transform", "KHR_materials_emissive_strength"])
    loaded = gltf_io.load_glb(used)
    ref = gltf_io.load_glb(base)
    assert np.array_equal(loaded.meshes[0].vertices, ref.meshes[0].vertices)
    assert any("KHR_texture_transform" in w for w in loaded.warnings) and not any("KHR_" in w for w in ref.warnings)
    for ext in ("KHR_draco_mesh_compression", "EXT_meshopt_compression", "KHR_mesh_quantization"):
        bad = str(tmp_path / "bad.glb")
        _patch_glb_json(base, bad, extensionsUsed=[ext], extensionsRequired=[ext])
        with pytest.raises(_lib.M2SError, match=ext):
            gltf_io.load_glb(bad)


def test_loader_threads_do_not_change_the_result(tmp_path, hiplib, monkeypatch):
    """Large meshes are de-indexed, and embedded images decoded, by several host threads: same bytes as the serial run."""
    scene = synth.cube_sphere(80, tex_size=64)                      # 76 800 triangles: above the per-thread grain
    p = str(tmp_path / "big.glb")
    gltf_io.write_glb(scene, p, with_tangents=False)                # exercises the fallback tangents too
    monkeypatch.setenv("M2S_HOST_THREADS", "1")
    serial = gltf_io.load_glb(p)
    for threads in ("2", "7"):
        monkeypatch.setenv("M2S_HOST_THREADS", threads)
        par = gltf_io.load_glb(p)
        assert np.array_equal(par.meshes[0].vertices.view(np.uint32), serial.meshes[0].vertices.view(np.uint32))
        for k, t in serial.meshes[0].textures.items():
            assert np.array_equal(par.meshes[0].textures[k], t)
        assert np.array_equal(par.meshes[0].bbox_min, serial.meshes[0].bbox_min)


def test_heterogeneous_scene_through_the_loader(tmp_path, hiplib, oracle):
    """synth.sponza_like as a FILE: 64 meshes, materials with three maps / an albedo map only / none, 2-triangle planes next to
    dense cloth — written as .glb, read back by the C++ loader: same geometry, same maps (and no map where there was none), same
    cumulative bounding boxes, and therefore the same fragment count per triangle as the in-memory scene (oracle)."""
    scene = synth.sponza_like(tex_scale=1.0 / 32.0)
    p = str(tmp_path / "hetero.glb")
    gltf_io.write_glb(scene, p, indexed=False)
    got = gltf_io.load_glb(p)
    assert got.n_meshes == scene.n_meshes == 64 and not got.warnings
    kinds = set()
    for a, b in zip(got.meshes, scene.meshes):
        assert a.n_triangles == b.n_triangles
        assert np.array_equal(a.vertices[:, 0:3], b.vertices[:, 0:3])
        assert np.array_equal(a.vertices[:, 9:12], b.vertices[:, 9:12])
        assert np.array_equal(a.bbox_min, b.bbox_min) and np.array_equal(a.bbox_max, b.bbox_max)
        assert set(a.textures) == set(b.textures)
        kinds.add(len(b.textures))
        for k in b.textures:
            assert np.array_equal(a.textures[k], b.textures[k])
    assert kinds == {0, 1, 3}
    R = 256
    assert np.array_equal(oracle.count_per_triangle(got, R), oracle.count_per_triangle(scene, R))
