"""CPU: this repository's loader and PLY writers/readers against the REFERENCE'S OWN host code.

Two layers:
  * golden  — outputs of the reference's SceneManager::loadModel / parsers::savePlyVector / parsers::loadPlyFile
              (compiled from /root/reference by oracle/Makefile into oracle/_ref/ref_host_check, GL stubbed),
              committed under tests/golden/ref_host/ by tests/golden/make_ref_golden.py.  Always runs.
  * live    — the same comparison on more inputs by running that binary; skipped where it was not built.
Bar: bit-exact (vertex floats compared as uint32, signed zeros included; .ply files byte for byte)."""
import os

import numpy as np
import pytest

import refhost
from mesh2splat_amd import gltf_io, synth
from mesh2splat_amd.converter import write_ply

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")
SCENES = ["trs_nested", "flat_nonindexed", "no_uv_u8", "mixed_materials_u32"]
live = pytest.mark.skipif(not refhost.available(), reason="oracle/_ref/ref_host_check not built (needs /root/reference)")


def assert_scene_equal(ref, mine):
    assert len(ref) == mine.n_meshes
    for r, m in zip(ref, mine.meshes):
        assert r["name"] == m.name
        assert r["vertices"].shape == m.vertices.shape
        assert np.array_equal(r["vertices"].view(np.uint32), np.ascontiguousarray(m.vertices).view(np.uint32)), m.name
        assert np.array_equal(r["bbox_min"], m.bbox_min) and np.array_equal(r["bbox_max"], m.bbox_max)   # cumulative bbox (Q1)
        assert np.array_equal(r["base_color"], np.asarray(m.base_color, np.float32))
        assert set(r["textures"]) == set(m.textures)
        for k, t in r["textures"].items():
            assert t.shape[2] == 4 and np.array_equal(t, m.textures[k])


@pytest.mark.parametrize("name", SCENES)
def test_loader_matches_reference_golden(hiplib, name):
    with open(os.path.join(GOLD, name + ".scene.bin"), "rb") as f:
        ref = refhost.parse_scene_dump(f.read())
    assert_scene_equal(ref, gltf_io.load_glb(os.path.join(GOLD, name + ".glb")))


AUTHORED = ["authored_grid_trs", "authored_nonindexed"]


@pytest.mark.parametrize("name", AUTHORED)
def test_loader_on_assets_authored_by_the_references_tiny_gltf(hiplib, name):
    """the loader against assets it did not generate.  These .glb files were written by the REFERENCE's own
    tiny_gltf (WriteGltfSceneToFile, binary) with images PNG-encoded by its stb_image_write — another JSON layout, another buffer
    packing, another deflate stream than mesh2splat_amd.gltf_io produces — and loaded by the reference's
    SceneManager::loadModel; this repository's loader must give the same bytes."""
    with open(os.path.join(GOLD, name + ".scene.bin"), "rb") as f:
        ref = refhost.parse_scene_dump(f.read())
    assert_scene_equal(ref, gltf_io.load_glb(os.path.join(GOLD, name + ".glb")))


def assert_loaded_equals_source(loaded, scene):
    assert loaded.n_meshes == scene.n_meshes
    for m, src in zip(loaded.meshes, scene.meshes):
        got = np.ascontiguousarray(m.vertices).reshape(-1, 17)
        want = np.ascontiguousarray(src.vertices, np.float32).reshape(-1, src.stride)
        assert got.shape[0] == want.shape[0]
        # positions, tangent handedness and texture coordinates bit for bit; normals and tangents are re-normalised by the loader
        # (as by the reference's: SceneManager.cpp:387-454) — one ulp
        for sl in (slice(0, 3), slice(9, 12)):
            assert np.array_equal(got[:, sl].view(np.uint32), want[:, sl].view(np.uint32)), m.name
        assert np.allclose(got[:, 3:9], want[:, 3:9], rtol=0, atol=2e-7), m.name
        for k, t in src.textures.items():
            assert np.array_equal(m.textures[k], t)


def test_interleaved_views_follow_the_file_not_the_reference(hiplib):
    """One interleaved buffer view with byteStride 48 and 16-bit indices, written by the reference's tiny_gltf.  The REFERENCE'S
    loader mis-reads it — its getBufferData ignores accessor / bufferView strides (SceneManager.cpp:50-61) and
    returns position bytes where normals are expected from the second vertex on; the committed dump of what it produced shows
    that.  This repository's loader follows the glTF specification: it must return exactly the arrays the file was written from."""
    mine = gltf_io.load_glb(os.path.join(GOLD, "authored_interleaved_u16.glb"))
    assert_loaded_equals_source(mine, synth.cube_sphere(3, tex_size=32))
    with open(os.path.join(GOLD, "authored_interleaved_u16.scene.bin"), "rb") as f:
        ref = refhost.parse_scene_dump(f.read())
    v = np.ascontiguousarray(mine.meshes[0].vertices).reshape(-1, 17)
    assert np.array_equal(ref[0]["vertices"][0], v[0]) and not np.array_equal(ref[0]["vertices"][1], v[1])   # the reference: right at vertex 0 only


@live
@pytest.mark.parametrize("flags", [0, 1, 4, 8, 1 | 8, 2, 1 | 2])
def test_loader_on_tiny_gltf_authored_assets_live(tmp_path, hiplib, flags):
    scene = synth.sphere_grid(2, n=3, tex_size=24)
    scene.meshes[1].textures.pop("normalTexture", None)
    glb = str(tmp_path / "authored.glb")
    if flags & 2:        # interleaved views: against the source (see above), identity node transforms
        refhost.write_glb_by_tinygltf(scene, glb, str(tmp_path), flags=flags)
        assert_loaded_equals_source(gltf_io.load_glb(glb), scene)
        return
    trs = [((k, 0.5 * k, -k), (0, 0, 0, 1), (1 + 0.1 * k, 1, 1)) for k in range(scene.n_meshes)]
    refhost.write_glb_by_tinygltf(scene, glb, str(tmp_path), flags=flags, trs=trs)
    assert_scene_equal(refhost.load_scene(glb, str(tmp_path)), gltf_io.load_glb(glb))


def golden_records():
    return np.fromfile(os.path.join(GOLD, "records.bin"), np.float32).reshape(-1, 24)


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_ply_writers_match_reference_golden(tmp_path, hiplib, oracle, fmt):
    rec = golden_records()
    sm = np.float32(0.65) / np.float32(40)
    want = open(os.path.join(GOLD, f"ref_fmt{fmt}.ply"), "rb").read()
    a, b = str(tmp_path / "prod.ply"), str(tmp_path / "orc.ply")
    write_ply(a, rec, fmt, sm)
    oracle.write_ply(b, rec, fmt, sm)
    assert open(a, "rb").read() == want      # product (m2s_write_ply)
    assert open(b, "rb").read() == want      # oracle restatement


@pytest.mark.parametrize("fmt", [0, 1])
def test_ply_reader_matches_reference_golden(hiplib, fmt):
    with open(os.path.join(GOLD, f"ref_read_fmt{fmt}.bin"), "rb") as f:
        want, want_pbr = refhost.parse_ply_dump(f.read())
    got, pbr = gltf_io.read_ply(os.path.join(GOLD, f"ref_fmt{fmt}.ply"))
    assert pbr == want_pbr == (fmt == 1)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def _ply_variant(src_path, dst_path, kind):
    """Re-encode a binary_little_endian .ply (float properties only) as ascii or binary_big_endian."""
    blob = open(src_path, "rb").read()
    h = blob.index(b"end_header\n") + 11
    header = blob[:h].decode()
    names = [ln.split()[2] for ln in header.splitlines() if ln.startswith("property")]
    rows = np.frombuffer(blob[h:], "<f4").reshape(-1, len(names))
    if kind == "big":
        out = header.replace("binary_little_endian", "binary_big_endian").encode() + rows.astype(">f4").tobytes()
    else:
        lines = [" ".join("%.9g" % v for v in r) for r in rows]
        lines[3:3] = [""]                                   # happly skips empty lines in front of a vertex line
        out = header.replace("binary_little_endian", "ascii").encode() + ("\n".join(lines) + "\n").encode()
    with open(dst_path, "wb") as f:
        f.write(out)


@pytest.mark.parametrize("fmt,kind", [(0, "ascii"), (1, "ascii"), (0, "big"), (1, "big")])
def test_ply_reader_accepts_what_happly_accepts(tmp_path, hiplib, fmt, kind):
    """VERDICT r3 (missing 5): ascii and big-endian .ply files — happly, the reference's reader (parsers.cpp:516-629), takes all three
    encodings.  Always: the records equal those of the little-endian original (a +inf opacity written as "inf" in ascii reads
    back as 0 in BOTH readers: operator>> does not parse it — so that column is compared where it is finite).  Where the
    reference's reader was built: bit-identical to it on the same file."""
    src = os.path.join(GOLD, f"ref_fmt{fmt}.ply")
    dst = str(tmp_path / f"{kind}.ply")
    _ply_variant(src, dst, kind)
    got, pbr = gltf_io.read_ply(dst)
    base, base_pbr = gltf_io.read_ply(src)
    assert pbr == base_pbr and got.shape == base.shape
    finite_alpha = base[:, 7] < 1.0 if kind == "ascii" else np.ones(len(base), bool)
    assert np.array_equal(got[finite_alpha].view(np.uint32), base[finite_alpha].view(np.uint32))
    if refhost.available():
        want, want_pbr = refhost.read_ply(dst, str(tmp_path))
        assert pbr == want_pbr
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


# ---- live -----------------------------------------------------------------------------------------------
def live_scene_cases():
    rng = np.random.default_rng(5)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    trs = [dict(rotation=q, scale=(-1.5, 2, -0.25)), dict(translation=(0.5, -0.0, -3)), dict(scale=(1, 1, -1), translation=(0, 0, 1)), {}]
    yield "trs", synth.sphere_grid(2, n=3, tex_size=16), dict(node_trs=trs, nested=True)
    yield "u16", synth.cube_sphere(8, tex_size=16), dict(index_type="u16")
    yield "soup", synth.random_soup(200, seed=3, textures=synth.procedural_textures(8, 1)), dict(indexed=False)
    yield "soup_flat", synth.random_soup(80, seed=4), dict(with_normals=False, with_tangents=False)
    yield "colocated", synth.colocated_spheres(3, n=3, tex_size=8), dict()


@live
@pytest.mark.parametrize("case", list(live_scene_cases()), ids=lambda c: c[0])
def test_loader_matches_reference_live(tmp_path, hiplib, case):
    name, scene, kw = case
    glb = str(tmp_path / (name + ".glb"))
    gltf_io.write_glb(scene, glb, **kw)
    assert_scene_equal(refhost.load_scene(glb, str(tmp_path)), gltf_io.load_glb(glb))


@live
@pytest.mark.parametrize("fmt", [0, 1, 2, 7])
def test_ply_matches_reference_live(tmp_path, hiplib, oracle, fmt):
    rng = np.random.default_rng(fmt)
    rec = rng.uniform(-1.0, 1.0, (5000, 24)).astype(np.float32)
    rec[:, 4:8] = rng.uniform(0.0, 1.0, (5000, 4))            # colour / alpha
    rec[:, 8:10] = rng.uniform(1e-4, 0.1, (5000, 2))          # scale (log taken)
    rec[:, 10] = 1e-7
    rec[::7, 7] = 1.0
    rec[:, 20:22] = rng.uniform(0.0, 1.0, (5000, 2))
    sm = np.float32(0.65) / np.float32(333)
    a, b, c = (str(tmp_path / n) for n in ("ref.ply", "prod.ply", "orc.ply"))
    refhost.write_ply(rec, a, fmt, sm, str(tmp_path))
    write_ply(b, rec, fmt, sm)
    oracle.write_ply(c, rec, fmt, sm)
    want = open(a, "rb").read()
    assert open(b, "rb").read() == want and open(c, "rb").read() == want
    if fmt in (0, 1):
        r, rp = refhost.read_ply(b, str(tmp_path))
        m, mp = gltf_io.read_ply(b)
        assert rp == mp and np.array_equal(r.view(np.uint32), m.view(np.uint32))


# ---- JPEG textures (stb_image through tiny_gltf in the reference; mesh2splat_amd/csrc/m2s_jpeg.cpp here) -----------
def _jpeg_cases():
    PIL = pytest.importorskip("PIL.Image")
    import io
    rng = np.random.default_rng(12)

    def smooth(w, h):
        y, x = np.mgrid[0:h, 0:w]
        a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 5.0 + y / 9.0), 128 + 80 * np.sin((x + y) / 13.0)], -1)
        return np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)

    def enc(arr, **kw):
        b = io.BytesIO()
        PIL.fromarray(arr).save(b, "JPEG", **kw)
        return b.getvalue()

    yield "q90_444", enc(smooth(64, 48), quality=90, subsampling=0)
    yield "q75_420_odd", enc(smooth(67, 45), quality=75, subsampling=2)
    yield "q60_422", enc(smooth(33, 70), quality=60, subsampling=1)
    yield "progressive_420", enc(smooth(100, 60), quality=80, subsampling=2, progressive=True)
    yield "progressive_444", enc(smooth(37, 29), quality=95, subsampling=0, progressive=True)
    yield "grayscale", enc(smooth(50, 50)[..., 0], quality=85)
    yield "noise_420_q50", enc(rng.integers(0, 256, (64, 64, 3), dtype=np.uint8), quality=50, subsampling=2)
    yield "restart_markers", enc(smooth(128, 96), quality=85, subsampling=2, restart_marker_blocks=3)
    yield "one_pixel", enc(smooth(1, 1), quality=90)
    yield "one_column_420", enc(smooth(1, 17), quality=90, subsampling=2)
    yield "optimized_huffman", enc(smooth(80, 80), quality=70, optimize=True, subsampling=2)
    yield "large_progressive", enc(smooth(512, 384), quality=88, subsampling=2, progressive=True, optimize=True)
    yield "q100", enc(smooth(40, 40), quality=100, subsampling=0)
    yield "q5", enc(smooth(96, 64), quality=5, subsampling=2)


@live
def test_jpeg_textures_match_reference_live(tmp_path, hiplib):
    """Every decoded texel equals what the reference's loader (tiny_gltf -> stb_image) produces: baseline and
    progressive, 4:4:4 / 4:2:2 / 4:2:0, grayscale, restart markers, odd sizes, extreme qualities."""
    n = 0
    for name, jpg in _jpeg_cases():
        scene = synth.cube_sphere(2, tex_size=8)
        glb = str(tmp_path / (name + ".glb"))
        gltf_io.write_glb(scene, glb, png_override={"baseColorTexture": jpg, "normalTexture": jpg})
        ref = refhost.load_scene(glb, str(tmp_path))
        mine = gltf_io.load_glb(glb)
        for key in ("baseColorTexture", "normalTexture"):
            a, b = ref[0]["textures"][key], mine.meshes[0].textures[key]
            assert a.shape == b.shape and np.array_equal(a, b), (name, key)
        n += 1
    assert n >= 14


@pytest.mark.parametrize("name", ["jpeg_baseline_420", "jpeg_progressive_422"])
def test_jpeg_textures_match_reference_golden(hiplib, name):
    with open(os.path.join(GOLD, name + ".scene.bin"), "rb") as f:
        ref = refhost.parse_scene_dump(f.read())
    assert_scene_equal(ref, gltf_io.load_glb(os.path.join(GOLD, name + ".glb")))


def test_jpeg_errors(tmp_path, hiplib):
    from mesh2splat_amd._lib import M2SError
    scene = synth.cube_sphere(2, tex_size=8)
    for bad in (b"\xff\xd8\xff\xe0\x00\x02", b"\xff\xd8\xff\xc9\x00\x0b\x08\x00\x08\x00\x08\x01\x01\x11\x00"):
        glb = str(tmp_path / "bad.glb")
        gltf_io.write_glb(scene, glb, png_override={"baseColorTexture": bad})
        with pytest.raises(M2SError):
            gltf_io.load_glb(glb)


def _rewrite_glb_json(path, fn):
    """Apply fn(doc) to the JSON chunk of a .glb (test helper)."""
    import json
    import struct
    raw = open(path, "rb").read()
    jlen = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jlen].decode())
    rest = raw[20 + jlen:]
    fn(doc)
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * ((4 - len(js) % 4) % 4)
    out = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + len(rest)) + struct.pack("<II", len(js), 0x4E4F534A) + js + rest
    open(path, "wb").write(out)


@live
def test_image_uris_match_reference_live(tmp_path, hiplib):
    """Images referenced by a base64 data: URI or by a file next to the .glb (tiny_gltf resolves both)."""
    import base64
    scene = synth.cube_sphere(2, tex_size=8)
    png = gltf_io.encode_png(scene.meshes[0].textures["baseColorTexture"])
    glb = str(tmp_path / "uri.glb")
    gltf_io.write_glb(scene, glb)
    (tmp_path / "my tex.png").write_bytes(png)

    def edit(doc):
        doc["images"][0] = {"uri": "data:image/png;base64," + base64.b64encode(png).decode()}
        doc["images"][1] = {"uri": "my%20tex.png"}
    _rewrite_glb_json(glb, edit)
    assert_scene_equal(refhost.load_scene(glb, str(tmp_path)), gltf_io.load_glb(glb))


# ---- glTF features that real assets (SciFiHelmet, Sponza) carry and the synthetic writer does not produce ------------
def _edit_glb(src, dst, edit):
    """Re-pack a .glb with its JSON chunk edited by `edit(doc, bin_bytes) -> bin_bytes`."""
    import json
    import struct
    raw = open(src, "rb").read()
    jl = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jl])
    pos = 20 + jl
    bl = struct.unpack_from("<I", raw, pos)[0]
    binc = bytearray(raw[pos + 8: pos + 8 + bl])
    binc = edit(doc, binc) or binc
    doc["buffers"][0]["byteLength"] = len(binc)
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    binc += b"\0" * (-len(binc) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binc), 0x004E4942) + bytes(binc)
    open(dst, "wb").write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)


def _feature_edits():
    import struct

    def second_uv_set(doc, binc):
        # TEXCOORD_1 (different values) + materials that ask for texCoord 1 + KHR_texture_transform
        for mesh in doc["meshes"]:
            for prim in mesh["primitives"]:
                a0 = doc["accessors"][prim["attributes"]["TEXCOORD_0"]]
                n = a0["count"]
                off = len(binc)
                binc += struct.pack(f"<{2 * n}f", *([0.25, 0.75] * n))
                doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": 8 * n})
                doc["accessors"].append({"bufferView": len(doc["bufferViews"]) - 1, "componentType": 5126, "count": n, "type": "VEC2"})
                prim["attributes"]["TEXCOORD_1"] = len(doc["accessors"]) - 1
        for mat in doc.get("materials", []):
            pbr = mat.setdefault("pbrMetallicRoughness", {})
            for key in ("baseColorTexture", "metallicRoughnessTexture"):
                if key in pbr:
                    pbr[key]["texCoord"] = 1
                    pbr[key]["extensions"] = {"KHR_texture_transform": {"offset": [0.5, 0.25], "scale": [2.0, 3.0], "rotation": 0.3, "texCoord": 0}}
            if "normalTexture" in mat:
                mat["normalTexture"]["scale"] = 0.5
                mat["normalTexture"]["texCoord"] = 1
        doc["extensionsUsed"] = ["KHR_texture_transform"]
        return binc

    def material_extras(doc, binc):
        for i, mat in enumerate(doc.get("materials", [])):
            pbr = mat.setdefault("pbrMetallicRoughness", {})
            pbr["metallicFactor"], pbr["roughnessFactor"] = 0.3, 0.8
            mat["alphaMode"], mat["alphaCutoff"], mat["doubleSided"] = ("MASK", 0.4, True) if i % 2 else ("BLEND", 0.5, False)
            mat["emissiveFactor"] = [0.1, 0.2, 0.3]
            if "baseColorTexture" in pbr:
                mat["occlusionTexture"] = {"index": pbr["baseColorTexture"]["index"], "strength": 0.7}
                mat["emissiveTexture"] = {"index": pbr["baseColorTexture"]["index"]}
        doc["cameras"] = [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.1}}]
        doc["nodes"].append({"camera": 0, "translation": [0, 0, 5]})
        doc["scenes"][0]["nodes"].append(len(doc["nodes"]) - 1)
        doc["animations"] = []
        return binc

    def sparse_positions(doc, binc):
        # a sparse substitution on POSITION: the reference reads the base bufferView and ignores `sparse`
        prim = doc["meshes"][0]["primitives"][0]
        acc = doc["accessors"][prim["attributes"]["POSITION"]]
        o_idx, o_val = len(binc), len(binc) + 8
        binc += struct.pack("<2I", 0, 1) + struct.pack("<6f", 9, 9, 9, -9, -9, -9)
        doc["bufferViews"].append({"buffer": 0, "byteOffset": o_idx, "byteLength": 8})
        doc["bufferViews"].append({"buffer": 0, "byteOffset": o_val, "byteLength": 24})
        acc["sparse"] = {"count": 2, "indices": {"bufferView": len(doc["bufferViews"]) - 2, "componentType": 5125},
                         "values": {"bufferView": len(doc["bufferViews"]) - 1}}
        return binc

    def second_primitive_and_lines(doc, binc):
        # a mesh with two triangle primitives and one LINES primitive (mode 1)
        m0 = doc["meshes"][0]
        p = dict(m0["primitives"][0])
        m0["primitives"].append(dict(p))
        lines = dict(p)
        lines["mode"] = 1
        m0["primitives"].append(lines)
        return binc

    return [("second_uv_set_and_texture_transform", second_uv_set), ("material_extras_cameras", material_extras),
            ("sparse_accessor", sparse_positions), ("several_primitives_and_lines", second_primitive_and_lines)]


@live
@pytest.mark.parametrize("feature", _feature_edits(), ids=lambda f: f[0])
def test_loader_matches_reference_on_real_asset_features(tmp_path, hiplib, feature):
    """What SciFiHelmet / Sponza-class files contain beyond the synthetic scenes: a second UV set and texCoord indices,
    KHR_texture_transform, normal scale / occlusion / emissive / alpha modes, cameras, sparse accessors, several primitives per
    mesh and non-triangle primitives.  Whatever the reference's loader makes of them (mostly: ignores them), ours makes the same."""
    name, edit = feature
    base = str(tmp_path / "base.glb")
    gltf_io.write_glb(synth.sphere_grid(2, n=3, tex_size=16), base)
    glb = str(tmp_path / (name + ".glb"))
    _edit_glb(base, glb, edit)
    assert_scene_equal(refhost.load_scene(glb, str(tmp_path)), gltf_io.load_glb(glb))


def test_decoders_match_reference_on_generated_images(tmp_path):
    """Differential test of the PNG / JPEG decoders against the reference's loader (tiny_gltf -> stb_image, compiled from the
    reference) on ~150 VALID files written by Pillow: noise / gradient / checker content, odd sizes, qualities 1..100, 4:4:4 /
    4:2:2 / 4:2:0, baseline and progressive, optimised tables, restart intervals; PNG grey / grey+alpha / RGB / RGBA / palette
    (+ tRNS) / 1-bit at several compression levels.  Every decoded byte must be the reference's."""
    PIL = pytest.importorskip("PIL")
    if not refhost.available():
        pytest.skip("oracle/_ref/ref_host_check not built (needs /root/reference at build time)")
    import sys
    from PIL import Image
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import diff_decoders
    rng = np.random.default_rng(321)
    d = tmp_path / "img"
    d.mkdir()
    n = 0

    def pic(w, h, kind):
        y, x = np.mgrid[0:h, 0:w]
        if kind == 0:
            return rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        if kind == 1:
            return np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x ^ y) * 8) % 256, 255 - (x + y) % 256], -1).astype(np.uint8)
        return np.full((h, w, 4), ((x + y) % 2 * 255)[..., None], np.uint8)

    for (w, h) in ((1, 1), (9, 7), (17, 33), (64, 40), (130, 67)):
        for kind in (0, 1, 2):
            a = pic(w, h, kind)
            for mode in ("L", "RGB"):
                im = Image.fromarray(a, "RGBA").convert(mode)
                for q, ss in ((1, 0), (35, 2), (75, 1), (100, 0)):
                    if mode == "L" and ss:
                        ss = 0
                    kw = dict(quality=q, subsampling=ss)
                    r = rng.random()
                    if r < 0.35:
                        kw["progressive"] = True
                    if r > 0.7:
                        kw["optimize"] = True
                    if w * h > 64 and rng.random() < 0.3:
                        kw["restart_marker_blocks"] = int(rng.integers(1, 5))
                    im.save(str(d / f"v{n}.jpg"), **kw)
                    n += 1
            for mode in ("L", "LA", "RGB", "RGBA"):
                Image.fromarray(a, "RGBA").convert(mode).save(str(d / f"v{n}.png"), compress_level=int(rng.choice([0, 1, 6, 9])))
                n += 1
            p = Image.fromarray(a, "RGBA").convert("RGB").convert("P", palette=Image.ADAPTIVE, colors=int(rng.choice([2, 16, 200])))
            p.save(str(d / f"v{n}.png"), **({"transparency": 0} if rng.random() < 0.5 else {}))
            n += 1
            Image.fromarray(a, "RGBA").convert("1").save(str(d / f"v{n}.png"))
            n += 1
    old_argv = sys.argv
    sys.argv = ["diff_decoders", str(d)]
    try:
        assert diff_decoders.main() == 0, "a decoded image differs from the reference's (see the captured output)"
    finally:
        sys.argv = old_argv
