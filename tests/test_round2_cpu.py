"""CPU: round-2 host logic — slice writers, the native shard plan, the logf restatement of the device-side row encoder,
and the hardening of the two file readers against crafted input (no GPU, no compute calls)."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from mesh2splat_amd import dist as m2d
from mesh2splat_amd import synth
from mesh2splat_amd.converter import write_ply, write_ply_slice

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(n, seed=3):
    rng = np.random.default_rng(seed)
    r = rng.random((n, 24), dtype=np.float32)
    r[:, 8:11] = r[:, 8:11] * 0.01 + 1e-4          # scale > 0
    r[:, 12:15] = r[:, 12:15] * 2 - 1               # normals of both signs (octahedral wrap)
    r[::7, 7] = 1.0                                 # opaque -> +inf opacity (Q8)
    return r


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_slice_writers_reproduce_the_whole_file(tmp_path, hiplib, fmt):
    """Several writers, one file: any split, any order of the writers, a stale longer file underneath."""
    rec = _records(10_000)
    whole = tmp_path / "whole.ply"
    write_ply(str(whole), rec, fmt, 0.65 / 128)
    want = whole.read_bytes()
    for cuts in ([0, 10_000], [0, 1, 9_999, 10_000], [0, 2_500, 2_500, 7_000, 10_000]):
        p = tmp_path / f"sliced_{len(cuts)}.ply"
        p.write_bytes(b"x" * (len(want) + 1000))      # leftovers of an older, longer export
        parts = list(zip(cuts[:-1], cuts[1:]))
        for a, b in reversed(parts):                   # the header's writer comes LAST here
            write_ply_slice(str(p), rec[a:b], fmt, 0.65 / 128, a, len(rec))
        assert p.read_bytes() == want
    with pytest.raises(Exception):
        write_ply_slice(str(tmp_path / "bad.ply"), rec[:10], fmt, 1.0, 5, 10)   # slice beyond the file


def test_native_shard_plan_equals_python_plan(hiplib):
    for scene, R in ((synth.sphere_grid(2, n=6), 128), (synth.cube_sphere(24, tex_size=16), 256),
                     (synth.sphere_grid(3, n=5, tex_size=8), 64), (synth.unit_quad(), 64)):
        est = m2d.estimate_fragments(scene, R)
        for world in (1, 2, 3, 4, 8):
            assert m2d.shard_ranges_native(scene, R, world) == m2d.shard_ranges(est, world)


def test_logf_restatement_matches_libm():
    """m2s_export.hip's logf_glibc, restated in C with the same constants, against the host's logf (every 61st positive float)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "_build/logf_check"], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "logf_check"), "61"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "nofma=0 fma=0" in r.stdout
    # the constants in the device source are the ones the check program uses
    dev = open(os.path.join(ROOT, "mesh2splat_amd", "csrc", "m2s_export.hip")).read()
    chk = open(os.path.join(ROOT, "oracle", "logf_check.c")).read()
    import re
    consts = set(re.findall(r"-?0x1\.[0-9a-f]+p[+-]\d+", chk))
    assert len(consts) >= 34 and all(c in dev for c in consts)


# ---- crafted input -------------------------------------------------------------------------------------------------
def _glb(doc: dict, bin_chunk: bytes) -> bytes:
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    bin_chunk += b"\0" * (-len(bin_chunk) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(bin_chunk), 0x004E4942) + bin_chunk
    return struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body


def _load(hiplib, path):
    import ctypes as C
    h = C.c_void_p()
    st = hiplib.m2s_load_glb(os.fsencode(str(path)), C.byref(h))
    if st == 0:
        hiplib.m2s_free_host_scene(h)
    return st, hiplib.m2s_io_last_error().decode()


def _tri_doc(**acc_over):
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32).tobytes()
    idx = np.array([0, 1, 2], np.uint16).tobytes() + b"\0\0"
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
           "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
           "buffers": [{"byteLength": len(pos) + len(idx)}],
           "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(pos)}, {"buffer": 0, "byteOffset": len(pos), "byteLength": 6}],
           "accessors": [{"bufferView": 0, "componentType": 5126, "count": 3, "type": "VEC3"},
                         {"bufferView": 1, "componentType": 5123, "count": 3, "type": "SCALAR"}]}
    doc["accessors"][0].update(acc_over)
    return doc, pos + idx


def test_glb_accessor_bounds_cannot_wrap(tmp_path, hiplib):
    doc, b = _tri_doc()
    p = tmp_path / "ok.glb"
    p.write_bytes(_glb(doc, b))
    assert _load(hiplib, p)[0] == 0
    for over in ({"count": -1}, {"count": 2 ** 62}, {"byteOffset": -8}, {"byteOffset": 2 ** 63 - 1}, {"count": 1e300},
                 {"count": 4}):
        doc, b = _tri_doc(**over)
        p = tmp_path / "bad.glb"
        p.write_bytes(_glb(doc, b))
        st, msg = _load(hiplib, p)
        assert st != 0 and msg, over
    doc, b = _tri_doc()
    doc["bufferViews"][0]["byteStride"] = -12
    p.write_bytes(_glb(doc, b))
    assert _load(hiplib, p)[0] != 0
    doc, b = _tri_doc()
    doc["bufferViews"][0]["byteOffset"] = 2 ** 64          # not representable in 64 bits
    p.write_bytes(_glb(doc, b))
    assert _load(hiplib, p)[0] != 0


def test_glb_json_nesting_is_bounded(tmp_path, hiplib):
    js = b'{"asset":{"version":"2.0"},"extras":' + b"[" * 200_000 + b"]" * 200_000 + b"}"
    js += b" " * (-len(js) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js
    p = tmp_path / "deep.glb"
    p.write_bytes(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)
    st, msg = _load(hiplib, p)
    assert st != 0 and "nesting" in msg


def test_glb_numbers_do_not_depend_on_the_locale(tmp_path, hiplib):
    """1.5 must stay 1.5 under a comma-decimal LC_NUMERIC (strtod would stop at the '.')."""
    import ctypes as C
    import locale
    doc, b = _tri_doc()
    doc["nodes"][0]["translation"] = [1.5, 0.25, -2.75]
    p = tmp_path / "t.glb"
    p.write_bytes(_glb(doc, b))
    old = locale.setlocale(locale.LC_NUMERIC)
    try:
        for cand in ("de_DE.UTF-8", "fr_FR.UTF-8", "de_DE", "C.UTF-8"):
            try:
                locale.setlocale(locale.LC_NUMERIC, cand)
                break
            except locale.Error:
                continue
        h = C.c_void_p()
        assert hiplib.m2s_load_glb(os.fsencode(str(p)), C.byref(h)) == 0
        m = hiplib.m2s_host_scene_meshes(h)[0]
        v = np.ctypeslib.as_array(C.cast(m.vertices, C.POINTER(C.c_float)), shape=(3, m.stride_floats))
        assert np.allclose(v[0, :3], [1.5, 0.25, -2.75])
        hiplib.m2s_free_host_scene(h)
    finally:
        locale.setlocale(locale.LC_NUMERIC, old)


def test_read_ply_distrusts_the_header(tmp_path, hiplib):
    import ctypes as C
    rec = _records(100)
    good = tmp_path / "good.ply"
    write_ply(str(good), rec, 1, 1.0)
    data = good.read_bytes()

    def read(path):
        out, n, pbr = C.c_void_p(), C.c_uint64(), C.c_int()
        st = hiplib.m2s_read_ply(os.fsencode(str(path)), C.byref(out), C.byref(n), C.byref(pbr))
        if st == 0:
            hiplib.m2s_free_records(out)
        return st, n.value
    assert read(good) == (0, 100)
    for claimed in (101, 2 ** 61, 2 ** 64 - 1):
        bad = tmp_path / "bad.ply"
        bad.write_bytes(data.replace(b"element vertex 100\n", b"element vertex %d\n" % claimed))
        st, n = read(bad)
        assert st != 0 and n == 0
    trunc = tmp_path / "trunc.ply"
    trunc.write_bytes(data[:-10])
    assert read(trunc)[0] != 0


def test_in_process_group_ids_and_image_header_guards(tmp_path, hiplib):
    """(a) m2s_dist_local_id: argument checks and the shape of the id, without touching a device; a world-size mismatch is
    refused before any device call matters.  (b) The readers refuse images whose header demands far more pixels than their data
    can describe (a 100-byte file must not make the loader allocate gigabytes)."""
    import ctypes as C
    from mesh2splat_amd import _lib
    L = _lib.load()
    buf = (C.c_uint8 * 128)()
    assert L.m2s_dist_local_id(0, buf) == 1   # M2S_ERR_INVALID
    assert L.m2s_dist_local_id(2, None) == 1   # M2S_ERR_INVALID
    assert L.m2s_dist_local_id(3, buf) == _lib.M2S_OK
    assert bytes(buf[:8]) == b"M2SLOCAL" and bytes(buf) != bytes(128)
    ident = m2d.local_group_id(2)
    assert len(ident) == 128 and ident[:8] == b"M2SLOCAL"
    # (b) a JPEG whose frame header claims 60000 x 60000 pixels, followed by a few bytes of "scan"; and the PNG counterpart
    import zlib
    from mesh2splat_amd import gltf_io
    sof = b"\xff\xc0" + struct.pack(">HBHHB", 17, 8, 20000, 20000, 3) + b"\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    dqt = b"\xff\xdb" + struct.pack(">H", 67) + b"\x00" + bytes([16] * 64)
    jpg = b"\xff\xd8" + dqt + sof + b"\xff\xda" + struct.pack(">HB", 12, 3) + b"\x01\x00\x02\x11\x03\x11\x00\x3f\x00" + b"\x00" * 64 + b"\xff\xd9"

    def chunk(t, body):
        return struct.pack(">I", len(body)) + t + body + struct.pack(">I", zlib.crc32(t + body))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 30000, 30000, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
    for name, blob, mime in (("bomb.jpg", jpg, "image/jpeg"), ("bomb.png", png, "image/png")):
        scene = synth.unit_quad(textures={"baseColorTexture": np.full((4, 4, 4), 200, np.uint8)})
        glb = str(tmp_path / (name + ".glb"))
        gltf_io.write_glb(scene, glb)
        raw = open(glb, "rb").read()
        jl = struct.unpack_from("<I", raw, 12)[0]
        doc = json.loads(raw[20:20 + jl])
        pos = 20 + jl
        bl = struct.unpack_from("<I", raw, pos)[0]
        binc = bytearray(raw[pos + 8: pos + 8 + bl])
        off = len(binc)
        binc += blob
        doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(blob)})
        doc["images"][0] = {"bufferView": len(doc["bufferViews"]) - 1, "mimeType": mime}
        doc["buffers"][0]["byteLength"] = len(binc)
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        binc += b"\0" * (-len(binc) % 4)
        body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binc), 0x004E4942) + bytes(binc)
        open(glb, "wb").write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)
        with pytest.raises(Exception) as e:
            gltf_io.load_glb(glb)
        assert "too short for its dimensions" in str(e.value) or "too large" in str(e.value), str(e.value)
