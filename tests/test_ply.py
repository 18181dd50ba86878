"""CPU: .ply export (K-9).  Three independent writers must agree byte for byte: the product's
m2s_write_ply (host C++), the oracle's restatement of parsers.cpp:232-514, and a struct.pack writer
in this file that follows the reference field by field."""
import ctypes
import math
import os
import struct

import numpy as np
import pytest

from mesh2splat_amd import synth
from mesh2splat_amd.converter import write_ply

SH_C0 = np.float32(0.28209479177387814)
_libm = ctypes.CDLL("libm.so.6")       # std::log(float) == glibc logf: use the same library function
_libm.logf.restype = ctypes.c_float
_libm.logf.argtypes = [ctypes.c_float]


def py_writer(records, fmt, sm):
    """Field-by-field transcription of parsers.cpp (standard 431-514, PBR 232-316, compressed 339-428)."""
    f32 = np.float32
    sm = f32(sm)

    def inv_sigmoid(a):
        a = f32(min(max(a, f32(0)), f32(1)))
        return f32(-_libm.logf(float(f32(f32(f32(1) / f32(a + f32(1e-8))) - f32(1)))))

    def to_byte(v):
        c = min(max(f32(v), f32(0)), f32(1))
        r = float(f32(c * f32(255)))
        return int(math.floor(r + 0.5))      # std::round for non-negative values

    def logf(x):
        return f32(_libm.logf(float(f32(x))))

    n = len(records)
    if fmt == 0:
        props = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(45)] + \
                ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
        head = "".join(f"property float {p}\n" for p in props)
    elif fmt == 1:
        props = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "metallicFactor", "roughnessFactor",
                 "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
        head = "".join(f"property float {p}\n" for p in props)
    else:
        head = ("property float x\nproperty float y\nproperty float z\n"
                "property uint8 red\nproperty uint8 green\nproperty uint8 blue\nproperty uint8 opacity\n"
                "property float rot_0\nproperty float rot_1\nproperty float rot_2\nproperty float rot_3\n"
                "property float scale_0\nproperty float scale_1\nproperty float scale_2\n"
                "property uint8 octa_nx\nproperty uint8 octa_ny\nproperty uint8 roughness\nproperty uint8 metallic\n")
    out = bytearray(f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\n{head}end_header\n".encode())
    for g in records:
        pos, col, scl, nrm, rot, pbr = g[0:4], g[4:8], g[8:12], g[12:16], g[16:20], g[20:24]
        if fmt in (0, 1):
            out += struct.pack("<6f", *pos[:3], *nrm[:3])
            out += struct.pack("<3f", *[f32(f32(col[k] - f32(0.5)) / SH_C0) for k in range(3)])
            out += struct.pack("<45f", *([0.0] * 45)) if fmt == 0 else struct.pack("<2f", pbr[0], pbr[1])
            out += struct.pack("<f", inv_sigmoid(col[3]))
            out += struct.pack("<3f", *[logf(f32(scl[k] * sm)) for k in range(3)])
            out += struct.pack("<4f", *rot)
        else:
            out += struct.pack("<3f", *pos[:3])
            out += bytes([to_byte(col[0]), to_byte(col[1]), to_byte(col[2]), to_byte(col[3])])
            out += struct.pack("<4f", *rot)
            mn = min(scl[0], scl[1])
            out += struct.pack("<3f", logf(f32(scl[0] * sm)), logf(f32(scl[1] * sm)), logf(f32(mn * sm)))
            d = f32(f32(f32(abs(nrm[0]) + abs(nrm[1])) + abs(nrm[2])) + f32(1e-8))
            nx, ny, nz = f32(nrm[0] / d), f32(nrm[1] / d), f32(nrm[2] / d)
            if nz >= 0:
                ex, ey = nx, ny
            else:
                s = f32(1.0) if (nx >= 0 and ny >= 0) else f32(-1.0)
                ex, ey = f32(f32(f32(1) - abs(ny)) * s), f32(f32(f32(1) - abs(nx)) * s)
            ex, ey = f32(f32(ex * f32(0.5)) + f32(0.5)), f32(f32(ey * f32(0.5)) + f32(0.5))

            def q(v):
                r = float(f32(v * f32(255)))
                r = math.floor(r + 0.5) if r >= 0 else -math.floor(-r + 0.5)
                return int(min(max(r, 0.0), 255.0))
            out += bytes([q(ex), q(ey), to_byte(pbr[1]), to_byte(pbr[0])])
    return bytes(out)


def sample_records(oracle):
    scene = synth.random_soup(60, seed=11, textures=synth.procedural_textures(16, 2))
    scene.meshes[0].base_color = (1.0, 0.9, 0.8, 1.0)
    _, rec, _ = oracle.convert(scene, 48, cap=0)
    # opaque fragments (alpha == 1 -> opacity +inf, Q8), semi-transparent, out-of-range colour
    rec = rec.copy()
    rec[::3, 7] = 1.0
    rec[1::3, 7] = 0.37
    rec[5, 4:7] = (1.5, -0.25, 0.5)
    rec[7, 12:15] = (0.0, 0.0, -1.0)
    rec[8, 12:15] = (-0.3, 0.2, -0.6)
    return rec


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_three_writers_agree(tmp_path, oracle, hiplib, fmt):
    rec = sample_records(oracle)
    assert len(rec) > 50
    sm = np.float32(0.65) / np.float32(48)
    a, b = str(tmp_path / "prod.ply"), str(tmp_path / "orc.ply")
    write_ply(a, rec, fmt, sm)
    oracle.write_ply(b, rec, fmt, sm)
    A, B = open(a, "rb").read(), open(b, "rb").read()
    assert A == B
    assert A == py_writer(rec, fmt, sm)
    row = {0: 248, 1: 76, 2: 48}[fmt]
    assert len(A) - A.index(b"end_header\n") - 11 == row * len(rec)


def test_opaque_is_plus_inf_and_default_format(tmp_path, oracle, hiplib):
    rec = sample_records(oracle)[:4].copy()
    rec[:, 7] = 1.0
    p = str(tmp_path / "a.ply")
    write_ply(p, rec, 7, 0.01)                   # unknown format -> standard (parsers.cpp:646-648)
    raw = open(p, "rb").read()
    body = np.frombuffer(raw[raw.index(b"end_header\n") + 11:], np.float32).reshape(4, 62)
    assert np.all(np.isposinf(body[:, 54]))      # opacity = -log(1/(1+1e-8) - 1) = +inf in fp32
    assert np.all(body[:, 9:54] == 0)
    assert np.allclose(body[:, 57], np.log(np.float32(1e-7) * np.float32(0.01)))


@pytest.mark.parametrize("fmt,rows", [(0, 300_000), (2, 600_000)])
def test_large_multithreaded_write(tmp_path, oracle, hiplib, fmt, rows):
    """More rows than one writer chunk (2^18): encoding threads + the double-buffered background file writes."""
    rng = np.random.default_rng(0)
    rec = rng.uniform(0.01, 1.0, (rows, 24)).astype(np.float32)
    a, b = str(tmp_path / "p.ply"), str(tmp_path / "o.ply")
    write_ply(a, rec, fmt, 0.001)
    oracle.write_ply(b, rec, fmt, 0.001)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_empty_and_io_error(tmp_path, hiplib):
    p = str(tmp_path / "e.ply")
    write_ply(p, np.zeros((0, 24), np.float32), 1, 1.0)
    assert open(p, "rb").read().startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 0\n")
    from mesh2splat_amd._lib import M2SError
    with pytest.raises(M2SError):
        write_ply(str(tmp_path / "no_such_dir" / "x.ply"), np.zeros((1, 24), np.float32), 0, 1.0)
