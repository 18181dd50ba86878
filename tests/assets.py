"""Test infrastructure: .glb files SHAPED like the two assets BASELINE.json names and no box here has (no network, none in the
reference: SURVEY 8c) — authored by the REFERENCE's own tiny_gltf + stb_image_write through oracle/_ref/ref_host_check glbwrite2, so
that what the loader (m2s_gltf.cpp, following SceneManager.cpp:195-459) meets is a file it did not write:

  helmet_like   BASELINE configs[1] "SciFiHelmet.glb": ONE primitive of 70 074 triangles, indexed with uint32 and real vertex reuse,
                attribute views that state their byteStride (flags 64, like the sample asset's exporter; flags 2: ONE interleaved
                view of byteStride 48 — which the reference's getBufferData, SceneManager.cpp:50-61, reads as if it were tightly
                packed: such a file is checked against the separate-views file, not against the reference's loader),
                a node with translation / rotation / non-uniform scale, a six-chart
                UV atlas with seams (vertices doubled along chart borders), two charts outside [0, 1] (REPEAT), tangents of both
                handednesses, four tex^2 maps: base colour + normal as PNG, metallic-roughness + occlusion as JPEG (the
                reference ignores occlusion);
  sponza_like   BASELINE configs[3] "Sponza.glb": ONE mesh of 103 primitives under one node, 25 materials sharing 34 images
                (several materials on one normal map; four materials with a base-colour map only, three with no map), two-triangle
                floor and walls beside columns, panels, props, dense curtains and sub-pixel foliage; ~276 k triangles.
The generators are deterministic across machines: fixed-seed numpy bit generators and only correctly rounded operations (+ - * /
sqrt floor) — the waves are parabolic "sines" (psin below), not libm's, whose last bit depends on the CPU's vector unit; stb's encoders
are plain C.  tests/golden/asset_hashes.json pins the sha256 of the .glb files they produce."""
from __future__ import annotations

import hashlib
import os
import struct
import subprocess

import numpy as np

from mesh2splat_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_host_check")


def available() -> bool:
    return os.path.isfile(BIN) and os.access(BIN, os.X_OK)


def psin(turns):
    """parabolic sine of period 1 (argument in TURNS): 16 s (1/2 - s) on the first half period, mirrored on the second; C1"""
    s = turns - np.floor(turns)
    return np.where(s < 0.5, 16.0 * s * (0.5 - s), -16.0 * (s - 0.5) * (1.0 - s))


def pcos(turns):
    return psin(turns + 0.25)


def dpsin(turns):
    """d psin / d turns"""
    s = turns - np.floor(turns)
    return np.where(s < 0.5, 8.0 - 32.0 * s, 32.0 * s - 24.0)


def _norm(v):
    return np.sqrt(v[..., 0] * v[..., 0] + v[..., 1] * v[..., 1] + v[..., 2] * v[..., 2])[..., None]


# ---- indexed parametric grids -----------------------------------------------------------------------------------------------------
def grid(P, dPds, dPdt, uv, tangent_w=1.0):
    """(nv+1, nu+1, .) arrays -> (pos, normal, tangent, uv, index): one vertex per grid point, two triangles per cell."""
    n = np.stack([dPds[..., 1] * dPdt[..., 2] - dPds[..., 2] * dPdt[..., 1], dPds[..., 2] * dPdt[..., 0] - dPds[..., 0] * dPdt[..., 2],
                  dPds[..., 0] * dPdt[..., 1] - dPds[..., 1] * dPdt[..., 0]], -1)
    n = n / np.maximum(_norm(n), 1e-30)
    t = dPds / np.maximum(_norm(dPds), 1e-30)
    nv1, nu1 = P.shape[:2]
    tan = np.concatenate([t, np.full(P.shape[:2] + (1,), float(tangent_w))], -1)
    i = np.arange(nv1 * nu1, dtype=np.uint32).reshape(nv1, nu1)
    a00, a10, a01, a11 = i[:-1, :-1], i[:-1, 1:], i[1:, :-1], i[1:, 1:]
    idx = np.stack([a00, a10, a11, a00, a11, a01], -1).reshape(-1)
    f = np.float32
    return P.reshape(-1, 3).astype(f), n.reshape(-1, 3).astype(f), tan.reshape(-1, 4).astype(f), uv.reshape(-1, 2).astype(f), idx


def plane(nu, nv, origin, U, V, amp=0.0, waves=(3.0, 2.0), uv_tile=1.0):
    origin, U, V = (np.asarray(x, np.float64) for x in (origin, U, V))
    N = np.array([U[1] * V[2] - U[2] * V[1], U[2] * V[0] - U[0] * V[2], U[0] * V[1] - U[1] * V[0]])
    N = N / np.sqrt(N[0] * N[0] + N[1] * N[1] + N[2] * N[2])
    s, t = np.meshgrid(np.arange(nu + 1) / float(nu), np.arange(nv + 1) / float(nv), indexing="xy")
    w0, w1 = waves
    d = amp * psin(w0 * s) * psin(w1 * t)
    P = origin + s[..., None] * U + t[..., None] * V + d[..., None] * N
    dPds = U + (amp * w0 * dpsin(w0 * s) * psin(w1 * t))[..., None] * N
    dPdt = V + (amp * w1 * psin(w0 * s) * dpsin(w1 * t))[..., None] * N
    return grid(P, dPds, dPdt, np.stack([s * uv_tile, t * uv_tile], -1))


def cylinder(seg, rings, base, radius, height, uv_tile=1.0):
    base = np.asarray(base, np.float64)
    th, t = np.meshgrid(np.arange(seg + 1) / float(seg), np.arange(rings + 1) / float(rings), indexing="xy")    # th in turns
    c, s_ = pcos(th), psin(th)
    P = base + np.stack([radius * c, height * t, radius * s_], -1)
    dPds = np.stack([radius * dpsin(th + 0.25), np.zeros_like(th), radius * dpsin(th)], -1)
    dPdt = np.broadcast_to(np.array([0.0, height, 0.0]), P.shape).copy()
    return grid(P, dPdt, dPds, np.stack([th * uv_tile, t * uv_tile], -1))


def merge(parts):
    """several indexed parts -> one primitive (indices rebased; vertices along the parts' borders stay doubled: UV seams)"""
    pos, nrm, tan, uv, idx, base = [], [], [], [], [], 0
    for p, n, t, u, i in parts:
        pos.append(p); nrm.append(n); tan.append(t); uv.append(u); idx.append(i + np.uint32(base))
        base += p.shape[0]
    return np.concatenate(pos), np.concatenate(nrm), np.concatenate(tan), np.concatenate(uv), np.concatenate(idx)


IDENTITY = ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0, 1.0), (1.0, 1.0, 1.0))


# ---- the two files ----------------------------------------------------------------------------------------------------------------
def helmet_like(tex: int = 2048) -> dict:
    n = 76                                           # six charts of 76 x 76 cells = 69 312 triangles ...
    charts = []
    faces = [((1, 0, 0), (0, 1, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)), ((0, 1, 0), (0, 0, 1), (1, 0, 0)),
             ((0, -1, 0), (1, 0, 0), (0, 0, 1)), ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (0, 1, 0), (1, 0, 0))]
    eps = 1e-4

    def shell(c, a, b, s, t):
        c, a, b = (np.asarray(x, np.float64) for x in (c, a, b))
        q = c + (2 * s[..., None] - 1) * a + (2 * t[..., None] - 1) * b
        d = q / _norm(q)
        r = 0.5 * (1.0 + 0.08 * psin(0.8 * d[..., 0]) * pcos(0.65 * d[..., 1]) + 0.05 * psin(1.1 * d[..., 2]))   # dented sphere
        return d * r[..., None]

    for k, (c, a, b) in enumerate(faces):
        s, t = np.meshgrid(np.arange(n + 1) / float(n), np.arange(n + 1) / float(n), indexing="xy")
        P = shell(c, a, b, s, t)
        dPds = (shell(c, a, b, s + eps, t) - shell(c, a, b, s - eps, t)) / (2 * eps)
        dPdt = (shell(c, a, b, s, t + eps) - shell(c, a, b, s, t - eps)) / (2 * eps)
        # atlas: 3 x 2 cells with a margin; chart 4 lies one tile to the right (u in [1, 2)), chart 5 one tile below (v < 0)
        cu, cv = k % 3, k // 3
        u = (cu + 0.04 + 0.92 * s) / 3.0 + (1.0 if k == 4 else 0.0)
        v = (cv + 0.04 + 0.92 * t) / 2.0 - (1.0 if k == 5 else 0.0)
        charts.append(grid(P, dPds, dPdt, np.stack([u, v], -1), tangent_w=-1.0 if k in (1, 3) else 1.0))
    # ... + a visor strip of 127 x 3 cells = 762 triangles: 70 074, the real file's triangle count
    charts.append(plane(127, 3, (-0.3, 0.05, 0.52), (0.6, 0.0, 0.0), (0.0, 0.1, 0.03), amp=0.01, waves=(2.0, 1.0), uv_tile=1.0))
    pos, nrm, tan, uv, idx = merge(charts)
    assert idx.shape[0] == 3 * 70074
    t = synth.procedural_textures(tex, synth.SEED + 11)
    yy, xx = np.mgrid[0:tex, 0:tex]
    occl = np.empty((tex, tex, 4), np.uint8)
    occl[..., 0] = occl[..., 1] = occl[..., 2] = np.floor(128.0 + 127.0 * psin(xx / float(tex)) * pcos(yy / float(tex))).astype(np.uint8)
    occl[..., 3] = 255
    images = [(t["baseColorTexture"], 0), (t["normalTexture"], 0), (t["metallicRoughnessTexture"], 1), (occl, 1)]
    # 30 degrees about (1, 1, 0) / sqrt 2: (sin 15 / sqrt 2, sin 15 / sqrt 2, 0, cos 15), as literals
    trs = ((0.3, -0.2, 0.1), (0.18301270189221933, 0.18301270189221933, 0.0, 0.9659258262890683), (1.0, 1.2, 0.9))
    return {"flags": 64, "images": images, "materials": [("helmet", (1.0, 1.0, 1.0, 1.0), (0, 1, 2, 3))],
            "prims": [("SciFiHelmet", pos, nrm, tan, uv, idx, 0, trs)]}


def sponza_like(tex_scale: float = 1.0) -> dict:
    rng = np.random.default_rng(20240601)
    axes = np.eye(3)

    def oriented(k):
        return axes[(k + 1) % 3], axes[(k + 2) % 3]

    def size(px):
        return max(16, int(px * tex_scale))

    # 34 images: 18 base-colour maps, 8 normal maps (shared), 8 metallic-roughness maps (shared); every third one JPEG
    images = []
    for k in range(18):
        images.append((synth.procedural_textures(size((1024, 512, 256)[k % 3]), synth.SEED + 100 + k)["baseColorTexture"], 1 if k % 3 == 2 else 0))
    for k in range(8):
        images.append((synth.procedural_textures(size((1024, 512)[k % 2]), synth.SEED + 200 + k)["normalTexture"], 0))
    for k in range(8):
        images.append((synth.procedural_textures(size((1024, 512)[k % 2]), synth.SEED + 300 + k)["metallicRoughnessTexture"], 1 if k % 2 else 0))
    materials = []
    for k in range(25):
        col = (0.6 + 0.4 * ((k * 37) % 11) / 10.0, 0.6 + 0.4 * ((k * 17) % 7) / 6.0, 0.9, 1.0)
        if k < 18:
            # (a normal / metallic-roughness map of the base-colour map's size where there is one, so that both kinds of sampler run)
            a = k
            nmap = 18 + (k % 8) if k % 5 else 18 + ((k % 3) % 2)
            mmap = 26 + (k % 8) if k % 5 else 26 + ((k % 3) % 2)
            materials.append(("mat_%02d" % k, col, (a, nmap, mmap, -1)))
        elif k < 22:
            materials.append(("mat_%02d_albedo_only" % k, col, (k - 18, -1, -1, -1)))
        else:
            materials.append(("mat_%02d_plain" % k, col, (-1, -1, -1, -1)))
    prims = []

    def add(name, g, mat):
        prims.append((name, *g, int(mat), IDENTITY))

    add("floor", plane(1, 1, (0, 0, 1), (1, 0, 0), (0, 0, -1), uv_tile=8.0), 0)
    for i in range(4):
        k = i % 3
        U, V = oriented(k)
        o = rng.uniform(0.05, 0.55, 3)
        add("wall_%d" % i, plane(1, 1, o, 0.4 * U, 0.4 * V, uv_tile=4.0), 1 + i % 2)
    rest = []
    for i in range(20):
        r, hgt = rng.uniform(0.015, 0.04), rng.uniform(0.3, 0.6)
        rest.append(("column_%d" % i, cylinder(32, 16, (rng.uniform(0.05, 0.95), 0.0, rng.uniform(0.05, 0.95)), r, hgt), 3 + i % 4))
    for i in range(12):
        sz = rng.uniform(0.1, 0.4)
        U, V = oriented(int(rng.integers(0, 3)))
        tilt = rng.uniform(-0.25, 0.25, 3)
        o = rng.uniform(0.0, 1.0 - sz, 3) * (1.0, 0.5, 1.0)
        rest.append(("panel_%d" % i, plane(30, 30, o, sz * (U + tilt * 0.5), sz * (V - tilt * 0.5), amp=0.01 * sz), 7 + i % 6))
    for i in range(44):
        u_ = float(rng.uniform(0.0, 1.0))
        sz = 0.02 * (1.0 + 14.0 * u_ * u_)                     # 0.02 ... 0.3, most of them small
        U, V = oriented(int(rng.integers(0, 3)))
        o = rng.uniform(0.0, 1.0 - sz, 3) * (1.0, 0.5, 1.0)
        rest.append(("prop_%d" % i, plane(12, 12, o, sz * U, sz * V, amp=0.05 * sz, waves=(1.0, 1.0)), (13 + i % 12) if i % 7 else 22 + i % 3))
    for i in range(16):
        sz = rng.uniform(0.1, 0.4)
        U, V = oriented(int(rng.integers(0, 3)))
        o = rng.uniform(0.0, 1.0 - sz, 3) * (1.0, 0.5, 1.0)
        rest.append(("curtain_%d" % i, plane(60, 60, o, sz * U, sz * V, amp=0.02, waves=(5.0, 3.0), uv_tile=2.0), 9 + i % 9))
    for i in range(6):
        U, V = oriented(i % 3)
        o = rng.uniform(0.1, 0.8, 3) * (1.0, 0.5, 1.0)
        rest.append(("foliage_%d" % i, plane(96, 96, o, 0.05 * U, 0.05 * V, amp=0.004, waves=(9.0, 7.0)), 18 + i % 4))
    for j in rng.permutation(len(rest)):
        add(*rest[j])
    assert len(prims) == 103 and len(materials) == 25 and len(images) == 34
    return {"flags": 32, "images": images, "materials": materials, "prims": prims}


# ---- spec2.bin + authoring ----------------------------------------------------------------------------------------------------------
def write_spec2(spec: dict, path: str):
    with open(path, "wb") as f:
        f.write(struct.pack("<II", int(spec["flags"]), len(spec["images"])))
        for img, jpeg in spec["images"]:
            img = np.ascontiguousarray(img, np.uint8)
            f.write(struct.pack("<III", img.shape[1], img.shape[0], int(jpeg)) + img.tobytes())
        f.write(struct.pack("<I", len(spec["materials"])))
        for name, col, tex in spec["materials"]:
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + np.asarray(col, np.float32).tobytes() + np.asarray(tex, np.int32).tobytes())
        f.write(struct.pack("<I", len(spec["prims"])))
        for name, pos, nrm, tan, uv, idx, mat, (t, r, s) in spec["prims"]:
            nb = name.encode()
            f.write(struct.pack("<I", len(nb)) + nb + struct.pack("<I", pos.shape[0]))
            for a in (pos, nrm, tan, uv):
                f.write(np.ascontiguousarray(a, np.float32).tobytes())
            f.write(struct.pack("<I", idx.shape[0]) + np.ascontiguousarray(idx, np.uint32).tobytes() + struct.pack("<I", mat))
            f.write(np.asarray(t, np.float32).tobytes() + np.asarray(r, np.float32).tobytes() + np.asarray(s, np.float32).tobytes())


def author(spec: dict, glb_path: str, tmp_dir: str) -> str:
    """spec -> .glb written by the reference's tiny_gltf; returns the file's sha256"""
    sp = os.path.join(tmp_dir, os.path.basename(glb_path) + ".spec2")
    write_spec2(spec, sp)
    r = subprocess.run([BIN, "glbwrite2", sp, glb_path], capture_output=True, text=True, timeout=600)
    os.unlink(sp)
    if r.returncode != 0:
        raise RuntimeError("ref_host_check glbwrite2 failed rc=%d: %s" % (r.returncode, r.stderr[-400:]))
    h = hashlib.sha256()
    with open(glb_path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 22), b""):
            h.update(chunk)
    return h.hexdigest()


def n_triangles(spec: dict) -> int:
    return sum(p[5].shape[0] // 3 for p in spec["prims"])
