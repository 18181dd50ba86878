"""Deterministic synthetic scenes for parity tests and bench.py (no assets ship with the
reference and there is no network): SURVEY.md section 8(d) inputs I-1 .. I-5.

Every generator returns host-side :class:`mesh2splat_amd.scene.Scene` objects whose vertex
arrays use the reference's VBO layout (SceneManager.cpp:483-512: pos3 normal3 tangent4 uv2
[normalizedUv2 scale3]), de-indexed, three vertices per triangle.
"""
from __future__ import annotations

import numpy as np

from .scene import Mesh, Scene

SEED = 0x4D32535F5345454F  # "M2S_SEED"


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = x.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        x += np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return x


def _hash_u8(w: int, h: int, seed: int, lane: int) -> np.ndarray:
    idx = np.arange(w * h, dtype=np.uint64) + np.uint64((seed + lane * 0x1000003) & 0xFFFFFFFFFFFFFFFF)
    return (splitmix64(idx) >> np.uint64(56)).astype(np.uint8).reshape(h, w)


def procedural_textures(size: int, seed: int = SEED) -> dict:
    """Albedo = 32x32 checker x per-texel hash noise; normal = normalise((hx,hy,4));
    metallic-roughness: G = row gradient, B = 64x64 checker, A = 255."""
    w = h = int(size)
    yy, xx = np.mgrid[0:h, 0:w]
    cell = max(1, w // 32)
    checker = (((xx // cell) + (yy // cell)) & 1).astype(np.float32) * 0.5 + 0.5
    albedo = np.empty((h, w, 4), np.uint8)
    for c in range(3):
        noise = _hash_u8(w, h, seed, c).astype(np.float32)
        albedo[..., c] = np.clip(np.rint(noise * checker), 0, 255).astype(np.uint8)
    albedo[..., 3] = 255
    hx = _hash_u8(w, h, seed, 3).astype(np.float32) / 127.5 - 1.0
    hy = _hash_u8(w, h, seed, 4).astype(np.float32) / 127.5 - 1.0
    nz = np.full_like(hx, 4.0)
    inv = 1.0 / np.sqrt(hx * hx + hy * hy + nz * nz)
    normal = np.empty((h, w, 4), np.uint8)
    normal[..., 0] = np.clip(np.rint((hx * inv * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    normal[..., 1] = np.clip(np.rint((hy * inv * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    normal[..., 2] = np.clip(np.rint((nz * inv * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    normal[..., 3] = 255
    mr = np.empty((h, w, 4), np.uint8)
    mr[..., 0] = 0
    mr[..., 1] = np.clip(np.rint(yy * (255.0 / max(1, h - 1))), 0, 255).astype(np.uint8)
    cell2 = max(1, w // 64)
    mr[..., 2] = ((((xx // cell2) + (yy // cell2)) & 1) * 255).astype(np.uint8)
    mr[..., 3] = 255
    return {
        "baseColorTexture": np.ascontiguousarray(albedo),
        "normalTexture": np.ascontiguousarray(normal),
        "metallicRoughnessTexture": np.ascontiguousarray(mr),
    }


def unit_quad(textures: dict | None = None, stride: int = 17, name: str = "quad_0") -> Scene:
    """I-1: verts (0,0,0),(1,0,0),(1,1,0),(0,1,0); tris (0,1,2),(0,2,3); n=(0,0,1); t=(1,0,0,1); uv=xy."""
    corners = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32)
    idx = [0, 1, 2, 0, 2, 3]
    v = np.zeros((6, stride), np.float32)
    for k, i in enumerate(idx):
        v[k, 0:3] = corners[i]
        v[k, 3:6] = (0, 0, 1)
        v[k, 6:10] = (1, 0, 0, 1)
        v[k, 10:12] = corners[i, :2]
    m = Mesh(name=name, vertices=v, base_color=(1.0, 1.0, 1.0, 1.0), textures=dict(textures or {}))
    return Scene([m])


def cube_sphere_vertices(n: int, radius: float = 1.0, center=(0.0, 0.0, 0.0), stride: int = 12) -> np.ndarray:
    """Cube-sphere with n x n quads per cube face => 12*n*n triangles, de-indexed.
    Normals analytic, tangent = d/d(longitude) (w=+1), equirectangular UVs."""
    n = int(n)
    g = np.linspace(-1.0, 1.0, n + 1, dtype=np.float64)
    a, b = np.meshgrid(g, g, indexing="xy")  # (n+1, n+1)
    one = np.ones_like(a)
    faces = [
        np.stack([one, b, -a], -1), np.stack([-one, b, a], -1),
        np.stack([a, one, -b], -1), np.stack([a, -one, b], -1),
        np.stack([a, b, one], -1), np.stack([-a, b, -one], -1),
    ]
    out = []
    for f in faces:
        d = f / np.linalg.norm(f, axis=-1, keepdims=True)  # unit directions on the grid points
        p00, p10 = d[:-1, :-1], d[:-1, 1:]
        p01, p11 = d[1:, :-1], d[1:, 1:]
        tri = np.stack([p00, p10, p11, p00, p11, p01], axis=2)  # (n, n, 6, 3)
        out.append(tri.reshape(-1, 3))
    d = np.concatenate(out, 0)  # (72 n^2 / 6 * 6 ... = 36 n^2, 3)
    pos = d * radius + np.asarray(center, np.float64)
    lon = np.arctan2(d[:, 2], d[:, 0])
    u = lon / (2.0 * np.pi) + 0.5
    v_ = np.arccos(np.clip(d[:, 1], -1.0, 1.0)) / np.pi
    tang = np.stack([-d[:, 2], np.zeros(len(d)), d[:, 0]], -1)
    tl = np.linalg.norm(tang, axis=-1, keepdims=True)
    pole = tl[:, 0] < 1e-12
    tang = np.where(pole[:, None], np.array([1.0, 0.0, 0.0]), tang / np.where(tl == 0, 1.0, tl))
    verts = np.zeros((len(d), stride), np.float32)
    verts[:, 0:3] = pos
    verts[:, 3:6] = d
    verts[:, 6:9] = tang
    verts[:, 9] = 1.0
    verts[:, 10] = u
    verts[:, 11] = v_
    return verts


def cube_sphere(n: int, tex_size: int = 0, seed: int = SEED, stride: int = 12, name: str = "sphere_0") -> Scene:
    """I-2 (n=76, 2048^2 maps) / I-3 (n=289, 2048^2 maps): one textured cube-sphere mesh."""
    tex = procedural_textures(tex_size, seed) if tex_size else {}
    m = Mesh(name=name, vertices=cube_sphere_vertices(n, stride=stride), base_color=(1.0, 1.0, 1.0, 1.0), textures=tex)
    return Scene([m])


def sphere_grid(n_meshes_side: int = 4, n: int = 18, tex_size: int = 1024, spacing: float = 1.0,
                radius_frac: float = 0.2, seed: int = SEED, stride: int = 12) -> Scene:
    """I-4 (Sponza stand-in): side^3 meshes x cube-sphere(n), distinct materials, one grid cell each.
    Exercises the cumulative bbox (SceneManager.cpp:476-527) and per-mesh texture switching."""
    meshes = []
    k = 0
    for iz in range(n_meshes_side):
        for iy in range(n_meshes_side):
            for ix in range(n_meshes_side):
                c = (ix * spacing, iy * spacing, iz * spacing)
                v = cube_sphere_vertices(n, radius=radius_frac * spacing, center=c, stride=stride)
                tex = procedural_textures(tex_size, seed + k) if tex_size else {}
                col = (0.5 + 0.5 * ((k * 37) % 11) / 10.0, 0.5 + 0.5 * ((k * 17) % 7) / 6.0, 1.0, 1.0)
                meshes.append(Mesh(name=f"sphere_{k}", vertices=v, base_color=col, textures=tex))
                k += 1
    return Scene(meshes)


def sponza_standin(tex_size: int = 1024, seed: int = SEED, stride: int = 12) -> Scene:
    """I-4 as BASELINE config 4 needs it: 64 meshes x 3 888 triangles, 64 materials, and — what SURVEY 8(d) demands and the
    survey's own radius (0.2 s: 8 266 992 Gaussians at R = 1024) does not deliver — FEWER Gaussians than the reference's
    7 000 000 cap, so that parity is defined (beyond the cap the reference keeps an arrival-order-dependent subset).
    radius 0.12 s: 6 612 408 Gaussians at R = 1024 (oracle count; the first four meshes, whose cumulative bounding box is
    still one sphere wide in two axes, contribute 5.7 M of them whatever the radius)."""
    return sphere_grid(4, n=18, tex_size=tex_size, spacing=1.0, radius_frac=0.12, seed=seed, stride=stride)


def c5_scene(n: int = 1021, tex_size: int = 4096, count: int = 4, seed: int = SEED, cache: str | None = None) -> Scene:
    """I-5 at FULL size (BASELINE config 5): `count` cube-spheres of 12 n^2 triangles in a row (n = 1021: 50 037 168 triangles,
    7.2 GB of live vertex data), each with its own `tex_size`^2 maps.  sphere_row's layout, but ONE sphere is generated and
    shifted in fp32 (one sphere takes ~35 s of numpy; positions differ from sphere_row's in the last bit); `cache`: optional .npy path for repeated runs inside one session."""
    import os
    if cache and os.path.exists(cache):
        base = np.load(cache)
    else:
        base = cube_sphere_vertices(n, radius=1.0, center=(0.0, 0.0, 0.0), stride=12)
        if cache:
            np.save(cache, base)
    meshes = []
    for k in range(count):
        v = base.copy()
        v[:, 0] += np.float32(2.5 * k)
        tex = procedural_textures(tex_size, seed + k) if tex_size else {}
        meshes.append(Mesh(name=f"sphere_{k}", vertices=v, base_color=(1.0, 1.0, 1.0, 1.0), textures=tex))
    return Scene(meshes)


def sphere_row(count: int = 4, n: int = 361, tex_size: int = 4096, seed: int = SEED, stride: int = 12) -> Scene:
    """I-5 proxy (BASELINE config 5 at 1/8 of its triangles by default): `count` cube-spheres of 12 n^2 triangles in a row,
    each with its own maps; the cumulative bbox grows mesh by mesh, so later meshes cover fewer pixels each."""
    meshes = []
    for k in range(count):
        v = cube_sphere_vertices(n, radius=1.0, center=(2.5 * k, 0.0, 0.0), stride=stride)
        tex = procedural_textures(tex_size, seed + k) if tex_size else {}
        meshes.append(Mesh(name=f"sphere_{k}", vertices=v, base_color=(1.0, 1.0, 1.0, 1.0), textures=tex))
    return Scene(meshes)


def colocated_spheres(count: int, n: int, tex_size: int, seed: int = SEED, stride: int = 12) -> Scene:
    """Weak-scaling scene: `count` identical-geometry cube-spheres at the origin (so the cumulative
    bbox is the same for every mesh and every mesh yields the same number of Gaussians), one
    material per mesh sharing the same texel data."""
    base = cube_sphere_vertices(n, stride=stride)
    tex = procedural_textures(tex_size, seed) if tex_size else {}
    return Scene([Mesh(name=f"sphere_{k}", vertices=base, base_color=(1.0, 1.0, 1.0, 1.0), textures=tex)
                  for k in range(count)])


def random_soup(n_tri: int, seed: int = 1, extent: float = 1.0, tri_size: float = 0.2, stride: int = 12,
                textures: dict | None = None, name: str = "soup_0") -> Scene:
    """Random triangle soup with random normals/tangents/uvs: stresses every branch of the setup
    (all three projection axes, both windings, all longest-edge cases)."""
    rng = np.random.default_rng(seed)
    c = rng.uniform(0, extent, (n_tri, 1, 3))
    size = tri_size * rng.uniform(0.02, 1.0, (n_tri, 1, 1)) ** 2
    p = c + rng.uniform(-1, 1, (n_tri, 3, 3)) * size
    nrm = rng.normal(size=(n_tri, 3, 3))
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    tng = rng.normal(size=(n_tri, 3, 3))
    tng /= np.linalg.norm(tng, axis=-1, keepdims=True)
    v = np.zeros((n_tri, 3, stride), np.float32)
    v[..., 0:3] = p
    v[..., 3:6] = nrm
    v[..., 6:9] = tng
    v[..., 9] = rng.choice([-1.0, 1.0], (n_tri, 3))
    v[..., 10:12] = rng.uniform(-0.5, 2.5, (n_tri, 3, 2))
    m = Mesh(name=name, vertices=v.reshape(-1, stride), base_color=(0.8, 0.6, 0.4, 0.9), textures=dict(textures or {}))
    return Scene([m])


# ---- heterogeneous scene (round 5): what a real asset looks like to the converter ----------------------------------

def _surface_mesh(P, dPds, dPdt, uv, stride: int = 12) -> np.ndarray:
    """De-indexed triangles of a parametric grid: P, dPds, dPdt (nv+1, nu+1, 3) float64, uv (nv+1, nu+1, 2);
    normal = normalize(dPds x dPdt), tangent = normalize(dPds) (w = +1).  Two triangles per cell: (00, 10, 11), (00, 11, 01)."""
    n = np.cross(dPds, dPdt)
    n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-30)
    t = dPds / np.maximum(np.linalg.norm(dPds, axis=-1, keepdims=True), 1e-30)
    att = np.concatenate([P, n, t, np.ones(P.shape[:2] + (1,)), uv], axis=-1)   # (nv+1, nu+1, 12)
    a00, a10, a01, a11 = att[:-1, :-1], att[:-1, 1:], att[1:, :-1], att[1:, 1:]
    tri = np.stack([a00, a10, a11, a00, a11, a01], axis=2)                        # (nv, nu, 6, 12)
    v = np.zeros((tri.shape[0] * tri.shape[1] * 6, stride), np.float32)
    v[:, :12] = tri.reshape(-1, 12)
    return v


def patch_vertices(nu: int, nv: int, origin, U, V, amp: float = 0.0, waves=(3.0, 2.0), uv_tile: float = 1.0,
                   stride: int = 12) -> np.ndarray:
    """nu x nv cells on the parallelogram origin + s U + t V (s, t in [0, 1]), displaced along its normal by
    amp sin(2 pi w0 s) sin(2 pi w1 t): planes (amp = 0), cloth."""
    origin, U, V = (np.asarray(x, np.float64) for x in (origin, U, V))
    N = np.cross(U, V)
    N /= np.linalg.norm(N)
    s, t = np.meshgrid(np.linspace(0.0, 1.0, nu + 1), np.linspace(0.0, 1.0, nv + 1), indexing="xy")
    w0, w1 = (2.0 * np.pi * w for w in waves)
    d = amp * np.sin(w0 * s) * np.sin(w1 * t)
    P = origin + s[..., None] * U + t[..., None] * V + d[..., None] * N
    dPds = U + (amp * w0 * np.cos(w0 * s) * np.sin(w1 * t))[..., None] * N
    dPdt = V + (amp * w1 * np.sin(w0 * s) * np.cos(w1 * t))[..., None] * N
    uv = np.stack([s * uv_tile, t * uv_tile], -1)
    return _surface_mesh(P, dPds, dPdt, uv, stride)


def cylinder_vertices(seg: int, rings: int, base, radius: float, height: float, uv_tile: float = 1.0, stride: int = 12) -> np.ndarray:
    """Open cylinder along +y: seg x rings cells."""
    base = np.asarray(base, np.float64)
    th, t = np.meshgrid(np.linspace(0.0, 2.0 * np.pi, seg + 1), np.linspace(0.0, 1.0, rings + 1), indexing="xy")
    c, s_ = np.cos(th), np.sin(th)
    P = base + np.stack([radius * c, height * t, radius * s_], -1)
    dPds = np.stack([-radius * s_, np.zeros_like(th), radius * c], -1)
    dPdt = np.broadcast_to(np.array([0.0, height, 0.0]), P.shape).copy()
    uv = np.stack([th / (2.0 * np.pi) * uv_tile, t * uv_tile], -1)
    return _surface_mesh(P, dPdt, dPds, uv, stride)   # (t, theta) order: outward normals


def sponza_like(seed: int = SEED, combo_only: bool = False, tex_scale: float = 1.0, stride: int = 12) -> Scene:
    """A Sponza-shaped workload (BASELINE config 4 names the real file, which no box here has): 64 meshes, ~267 k triangles whose
    pixel areas at R = 1024 spread over six decades inside ONE scene — a 2-triangle floor across the whole bounding box (half a
    million fragments per triangle), three 2-triangle walls, sixteen columns (~50-150 px per triangle), ten mid-size patches,
    twenty-four props, eight sheets of dense cloth (a few pixels per triangle) and two of sub-pixel foliage (a fragment every
    dozen triangles) — in shuffled mesh order behind the floor, materials with maps of 256^2 ... 2048^2, some with an albedo
    map only, some with none (`combo_only`: every textured material carries all three maps, so the lean kernel is allowed).
    The floor spans [0,1]^2 in x/z, so every cumulative bounding box has range 1 in each projection plane and a triangle's pixel
    area is its projected world area times R^2.  ~3.5 M Gaussians at R = 1024 (under the reference's 7 M cap)."""
    rng = np.random.default_rng(seed & 0xFFFFFFFF)
    axes = np.eye(3)

    def oriented(k):   # two in-plane unit vectors whose normal is axis k
        return axes[(k + 1) % 3], axes[(k + 2) % 3]

    meshes = []   # (name, vertices, tex size or 0, maps)
    meshes.append(("floor", patch_vertices(1, 1, (0, 0, 1), (1, 0, 0), (0, 0, -1), uv_tile=8.0, stride=stride), 2048, 3))
    rest = []
    for i in range(3):
        k = i % 3
        U, V = oriented(k)
        o = rng.uniform(0.05, 0.55, 3)
        o[k] = rng.uniform(0.1, 0.5)
        rest.append((f"wall_{i}", patch_vertices(1, 1, o, 0.4 * U, 0.4 * V, uv_tile=4.0, stride=stride), 1024, 3))
    for i in range(16):
        r, hgt = rng.uniform(0.015, 0.04), rng.uniform(0.3, 0.6)
        b = (rng.uniform(0.05, 0.95), 0.0, rng.uniform(0.05, 0.95))
        rest.append((f"column_{i}", cylinder_vertices(32, 16, b, r, hgt, stride=stride), 512 if i % 4 else 0, 3))
    for i in range(10):
        size = rng.uniform(0.1, 0.4)
        U, V = oriented(int(rng.integers(0, 3)))
        tilt = rng.normal(scale=0.15, size=3)
        o = rng.uniform(0.0, 1.0 - size, 3) * (1.0, 0.5, 1.0)
        rest.append((f"panel_{i}", patch_vertices(30, 30, o, size * (U + tilt * 0.5), size * (V - tilt * 0.5), amp=0.01 * size, stride=stride),
                     (1024, 512)[i % 2], 3 if i % 3 else 1))
    for i in range(24):
        size = float(np.exp(rng.uniform(np.log(0.02), np.log(0.3))))
        U, V = oriented(int(rng.integers(0, 3)))
        o = rng.uniform(0.0, 1.0 - size, 3) * (1.0, 0.5, 1.0)
        rest.append((f"prop_{i}", patch_vertices(12, 12, o, size * U, size * V, amp=0.05 * size, waves=(1.0, 1.0), stride=stride),
                     (256, 512, 256, 0)[i % 4], 3 if i % 5 else 1))
    for i in range(8):
        size = rng.uniform(0.1, 0.4)
        U, V = oriented(int(rng.integers(0, 3)))
        o = rng.uniform(0.0, 1.0 - size, 3) * (1.0, 0.5, 1.0)
        rest.append((f"cloth_{i}", patch_vertices(100, 100, o, size * U, size * V, amp=0.02, waves=(5.0, 3.0), uv_tile=2.0, stride=stride),
                     (2048, 1024, 1024, 512)[i % 4], 3))
    for i in range(2):
        U, V = oriented(i)
        o = rng.uniform(0.1, 0.8, 3) * (1.0, 0.5, 1.0)
        rest.append((f"foliage_{i}", patch_vertices(128, 128, o, 0.05 * U, 0.05 * V, amp=0.004, waves=(9.0, 7.0), stride=stride), 512, 3))
    order = rng.permutation(len(rest))
    meshes += [rest[j] for j in order]
    out = []
    tex_cache = {}
    for k, (name, v, tsize, maps) in enumerate(meshes):
        tex = {}
        if tsize:
            ts = max(16, int(tsize * tex_scale))
            if (ts, k % 7) not in tex_cache:
                tex_cache[(ts, k % 7)] = procedural_textures(ts, seed + (k % 7))
            full = tex_cache[(ts, k % 7)]
            if maps == 3 or combo_only:
                tex = dict(full)
            else:
                tex = {"baseColorTexture": full["baseColorTexture"]}
        col = (0.6 + 0.4 * ((k * 37) % 11) / 10.0, 0.6 + 0.4 * ((k * 17) % 7) / 6.0, 0.9, 1.0)
        out.append(Mesh(name=name, vertices=v, base_color=col, textures=tex))
    return Scene(out)
