"""CPU check of the sparse kernel's tier-1 test (m2s_sparse.hip: tier1_empty): a float32 numpy transcription run against the
oracle's exact per-triangle fragment counts.  A triangle the test drops must have count 0; reports the survivor rate too.
usage: python tools/tier1_check.py            (test infrastructure: uses the oracle)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mesh2splat_amd import synth          # noqa: E402
from oracle import oracle                  # noqa: E402

f32 = np.float32


def tier1_empty(v, bmin, bmax, R):
    """v: (T,3,3) float32 positions.  Returns the boolean 'certainly empty' per triangle."""
    v = v.astype(f32)
    p0, p1, p2 = v[:, 0], v[:, 1], v[:, 2]
    e1, e2, e3 = p1 - p0, p2 - p0, p2 - p1
    c = np.cross(e1, e2).astype(f32)
    a = np.abs(c)
    mx = a.max(1)
    md = np.median(a, axis=1).astype(f32)
    lm = np.maximum((e1 * e1).sum(1), np.maximum((e2 * e2).sum(1), (e3 * e3).sum(1))).astype(f32)
    cc = (c * c).sum(1).astype(f32)
    with np.errstate(all="ignore"):
        clear = (md < mx * f32(0.984375)) & (cc >= f32(1e-6) * lm * lm) & (cc > f32(1e-30)) & (lm < f32(1e18))
    first = (a[:, 0] > a[:, 1]) & (a[:, 0] > a[:, 2])
    second = ~first & (a[:, 1] > a[:, 2])
    zb = first | second
    ext = (bmax - bmin).astype(f32)
    ryz, rxz, rxy = max(ext[1], ext[2]), max(ext[0], ext[2]), max(ext[0], ext[1])
    if not (min(ryz, rxz, rxy) > 1e-30 and max(ryz, rxz, rxy) < 1e30):
        return np.zeros(len(v), bool)
    Rf = f32(R)
    syz, sxz, sxy = Rf * (f32(1) / ryz), Rf * (f32(1) / rxz), Rf * (f32(1) / rxy)
    s = np.where(first, syz, np.where(second, sxz, sxy)).astype(f32)
    bA = np.where(first, bmin[1], bmin[0]).astype(f32)
    bB = np.where(zb, bmin[2], bmin[1]).astype(f32)
    pa = np.where(first[:, None], v[:, :, 1], v[:, :, 0])
    pb = np.where(zb[:, None], v[:, :, 2], v[:, :, 1])
    with np.errstate(all="ignore"):
        x = ((pa - bA[:, None]) * s[:, None]).astype(f32)
        y = ((pb - bB[:, None]) * s[:, None]).astype(f32)
        mg = f32(3.0 / 256.0)
        ix0 = np.maximum(np.ceil(x.min(1) - f32(0.5) - mg), 0)
        ix1 = np.minimum(np.floor(x.max(1) - f32(0.5) + mg), Rf - 1)
        iy0 = np.maximum(np.ceil(y.min(1) - f32(0.5) - mg), 0)
        iy1 = np.minimum(np.floor(y.max(1) - f32(0.5) + mg), Rf - 1)
        empty = (ix0 > ix1) | (iy0 > iy1)
        wx, wy = ix1 - ix0, iy1 - iy0                 # candidate centres: (ix0 + 0.5 + kx, iy0 + 0.5 + ky), kx in [0, wx], ky in [0, wy]
        pcx, pcy = ix0 + f32(0.5), iy0 + f32(0.5)
        out_pos = np.zeros(len(v), bool)
        out_neg = np.zeros(len(v), bool)
        for i in range(3):
            j = (i + 1) % 3
            dx, dy = x[:, j] - x[:, i], y[:, j] - y[:, i]
            qx, qy = pcx - x[:, i], pcy - y[:, i]
            E = dx * qy - dy * qx
            sx_, sy_ = dx * wy, -dy * wx
            tol = f32(0.012) * (((np.abs(qy) + wy) + (np.abs(qx) + wx)) + (np.abs(dx) + np.abs(dy))) + f32(2e-4)
            out_pos |= (E + np.maximum(sx_, 0) + np.maximum(sy_, 0)) < -tol
            out_neg |= (E + np.minimum(sx_, 0) + np.minimum(sy_, 0)) > tol
        empty |= out_pos & out_neg & ~empty
    return clear & empty


def check(scene, R, label):
    m = scene.meshes[0]
    v = m.vertices.reshape(-1, 3, m.vertices.shape[1])[:, :, :3]
    cnt = np.zeros(len(v), np.uint32)
    arr, keep = oracle._c_meshes(scene)
    tot = oracle.lib().orc_count_per_triangle(arr, 1, R, cnt.ctypes.data)
    drop = tier1_empty(v, np.asarray(m.bbox_min, f32), np.asarray(m.bbox_max, f32), R)
    bad = int((drop & (cnt > 0)).sum())
    print(f"{label:38s} R={R:5d} T={len(v):8d} N={tot:9d} emit={np.mean(cnt > 0):.3f} survive={1 - drop.mean():.3f} WRONGLY DROPPED={bad}")
    return bad


if __name__ == "__main__":
    bad = 0
    for n, R in ((255, 512), (128, 256), (289, 1024), (400, 300), (511, 4096), (300, 97)):
        bad += check(synth.cube_sphere(n), R, f"cube_sphere({n})")
    for seed in range(6):
        for R in (64, 333, 1024, 4096):
            bad += check(synth.random_soup(200_000, seed=seed, tri_size=0.004 * (seed + 1)), R, f"random_soup(seed={seed})")
    # slivers and near-degenerate triangles: stretch a soup along one axis
    for seed in range(3):
        sc = synth.random_soup(200_000, seed=10 + seed, tri_size=0.01)
        vv = sc.meshes[0].vertices
        vv[:, seed % 3] *= np.float32(1e-3)
        sc.meshes[0].bbox_min = sc.meshes[0].bbox_max = None
        sc = type(sc)(sc.meshes)          # recompute the bounding box
        for R in (256, 2048):
            bad += check(sc, R, f"flattened soup (axis {seed % 3})")
    print("OK" if bad == 0 else f"FAILED: {bad}")
    sys.exit(1 if bad else 0)
