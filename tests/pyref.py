"""Second, independent restatement of the conversion pass in pure Python (numpy float32 scalars),
used ONLY to pin the C oracle on small cases: it transcribes the shader text line by line
(converterGS.glsl:326-443, converterFS.glsl:44-104) and tests every pixel centre with the
integer edge functions.  Far too slow for anything but a handful of triangles at R <= 64.
"""
import math

import numpy as np

f32 = np.float32


def _len(v):
    return f32(np.sqrt(f32(f32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])))


def _normalize(v):
    inv = f32(1.0) / _len(v)
    return np.array([v[0] * inv, v[1] * inv, v[2] * inv], f32)


def _cross(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]),
                     f32(a[0] * b[1]) - f32(a[1] * b[0])], f32)


def quat_cast(m):
    """m[c][r]; returns (x, y, z, w).  converterGS.glsl:131-183."""
    fx = f32(f32(m[0][0] - m[1][1]) - m[2][2])
    fy = f32(f32(m[1][1] - m[0][0]) - m[2][2])
    fz = f32(f32(m[2][2] - m[0][0]) - m[1][1])
    fw = f32(f32(m[0][0] + m[1][1]) + m[2][2])
    bi, fb = 0, fw
    if fx > fb:
        fb, bi = fx, 1
    if fy > fb:
        fb, bi = fy, 2
    if fz > fb:
        fb, bi = fz, 3
    bv = f32(np.sqrt(f32(fb + f32(1.0)))) * f32(0.5)
    mult = f32(0.25) / bv
    if bi == 0:
        w, x, y, z = bv, (m[1][2] - m[2][1]) * mult, (m[2][0] - m[0][2]) * mult, (m[0][1] - m[1][0]) * mult
    elif bi == 1:
        w, x, y, z = (m[1][2] - m[2][1]) * mult, bv, (m[0][1] + m[1][0]) * mult, (m[2][0] + m[0][2]) * mult
    elif bi == 2:
        w, x, y, z = (m[2][0] - m[0][2]) * mult, (m[0][1] + m[1][0]) * mult, bv, (m[1][2] + m[2][1]) * mult
    else:
        w, x, y, z = (m[0][1] - m[1][0]) * mult, (m[2][0] + m[0][2]) * mult, (m[1][2] + m[2][1]) * mult, bv
    return f32(x), f32(y), f32(z), f32(w)


def triangle_setup(p, bmin, bmax, R):
    """p: (3,3) float32.  Returns dict with axis pair, ortho uvs, snapped ints, scale, rot (w,x,y,z)."""
    p = np.asarray(p, f32)
    bmin = np.asarray(bmin, f32)
    bmax = np.asarray(bmax, f32)
    with np.errstate(all="ignore"):
        e1, e2, e3 = p[1] - p[0], p[2] - p[0], p[2] - p[1]
        l1, l2, l3 = _len(e1), _len(e2), _len(e3)
        if l2 > l1 and l2 > l3:
            e1, e2 = e2, e1
        elif l3 > l1 and l3 > l2:
            e1, e3 = e3, e1
        xa = _normalize(e1)
        nrm = _normalize(_cross(xa, e2))
        ax, ay, az = abs(nrm[0]), abs(nrm[1]), abs(nrm[2])
        if ax > ay and ax > az:
            A, B = 1, 2
        elif ay > az:
            A, B = 0, 2
        else:
            A, B = 0, 1
        rng = f32(max(bmax[A] - bmin[A], bmax[B] - bmin[B]))
        ou = [f32(f32(p[i][A] - bmin[A]) / rng) for i in range(3)]   # true division, as the shader writes it
        ov = [f32(f32(p[i][B] - bmin[B]) / rng) for i in range(3)]
        ya = _normalize(_cross(nrm, xa))
        q = quat_cast([xa, ya, nrm])
        rot = (q[3], q[0], q[1], q[2])
        U00, U10 = f32(ou[1] - ou[0]), f32(ou[2] - ou[0])
        U01, U11 = f32(ov[1] - ov[0]), f32(ov[2] - ov[0])
        det = f32(f32(U00 * U11) - f32(U01 * U10))
        I00 = I10 = I01 = I11 = f32(0)
        if det != 0:
            invdet = f32(1.0) / det
            I00, I10, I01, I11 = f32(U11 * invdet), f32(-U10 * invdet), f32(-U01 * invdet), f32(U00 * invdet)
        V0, V1 = p[1] - p[0], p[2] - p[0]
        Ju = np.array([f32(f32(V0[k] * I00) + f32(V1[k] * I01)) for k in range(3)], f32)
        Jv = np.array([f32(f32(V0[k] * I10) + f32(V1[k] * I11)) for k in range(3)], f32)
        half = f32(R) * f32(0.5)
        X, Y, ok = [], [], True
        for i in range(3):
            xw = f32(f32(half * f32(f32(ou[i] * f32(2)) - f32(1))) + half)
            yw = f32(f32(half * f32(f32(ov[i] * f32(2)) - f32(1))) + half)
            if not (abs(xw) < 16384 and abs(yw) < 16384):
                ok = False
                break
            X.append(int(np.rint(f32(xw * f32(256)))))
            Y.append(int(np.rint(f32(yw * f32(256)))))
    return dict(A=A, B=B, ou=ou, ov=ov, X=X, Y=Y, ok=ok, scale=(_len(Ju), _len(Jv)), rot=rot)


def covered_pixels(X, Y, R):
    """Pinned rasteriser: pixel centre rule, top-left ownership, both windings.  Returns [(y,x,E1,E2,area2)]."""
    area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0])
    if area2 == 0:
        return []
    sgn = -1 if area2 < 0 else 1
    edges = []
    for i in range(3):
        ia, ib = (i + 1) % 3, (i + 2) % 3
        dy, dx = Y[ib] - Y[ia], X[ib] - X[ia]
        a, b, c = -dy * sgn, dx * sgn, (dy * X[ia] - dx * Y[ia]) * sgn
        edges.append((a, b, c, a > 0 or (a == 0 and b > 0)))
    out = []
    for y in range(R):
        for x in range(R):
            Px, Py = 256 * x + 128, 256 * y + 128
            E = [a * Px + b * Py + c for a, b, c, _ in edges]
            if all(e > 0 or (e == 0 and own) for e, (_, _, _, own) in zip(E, edges)):
                out.append((y, x, E[1], E[2], abs(area2)))
    return out


def convert_untextured(verts, bmin, bmax, color, R):
    """verts (3T, >=12).  Returns list of 24-float records in canonical order."""
    verts = np.asarray(verts, f32)
    recs = []
    for t in range(verts.shape[0] // 3):
        v = verts[3 * t:3 * t + 3]
        s = triangle_setup(v[:, 0:3], bmin, bmax, R)
        if not s["ok"]:
            continue
        for (y, x, E1, E2, area2) in covered_pixels(s["X"], s["Y"], R):
            inva = f32(1.0) / f32(area2)
            l1, l2 = f32(f32(E1) * inva), f32(f32(E2) * inva)
            f = [f32(f32(v[0][k] + f32(l1 * f32(v[1][k] - v[0][k]))) + f32(l2 * f32(v[2][k] - v[0][k]))) for k in range(12)]
            rec = [f[0], f[1], f[2], 1.0] + [f32(1.0) * f32(color[k]) for k in range(4)] + \
                  [s["scale"][0], s["scale"][1], f32(1e-7), 0.0] + [f[3], f[4], f[5], 0.0] + list(s["rot"]) + \
                  [f32(0.1), f32(0.5), 0.0, 1.0]
            recs.append(rec)
    return np.array(recs, f32).reshape(-1, 24)


# ---- textures -------------------------------------------------------------------------------------
def build_mips(tex):
    levels = [np.asarray(tex, np.uint8)]
    h, w = tex.shape[:2]
    while max(w, h) > 1 and len(levels) < 5:
        src = levels[-1].astype(np.uint32)
        sh, sw = src.shape[:2]
        w, h = max(1, w // 2), max(1, h // 2)
        dst = np.zeros((h, w, 4), np.uint8)
        for y in range(h):
            for x in range(w):
                y0, y1 = min(2 * y, sh - 1), min(2 * y + 1, sh - 1)
                x0, x1 = min(2 * x, sw - 1), min(2 * x + 1, sw - 1)
                dst[y, x] = (src[y0, x0] + src[y0, x1] + src[y1, x0] + src[y1, x1] + 2) >> 2
        levels.append(dst)
    return levels


def _bilinear(img, uf, vf):
    H, W = img.shape[:2]
    up, vp = f32(f32(uf * f32(W)) - f32(0.5)), f32(f32(vf * f32(H)) - f32(0.5))
    fi, fj = math.floor(up), math.floor(vp)
    a, b = f32(up - f32(fi)), f32(vp - f32(fj))
    i0, j0, i1, j1 = fi % W, fj % H, (fi + 1) % W, (fj + 1) % H
    w00, w10 = f32(f32(1 - a) * f32(1 - b)), f32(a * f32(1 - b))
    w01, w11 = f32(f32(1 - a) * b), f32(a * b)
    t = img.astype(f32)
    return f32(f32(f32(w00 * t[j0, i0] + w10 * t[j0, i1]) + w01 * t[j1, i0]) + w11 * t[j1, i1])


def sample(levels, u, v, lam):
    u, v = f32(u), f32(v)
    uf, vf = f32(u - f32(math.floor(u))), f32(v - f32(math.floor(v)))
    k = f32(0.003921568859368563)
    q = len(levels) - 1
    if not lam > 0:
        return _bilinear(levels[0], uf, vf) * k
    if lam >= q:
        return _bilinear(levels[q], uf, vf) * k
    d = math.floor(lam)
    f = f32(f32(lam) - f32(d))
    return f32(f32(f32(1) - f) * _bilinear(levels[d], uf, vf) + f * _bilinear(levels[d + 1], uf, vf)) * k
