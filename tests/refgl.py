"""Helper (test infrastructure): run the reference's conversion path on Mesa llvmpipe (oracle/_ref/ref_gl_check) and compare
with the oracle.  GL's fixed-function stages are implementation-defined in their last bits, so this MEASURES differences."""
import json
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_gl_check")


def available():
    return os.path.isfile(EXE) and os.access(EXE, os.X_OK)


def run(scene, R, float_sampler=False, want_mips=False):
    """-> dict(info, counter, cap, records (arrival order), coverage (n,4) u32 [mesh, tri, x, y], mips or None); None if the
    GL context cannot be created on this machine."""
    from mesh2splat_amd import gltf_io
    env = dict(os.environ)
    if float_sampler:      # llvmpipe's fp32 texture path instead of its 8-bit fixed-point one
        env["GALLIVM_PERF"] = "no_aos_sampling,no_quad_lod"
    with tempfile.TemporaryDirectory() as d:
        glb = os.path.join(d, "scene.glb")
        gltf_io.write_glb(scene, glb, indexed=False)
        args = [EXE, glb, str(int(R)), os.path.join(d, "rec.bin"), os.path.join(d, "cov.bin")]
        if want_mips:
            args.append(os.path.join(d, "mips.bin"))
        r = subprocess.run(args, capture_output=True, text=True, env=env, timeout=900)
        if r.returncode == 3:
            return None
        assert r.returncode == 0, r.stderr[-2000:]
        info = json.loads(r.stdout.strip().splitlines()[-1])
        raw = open(os.path.join(d, "rec.bin"), "rb").read()
        counter, cap = (int(x) for x in np.frombuffer(raw[:8], np.uint32))
        rec = np.frombuffer(raw[16:], np.float32).reshape(-1, 24).copy()
        cov = np.frombuffer(open(os.path.join(d, "cov.bin"), "rb").read()[4:], np.uint32).reshape(-1, 4).copy()
        mips = None
        if want_mips and os.path.exists(os.path.join(d, "mips.bin")):
            mraw = open(os.path.join(d, "mips.bin"), "rb").read()
            w, h, n = (int(x) for x in np.frombuffer(mraw[:12], np.uint32))
            mips, off = [], 12
            for l in range(n):
                lw, lh = max(1, w >> l), max(1, h >> l)
                mips.append(np.frombuffer(mraw[off:off + lw * lh * 4], np.uint8).reshape(lh, lw, 4).copy())
                off += lw * lh * 4
    return {"info": info, "counter": counter, "cap": cap, "records": rec, "coverage": cov, "mips": mips}


def coverage_keys(scene, cov):
    """(mesh, triangle, x, y) -> the oracle's key: global triangle << 24 | y << 12 | x"""
    mf = np.cumsum([0] + [m.n_triangles for m in scene.meshes]).astype(np.uint64)
    return ((mf[cov[:, 0]] + cov[:, 1].astype(np.uint64)) << np.uint64(24)) | (cov[:, 3].astype(np.uint64) << np.uint64(12)) | cov[:, 2].astype(np.uint64)


FIELDS = (("position", slice(0, 3)), ("color", slice(4, 8)), ("scale", slice(8, 10)), ("normal", slice(12, 15)),
          ("rotation", slice(16, 20)), ("pbr", slice(20, 22)))


def compare(scene, R, oracle, **kw):
    """One scene: count, coverage set difference, per-field deviations of matched records (matched by nearest position)."""
    from scipy.spatial import cKDTree
    g = run(scene, R, **kw)
    if g is None:
        return None
    total, orec, keys = oracle.convert(scene, R, cap=0, want_keys=True)
    gk = set(coverage_keys(scene, g["coverage"]).tolist())
    ok = set(keys.tolist())
    out = {"R": int(R), "triangles": int(scene.n_triangles), "gl_counter": g["counter"], "oracle_counter": int(total),
           "count_delta": g["counter"] - int(total), "coverage_fragments_gl": len(g["coverage"]),
           "pixels_only_gl": len(gk - ok), "pixels_only_oracle": len(ok - gk), "gl_error": g["info"]["gl_error"],
           "execute_ms": g["info"]["execute_ms"], "gl": g["info"]["gl_version"] + " / " + g["info"]["gl_renderer"]}
    rec = g["records"][: min(len(g["records"]), len(orec))]
    if len(rec) and len(orec):
        dist, idx = cKDTree(orec[:, 0:3].astype(np.float64)).query(rec[:, 0:3].astype(np.float64))
        o = orec[idx]
        out["match_max_position_distance"] = float(dist.max())
        for name, sl in FIELDS:
            d = np.abs(rec[:, sl].astype(np.float64) - o[:, sl].astype(np.float64))
            out[name] = {"max_abs": float(d.max()), "mean_abs": float(d.mean()),
                         "bit_identical": float((rec[:, sl].view(np.uint32) == o[:, sl].view(np.uint32)).mean())}
    return out
