"""Per-wave timeline of k_fused3 (the lean team kernel) on a -DM2S_TIMELINE build of the library: when workgroups start, how long the
triangle phase, the wait for the base / the other waves' entries and the strips take, how many waves are alive over time.
    M2S_LIB_PATH=mesh2splat_amd/_build_tl/libm2s_hip.so python tools/timeline_fused3.py [c3|c2] [out.json]"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
import bench  # noqa: E402
from mesh2splat_amd import synth, _lib  # noqa: E402
from mesh2splat_amd.converter import Converter  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "c3"
out = sys.argv[2] if len(sys.argv) > 2 else None
n, tex, R = bench.WORKLOADS[name]
scene = synth.colocated_spheres(1, n, tex)
L = _lib.load()
c = Converter(0)
c.set_resolution_hint(R)
c.upload_scene(scene)
c.set_pipeline(os.environ.get("TL_PIPELINE", "lean"))
for _ in range(4):
    tot = c.convert(R)
L.m2s_debug_timeline_f3.restype = C.c_int
L.m2s_debug_timeline_f3.argtypes = [C.c_void_p, C.c_size_t]
L.m2s_debug_timeline_f3_clear()
c.set_profiling(True)
tot = c.convert(R)
kms = c.last_kernel_ms()
t = np.zeros((16384, 4, 8), np.uint64)
assert L.m2s_debug_timeline_f3(t.ctypes.data, t.nbytes) == 0
used = t[:, :, 0].any(axis=1)
nwg = int(np.nonzero(used)[0].max()) + 1
t = t[:nwg].astype(np.int64)
ok = t[:, 0, 0] > 0
T0 = t[ok][:, :, 0].min()
ts = (t[:, :, :6] - T0) * 10     # ns
ts[~ok] = 0
strips = t[:, :, 6]
print(json.dumps({"workload": name, "R": R, "gaussians": int(tot), "pipeline": str(c.last_pipeline), "kernel_ms": kms, "workgroups": nwg, "with_work": int(ok.sum())}))
st, en = ts[ok][:, :, 0], ts[ok][:, :, 5]
print(f"span {en.max()} ns; workgroup starts p50 {np.percentile(st, 50):.0f} p90 {np.percentile(st, 90):.0f} max {st.max()}")
d = np.diff(ts[ok], axis=2)
lab = ["triangle phase", "wait counts/base + expand", "wait all counts", "strips", "epilogue"]
print("mean ns per phase over waves:", {lab[i]: int(d[:, :, i].mean()) for i in range(5)}, "| wave life mean", int((en - st).mean()), "p90", int(np.percentile(en - st, 90)), "max", int((en - st).max()))
print("last wave (look-back) phase 1:", {"mean": int(d[:, 3, 1].mean()), "p90": int(np.percentile(d[:, 3, 1], 90)), "max": int(d[:, 3, 1].max())})
print("strips per wave mean", float(strips[ok].mean()), "max", int(strips[ok].max()), "| ns per strip", float(d[:, :, 3].sum() / max(strips[ok].sum(), 1)))
grid = np.arange(0, en.max() + 1, 5000)
alive = [int(((st <= g) & (en > g)).sum()) for g in grid]
instrips = [int(((ts[ok][:, :, 3] <= g) & (ts[ok][:, :, 4] > g)).sum()) for g in grid]
intri = [int(((ts[ok][:, :, 0] <= g) & (ts[ok][:, :, 1] > g)).sum()) for g in grid]
inwait = [int(((ts[ok][:, :, 1] <= g) & (ts[ok][:, :, 3] > g)).sum()) for g in grid]
print("every 5 us: waves alive", alive)
print("            in the triangle phase", intri)
print("            waiting (counts, base, expansion)", inwait)
print("            in strips", instrips)
if out:
    json.dump({"t": ts.tolist(), "strips": strips.tolist(), "lb": t[:, :, 7].tolist()}, open(out, "w"))
