"""Where the multi-pass pipeline's time goes, per workgroup: a -DM2S_TIMELINE build of the library (make OUT=../_build_tl
EXTRA=-DM2S_TIMELINE; never the shipping one) leaves 100 MHz timestamps of every wave of k_count_scan (eight phases) and k_emit2
(start, end, batches, kind) behind; this prints the kernels' spans, the slowest workgroups with their phases and meshes.
    M2S_LIB_PATH=mesh2splat_amd/_build_tl/libm2s_hip.so python tools/timeline_probe.py [hetero|c4|mid] [out.json]"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from mesh2splat_amd import synth, _lib  # noqa: E402
from mesh2splat_amd.converter import Converter  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "hetero"
out = sys.argv[2] if len(sys.argv) > 2 else None
import bench  # noqa: E402
n, tex, R = bench.WORKLOADS[name]
R = int(os.environ.get("TL_R", R))
scene = synth.sponza_like() if n == "sponza_like" else synth.sponza_standin(tex) if n == "grid" else synth.colocated_spheres(1, n, tex)
L = _lib.load()
c = Converter(0)
c.set_resolution_hint(R)
c.upload_scene(scene)
c.set_pipeline(os.environ.get("TL_PIPELINE", "multipass"))
for _ in range(4):
    tot = c.convert(R)
L.m2s_debug_timeline_clear.restype = C.c_int
L.m2s_debug_timeline.restype = C.c_int
L.m2s_debug_timeline.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.m2s_debug_timeline_clear()
c.set_profiling(True)
tot = c.convert(R)
kms = c.last_kernel_ms()
T = c.num_triangles
nb = (T + 255) // 256
tc = np.zeros((8192, 4, 8), np.uint64)
te = np.zeros((32768, 4, 4), np.uint64)
assert L.m2s_debug_timeline(tc.ctypes.data, tc.nbytes, te.ctypes.data, te.nbytes) == 0
tc = tc[:nb].astype(np.int64)
names, first = [], []
t0 = 0
for m in scene.meshes:
    names.append(m.name); first.append(t0); t0 += m.vertices.shape[0] // 3
first = np.array(first)


def mesh_of_block(b):
    return names[int(np.searchsorted(first, b * 256, side="right") - 1)]


T0 = tc[:, :, 0].min()
tc = (tc - T0) * 10           # ns
print(json.dumps({"workload": name, "R": R, "gaussians": int(tot), "pipeline": str(c.last_pipeline), "kernel_ms": kms, "blocks": nb}))
span = tc[:, :, 7].max()
print(f"k_count_scan: span {span} ns over {nb} blocks; starts: median {np.median(tc[:, 0, 0]):.0f} max {tc[:, :, 0].max()} ns")
ph = ["load+GS+raster_setup", "lane row loops", "tall loop", "scan+sync1+publish", "TriSetup", "lookback+sync2", "off/start"]
w_end = tc[:, :, 7].max(axis=1)
w_pub = tc[:, :, 4].max(axis=1)
print("  percentiles of block end      (ns):", [int(np.percentile(w_end, q)) for q in (10, 50, 90, 99, 100)])
print("  percentiles of block publish  (ns):", [int(np.percentile(w_pub, q)) for q in (10, 50, 90, 99, 100)])
dur = np.diff(tc, axis=2)     # phases 0->1 ... 6->7
life = tc[:, :, 7] - tc[:, :, 0]
print("  mean ns per phase over waves:", {ph[i]: int(dur[:, :, i].mean()) for i in range(7)})
print("  mean wave life", int(life.mean()), "max", int(life.max()))
order = np.argsort(-w_pub)[:12]
print("  latest publishers:")
for b in order:
    w = int(np.argmax(tc[b, :, 4]))
    print(f"    block {int(b):5d} {mesh_of_block(int(b)):12s} start {tc[b, w, 0]:6d} publish {tc[b, w, 4]:6d} end {w_end[b]:6d} | wave {w}: " +
          " ".join(f"{ph[i].split()[0]}={int(dur[b, w, i])}" for i in range(7)))
# per mesh kind: mean time to publish - start
kinds = {}
for b in range(nb):
    k = mesh_of_block(b).split("_")[0]
    kinds.setdefault(k, []).append((tc[b, :, 4].max() - tc[b, :, 0].min(), dur[b, :, 0].max(), dur[b, :, 1].max(), dur[b, :, 2].max(), dur[b, :, 5].max(), dur[b, :, 6].max()))
print("  per mesh kind (block means, ns): start->publish | setup | lane loops | tall | TriSetup | lookback+sync2")
for k, v in kinds.items():
    a = np.array(v, float)
    print(f"    {k:8s} n={len(v):4d} " + " ".join(f"{x:7.0f}" for x in a.mean(axis=0)) + f"   max publish {a[:, 0].max():.0f}")

used = te[:, :, 0].any(axis=1)
nwg = int(np.nonzero(used)[0].max()) + 1 if used.any() else 0
te = te[:nwg].astype(np.int64)
E0 = te[:, :, 0][te[:, :, 0] > 0].min()
st = (te[:, :, 0] - E0) * 10
en = (te[:, :, 1] - E0) * 10
kind = te[:, 0, 3]
print(f"k_emit2: {nwg} workgroups (fine active {int((kind == 1).sum())}, fine idle {int((kind == 2).sum())}, slices {int((kind == 3).sum())}); span {en.max()} ns")
for kk, lab in ((1, "fine"), (3, "slice")):
    sel = kind == kk
    if not sel.any():
        continue
    d = (en - st)[sel]
    print(f"  {lab}: wave life mean {d.mean():.0f} p50 {np.percentile(d, 50):.0f} p90 {np.percentile(d, 90):.0f} max {d.max()} ns; start p50 {np.percentile(st[sel], 50):.0f} p99 {np.percentile(st[sel], 99):.0f} max {st[sel].max()}; end p50 {np.percentile(en[sel], 50):.0f} p90 {np.percentile(en[sel], 90):.0f} p99 {np.percentile(en[sel], 99):.0f} max {en[sel].max()}")
    if kk == 3:
        nbt = te[:, :, 2][sel]
        print(f"    batches per wave: mean {nbt.mean():.1f} p90 {np.percentile(nbt, 90):.0f} max {nbt.max()}")
        # how many waves are still running over time
        grid = np.arange(0, en.max() + 1, 5000)
        act = [(int(((st[sel] <= g) & (en[sel] > g)).sum()), int(((st[kind == 1] <= g) & (en[kind == 1] > g)).sum())) for g in grid]
        print("    waves alive every 5 us (slice, fine):", act)
        late = np.argsort(-en[sel].max(axis=1))[:8]
        idx = np.nonzero(sel)[0]
        for j in late:
            wg = idx[j]
            print(f"    late wg {int(wg)}: starts {st[wg].tolist()} ends {en[wg].tolist()} batches {te[wg, :, 2].tolist()}")
if out:
    json.dump({"count": tc.tolist(), "emit_start": st.tolist(), "emit_end": en.tolist(), "emit_kind": kind.tolist()}, open(out, "w"))
