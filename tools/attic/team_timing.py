#!/usr/bin/env python
"""Debug tool: per-workgroup cycle accounting of k_fused2 from a -DM2S_TIMING build.
   make -C mesh2splat_amd/csrc OUT=../_build/timing EXTRA=-DM2S_TIMING
   M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so python tools/team_timing.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import _lib, synth
from mesh2splat_amd.converter import Converter
N, R = int(os.environ.get('TT_N', 289)), int(os.environ.get('TT_R', 1024))
scene = synth.cube_sphere(N, tex_size=2048)
c = Converter(0); c.set_pipeline("team"); c.set_resolution_hint(R); c.upload_scene(scene)
c.set_max_gaussians(0)
for _ in range(3): c.convert(R)
c.set_profiling(True); n = c.convert(R); print('n', N, 'R', R, 'triangles', scene.n_triangles, "gaussians", n, c.last_kernel_ms())
L = _lib.load(); S, B = 16, 8192
buf = np.zeros(S * B, np.uint64)
assert L.m2s_debug_read_timing2(buf.ctypes.data_as(C.c_void_p), C.c_size_t(S * B)) == 0
t = buf.reshape(S, B).astype(np.float64)
nb = int((t[7] > 0).sum())          # (a banded launch ends its bands in smaller workgroups: more than triangles / 256)
t = t[:, t[7] > 0]
def st(x): return f"median {np.median(x):9.0f} mean {x.mean():9.0f} p90 {np.percentile(x, 90):9.0f}"
print("workgroups", nb, "(wave 0 of each)")
for i, name in enumerate(["total", "wait counts", "wait entries", "wait base", "strips", "entries / workgroup"]):
    print(f"{name:22s}", st(t[i]))
for i, name in ((11, "until counts published"), (13, "wave 0 in strip loop"), (14, "wave 3 in strip loop"), (15, "wave 3 strips")):
    print(f"{name:22s}", st(t[i]))
print("cycles per strip: wave 0 %.0f, wave 3 %.0f" % (t[13].sum() / max(t[4].sum(), 1), t[14].sum() / max(t[15].sum(), 1)))

# timeline: how many workgroups are in flight over the kernel's duration, and when each XCD runs out of work
t0, t1, xcd = t[6], t[7], t[8].astype(int)
if t1.max() > 0:
    z = t0.min(); t0 = t0 - z; t1 = t1 - z; span = t1.max()
    print(f"kernel span {span:.0f} ticks of 10 ns (first start to last end, wave 0 of each workgroup); sum of workgroup times / span = {(t1 - t0).sum() / span:.1f} in flight on average")
    edges = np.linspace(0, span, 21)
    occ = [(np.minimum(t1, b) - np.maximum(t0, a)).clip(0).sum() / (b - a) for a, b in zip(edges[:-1], edges[1:])]
    print("in flight per 5 % of the span:", " ".join(f"{o:.0f}" for o in occ))
    for x in range(8):
        m = xcd == x
        if m.any(): print(f"XCD {x}: workgroups {m.sum():5d} first start {t0[m].min():7.0f} last start {t0[m].max():7.0f} last end {t1[m].max():7.0f} entries {t[5][m].sum():9.0f} busy {(t1[m] - t0[m]).sum():11.0f}")

if os.environ.get("TT_DETAIL"):
    # what makes the slow workgroups slow: duration against the number of strips wave 0 shaded and the entries of the workgroup
    tot, strips0, ent = t[0], t[4].astype(int), t[5]
    for k in sorted(set(strips0)):
        m = strips0 == k
        print(f"wave 0 shaded {k} strips: {m.sum():5d} workgroups, total median {np.median(tot[m]):8.0f} max {tot[m].max():8.0f}, entries median {np.median(ent[m]):6.0f} max {ent[m].max():6.0f}, counts published after {np.median(t[11][m]):7.0f}")
    q = np.argsort(-tot)[:8]
    print("slowest:", [(int(tot[i]), int(strips0[i]), int(ent[i]), int(t[11][i]), int(t[2][i]), int(t[3][i])) for i in q], "(total, strips of wave 0, entries, until counts, wait entries, wait base)")
    print("entries per workgroup: min %d p10 %d median %d p90 %d max %d; more than 1024: %d, more than 768: %d" % (ent.min(), np.percentile(ent, 10), np.median(ent), np.percentile(ent, 90), ent.max(), (ent > 1024).sum(), (ent > 768).sum()))
