#!/usr/bin/env python
"""Debug tool: per-phase timing of k_fused from a -DM2S_TIMING build (not part of the product).
   make -C mesh2splat_amd/csrc OUT=../_build/timing EXTRA=-DM2S_TIMING
   M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so python tools/fused_timing.py [workload]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import _lib, synth  # noqa: E402
from mesh2splat_amd.converter import Converter  # noqa: E402

wl = {"c3": (289, 2048, 1024), "c2": (76, 2048, 512), "small": (24, 256, 256)}[sys.argv[1] if len(sys.argv) > 1 else "c3"]
scene = synth.cube_sphere(wl[0], tex_size=wl[1])
conv = Converter(0)
conv.upload_scene(scene)
for _ in range(3):
    n = conv.convert(wl[2])
conv.set_profiling(True)
n = conv.convert(wl[2])
print("gaussians", n, "kernel ms", conv.last_kernel_ms())
L = _lib.load()
W, S = 16384, 16
buf = np.zeros(S * W, np.uint64)
rc = L.m2s_debug_read_timing(buf.ctypes.data_as(C.c_void_p), C.c_size_t(S * W))
assert rc == 0
t = buf.reshape(S, W).astype(np.int64)
nw = min(W, (scene.n_triangles + 63) // 64)
t = t[:, :nw]
t0 = t[0].min()
names = ["load+setup", "count", "scan+publish", "trishade", "expand", "shade strip0", "resolve base", "rest (stores+more strips)"]
print(f"waves {nw}; kernel span {(t[8].max() - t0)} ticks")
for i, nm in enumerate(names):
    d = (t[i + 1] - t[i]).astype(np.float64)
    print(f"{nm:28s} median {np.median(d):9.0f}  mean {d.mean():9.0f}  p90 {np.percentile(d, 90):9.0f}  max {d.max():9.0f}")
life = (t[8] - t[0]).astype(np.float64)
print(f"{'wave lifetime':28s} median {np.median(life):9.0f}  mean {life.mean():9.0f}  p90 {np.percentile(life, 90):9.0f}")
start = t[0] - t0
print("wave start time percentiles (ticks):", [int(np.percentile(start, q)) for q in (1, 25, 50, 75, 99)])
sub = [("  strip0: LDS + uv loads -> U,V", t[12] - t[5]), ("  strip0: texel fetch + filter", t[13] - t[12]), ("  strip0: interp + TBN", t[14] - t[13])]
for nm, d in sub:
    d = d.astype(np.float64)
    print(f"{nm:34s} median {np.median(d):9.0f}  mean {d.mean():9.0f}  p90 {np.percentile(d, 90):9.0f}")
print("frags/wave mean", t[9].mean(), "xcc ids", np.unique(t[10]))
# concurrency estimate: sum of lifetimes / span
print("avg concurrent waves", life.sum() / (t[8].max() - t0))
