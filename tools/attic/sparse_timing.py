#!/usr/bin/env python
"""Debug tool: per-workgroup cycle accounting of k_sparse from a -DM2S_TIMING build.
   make -C mesh2splat_amd/csrc OUT=../_build/timing EXTRA=-DM2S_TIMING
   M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so TT_N=721 TT_R=1448 python tools/sparse_timing.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import _lib, synth
from mesh2splat_amd.converter import Converter
N, R = int(os.environ.get('TT_N', 721)), int(os.environ.get('TT_R', 1448))
scene = synth.cube_sphere(N, tex_size=2048)
c = Converter(0); c.set_pipeline("sparse"); c.upload_scene(scene)
c.set_max_gaussians(0)
for _ in range(3): c.convert(R)
c.set_profiling(True); n = c.convert(R); print('n', N, 'R', R, 'triangles', scene.n_triangles, "gaussians", n, c.last_pipeline, c.last_kernel_ms())
L = _lib.load(); S, B = 32, 16384
buf = np.zeros(S * B, np.uint64)
assert L.m2s_debug_read_timing_sparse(buf.ctypes.data_as(C.c_void_p), C.c_size_t(S * B)) == 0
t = buf.reshape(S, B).astype(np.float64)
nb = min(B, (scene.n_triangles + 511) // 512)
t = t[:, :nb]
def st(x): return f"median {np.median(x):9.0f} mean {x.mean():9.0f} p10 {np.percentile(x, 10):9.0f} p90 {np.percentile(x, 90):9.0f}"
print("workgroups", nb, "(wave 0 of each)")
for i, name in enumerate(["total", "until survivors listed", "in rounds", " waiting for previous round", "waiting for last count", "in strips (incl. that wait)",
                          " waiting for expansions", " waiting for base", "rounds / workgroup", "entries / workgroup", "survivors / workgroup",
                          "rounds by wave 0", "strips by wave 0"]):
    print(f"{name:30s}", st(t[i]))
print("first round of wave 0 (cumulative):")
for i, name in zip(range(16, 21), ["inputs + geo + raster setup", "+ coverage", "+ scans, prefix wait, publish", "+ fragment constants -> LDS", "+ entries -> LDS"]):
    sel = t[11] > 0
    print(f"  {name:28s}", st(t[i][sel]))
print("first strip of wave 0 (cumulative from entry read):")
for i, name in zip(range(22, 26), ["uv arrived", "texels filtered", "interpolation + TBN", "shade_from_tri done"]):
    sel = t[12] > 0
    print(f"  {name:28s}", st(t[i][sel]))
