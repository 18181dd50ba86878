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
c = Converter(0); c.set_pipeline("team"); c.upload_scene(scene)
c.set_max_gaussians(0)
for _ in range(3): c.convert(R)
c.set_profiling(True); n = c.convert(R); print('n', N, 'R', R, 'triangles', scene.n_triangles, "gaussians", n, c.last_kernel_ms())
L = _lib.load(); S, B = 16, 8192
buf = np.zeros(S * B, np.uint64)
assert L.m2s_debug_read_timing2(buf.ctypes.data_as(C.c_void_p), C.c_size_t(S * B)) == 0
t = buf.reshape(S, B).astype(np.float64)
nb = min(B, ((scene.n_triangles + 63) // 64 + 3) // 4)
t = t[:, :nb]
def st(x): return f"median {np.median(x):9.0f} mean {x.mean():9.0f} p90 {np.percentile(x, 90):9.0f}"
print("workgroups", nb, "(wave 0 of each)")
for i, name in enumerate(["total", "wait counts", "wait entries", "wait base", "strips", "entries / workgroup"]):
    print(f"{name:22s}", st(t[i]))
