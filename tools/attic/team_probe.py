#!/usr/bin/env python
"""Debug probe: k_fused (wave) vs k_fused2 (team) on the C3 workload: bit-identical output? kernel time?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
n, tex, R = (289, 2048, 1024) if len(sys.argv) < 2 else (int(sys.argv[1]), 256, int(sys.argv[2]))
scene = synth.cube_sphere(n, tex_size=tex)
outs = {}
for pipe in ("wave", "team"):
    c = Converter(0); c.set_pipeline(pipe); c.upload_scene(scene)
    for _ in range(3): tot = c.convert(R)
    t0 = time.perf_counter()
    for _ in range(20): tot = c.convert(R)
    dt = (time.perf_counter() - t0) / 20
    c.set_profiling(True); c.convert(R); ms = c.last_kernel_ms(); c.set_profiling(False)
    outs[pipe] = (tot, c.download())
    print(pipe, "total", tot, "ms/convert", round(dt * 1e3, 4), {k: round(v, 4) for k, v in ms.items() if v})
    c.close()
a, b = outs["wave"], outs["team"]
print("totals equal", a[0] == b[0], "records bit-identical", a[1].shape == b[1].shape and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)))
if a[1].shape == b[1].shape and not np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)):
    d = np.argwhere((a[1].view(np.uint32) != b[1].view(np.uint32)).any(axis=1))[:, 0]
    print("differing records", len(d), d[:10])
