#!/usr/bin/env python
"""Debug probe: big-triangle path (unit quad) and the C4 stand-in, AUTO pipeline, kernel times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
tex = synth.procedural_textures(1024, 3)
for name, scene, Rs in [("quad", synth.unit_quad(textures=tex), (1024, 2048, 4096)), ("c4", synth.sphere_grid(4, n=18, tex_size=256), (1024,))]:
    c = Converter(0); c.upload_scene(scene); c.set_max_gaussians(0)
    for R in Rs:
        c.convert(R); c.convert(R)
        t0 = time.perf_counter()
        for _ in range(20): n = c.convert(R)
        dt = (time.perf_counter() - t0) / 20
        c.set_profiling(True); c.convert(R); ms = c.last_kernel_ms(); c.set_profiling(False)
        print(name, "R", R, "frags", n, "ms", round(dt * 1e3, 4), "B/s", round(n / dt / 1e9, 2), {k: round(v, 4) for k, v in ms.items() if v})
    c.close()
