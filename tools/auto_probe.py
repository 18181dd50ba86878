#!/usr/bin/env python
"""Debug probe: AUTO (fused first) vs forced multi-pass over a sweep of fragments per triangle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n in (289, 204, 144, 102, 72, 51, 36, 18):
    scene = synth.cube_sphere(n, tex_size=1024)
    row = []
    for pipe in ("auto", "multipass"):
        c = Converter(0); c.set_pipeline(pipe); c.upload_scene(scene); c.set_max_gaussians(0)
        for _ in range(3): tot = c.convert(R)
        t0 = time.perf_counter()
        for _ in range(20): tot = c.convert(R)
        dt = (time.perf_counter() - t0) / 20
        c.set_profiling(True); c.convert(R); ms = c.last_kernel_ms(); c.set_profiling(False)
        row.append((pipe, round(dt * 1e3, 4), {k: round(v, 3) for k, v in ms.items() if v}))
        c.close()
    print("n", n, "tris", scene.n_triangles, "frags/tri", round(tot / scene.n_triangles, 1), row)
