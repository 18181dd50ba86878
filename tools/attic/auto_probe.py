#!/usr/bin/env python
"""AUTO's crossover: the single-pass kernel (forced: 'team') against the multi-pass pipeline over a sweep of fragments per
triangle (2.74 M fragments at R = 1024 from 1 M ... 3.9 k triangles), blocking conversions, medians."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for n in (289, 236, 204, 170, 144, 120, 102, 86, 72, 51, 36, 18):
    scene = synth.cube_sphere(n, tex_size=1024)
    row = []
    for pipe in ("team", "multipass", "auto"):
        c = Converter(0); c.set_pipeline(pipe); c.upload_scene(scene); c.set_max_gaussians(0)
        for _ in range(3): tot = c.convert(R)
        ts = []
        for _ in range(24):
            t0 = time.perf_counter(); tot = c.convert(R); ts.append((time.perf_counter() - t0) * 1e3)
        c.set_profiling(True); c.convert(R); ms = c.last_kernel_ms(); c.set_profiling(False)
        row.append((pipe, c.last_pipeline, round(float(np.median(ts)), 4), {k: round(v, 3) for k, v in ms.items() if v}))
        c.close()
    print("n", n, "tris", scene.n_triangles, "frags/tri", round(tot / scene.n_triangles, 1), row, flush=True)
