"""Kernel time of one scene under several single-pass forms on the same box: python tools/pipe_ab.py n:tex:R[,..] pipe[,pipe..]
(M2S_LIB_PATH selects an A/B build).  Blocking conversions, HIP events (m2s_set_profiling), median of 15."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter

cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1].split(",")]
pipes = sys.argv[2].split(",")
for n, tex, R in cases:
    scene = synth.cube_sphere(n, tex_size=tex)
    row = {}
    for pipe in pipes:
        c = Converter(0)
        c.set_pipeline(pipe)
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        c.set_profiling(True)
        total = c.convert(R)
        c.convert(R)
        ms = []
        for _ in range(15):
            c.convert(R)
            ms.append(sum(c.last_kernel_ms().values()))
        row[pipe] = (c.last_pipeline, round(float(np.median(ms)), 4))
        c.close()
    alg = 96.0 * total + 144.0 * scene.n_triangles
    print(f"n={n} R={R} T={scene.n_triangles} N={total} f/t={total / scene.n_triangles:.2f}", row,
          {p: round(alg / (v[1] * 1e-3) / 8e12, 3) for p, v in row.items()}, flush=True)
