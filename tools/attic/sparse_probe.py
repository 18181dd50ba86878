"""Kernel time of the single-pass forms on scenes whose triangles are mostly smaller than a pixel (the C5 regime: T >> N).
usage: [M2S_LIB_PATH=...] python tools/sparse_probe.py [n:R ...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
cases = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(289, 1024), (721, 1024), (1021, 2048)]
for n, R in cases:
    scene = synth.cube_sphere(n, tex_size=2048)
    c = Converter(0); c.upload_scene(scene); c.set_max_gaussians(0); c.set_profiling(True)
    row = []
    for pipe in ("team", "wave"):
        c.set_pipeline(pipe)
        total = c.convert(R)
        ms = []
        for _ in range(12):
            c.convert(R); ms.append(c.last_kernel_ms()["fused"])
        row.append((pipe, c.last_pipeline, round(float(np.median(ms)), 4)))
    alg = 96.0 * total + 144.0 * scene.n_triangles
    print("n", n, "R", R, "T", scene.n_triangles, "N", total, "frags/tri", round(total / scene.n_triangles, 2), row,
          "frac(best)", round(alg / (min(r[2] for r in row) * 1e-3) / 8e12, 3), flush=True)
    c.close()
