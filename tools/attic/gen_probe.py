#!/usr/bin/env python
"""Is k_fused2's time a staircase in the number of workgroups (768 resident at a time)?  Kernel ms (HIP events, median) for
cube-spheres around the C3 size at R = 1024; workgroups = ceil(T / 256), generations = workgroups / 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
tex = synth.procedural_textures(2048)
for n in [int(x) for x in sys.argv[1:]] or (253, 262, 271, 277, 280, 283, 286, 289, 292, 295, 300, 310, 320):
    scene = synth.cube_sphere(n, tex_size=0)
    scene.meshes[0].textures = tex
    c = Converter(0); c.set_pipeline("team"); c.upload_scene(scene)
    for _ in range(4): tot = c.convert(1024)
    c.set_profiling(True)
    ms = []
    for _ in range(30):
        c.convert(1024); ms.append(c.last_kernel_ms()["fused"])
    T = scene.n_triangles
    wg = (T + 255) // 256
    m = float(np.median(ms))
    print(f"n {n} T {T} wgs {wg} generations {wg/768:.3f} frags {tot} kernel_ms {m:.4f} ns/tri {m*1e6/T:.2f} us/generation {m*1e3/(wg/768):.2f}", flush=True)
    c.close()
