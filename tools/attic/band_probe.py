"""AUTO's crossover band: cube-spheres of 11-18 fragments per triangle at R = 1024 under auto / multipass / team: blocking ms, kernel ms
(HIP events), what ran.  usage: python tools/band_probe.py [n ...]   (default 140 127 115: 11.6 / 14.1 / 17.2 fragments per triangle)"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch  # noqa: F401
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    R = 1024
    for n in [int(x) for x in sys.argv[1:]] or [140, 127, 115]:
        scene = synth.cube_sphere(n, tex_size=2048)
        for setting in ("auto", "multipass", "team"):
            c = Converter(0)
            c.set_pipeline(setting)
            c.set_resolution_hint(R)
            c.upload_scene(scene)
            total = c.convert(R)
            c.convert(R)
            c.set_profiling(True)
            wall, kern = [], []
            for _ in range(30):
                t0 = time.perf_counter()
                c.convert(R)
                wall.append((time.perf_counter() - t0) * 1e3)
                kern.append(sum(c.last_kernel_ms().values()))
            b = 96.0 * c.num_stored + 144.0 * scene.n_triangles
            print(json.dumps({"n": n, "triangles": scene.n_triangles, "frags_per_triangle": round(total / scene.n_triangles, 2), "setting": setting, "ran": c.last_pipeline,
                              "blocking_ms": round(float(np.median(wall)), 4), "kernels_ms": round(float(np.median(kern)), 4),
                              "frac_kernels": round(b / (float(np.median(kern)) * 1e-3) / 8e12, 4)}), flush=True)
            c.close()


if __name__ == "__main__":
    main()
