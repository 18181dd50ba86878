"""k_sparse against k_fused2 over the density range where one hands over to the other: one 6.2 M-triangle cube-sphere (n = 721),
R from 512 to 2500 = 0.11 ... 2.6 fragments per triangle.  Kernel times by HIP events (m2s_set_profiling), median of 7.
usage: python tools/sparse_crossover.py [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mesh2splat_amd import synth                      # noqa: E402
from mesh2splat_amd.converter import Converter        # noqa: E402

n = int(os.environ.get("SC_N", 721))
scene = synth.cube_sphere(n, tex_size=2048)
T = scene.n_triangles
rows = []
convs = {}
for pipe in ("team", "sparse"):
    c = Converter(0)
    c.set_pipeline(pipe)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    c.set_profiling(True)
    convs[pipe] = c
for R in [int(x) for x in os.environ.get("SC_RS", "512,724,1024,1448,1774,2048,2500").split(",")]:
    row = {"R": R}
    for pipe, c in convs.items():
        total = c.convert(R)
        ms = []
        for _ in range(7):
            assert c.convert(R) == total
            ms.append(c.last_kernel_ms()["fused"])
        row[pipe] = float(np.median(ms))
        row[pipe + "_ran"] = c.last_pipeline
        row["gaussians"] = int(total)
    row["fragments_per_triangle"] = row["gaussians"] / T
    row["frac_team"] = (96.0 * row["gaussians"] + 144.0 * T) / (row["team"] * 1e-3) / 8e12
    row["frac_sparse"] = (96.0 * row["gaussians"] + 144.0 * T) / (row["sparse"] * 1e-3) / 8e12
    rows.append(row)
    print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump({"triangles": T, "rows": rows}, open(sys.argv[1], "w"), indent=1)
