#!/usr/bin/env python
"""GPU probe: k_prepass on the records of the C3 conversion (2.74 M Gaussians) — kernel time by HIP events, algorithmic
bytes (96 B read per Gaussian + 100 B written per survivor) over it, fraction of the 8 TB/s HBM peak."""
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import camera  # noqa: E402
from mesh2splat_amd import synth  # noqa: E402
from mesh2splat_amd.converter import Converter  # noqa: E402
from mesh2splat_amd.prepass import PrepassParams  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 289
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    only = sys.argv[3] if len(sys.argv) > 3 else ""          # "input" / "arrival": just that append order, whole sphere in view
    c = Converter(0)
    c.upload_scene(synth.cube_sphere(n, tex_size=2048 if n >= 200 else 256))
    c.set_max_gaussians(0)
    total = c.convert(R)
    res = (1920, 1080)
    out = {"records": total}
    for name, eye, arrival in (("outside", (1.6, 1.1, 2.3), False), ("close", (0.9, 0.5, 0.9), False),
                               ("outside_arrival_order", (1.6, 1.1, 2.3), True), ("close_arrival_order", (0.9, 0.5, 0.9), True)):
        if only and name != {"input": "outside", "arrival": "outside_arrival_order"}[only]:
            continue
        p = PrepassParams(view_mat=camera.look_at(eye, (0.1, 0.0, -0.1)), proj_mat=camera.perspective(45.0, res[0] / res[1], 0.01, 100.0),
                          renderer_resolution=res, resolution_target=R, arrival_order=arrival)
        c.set_profiling(True)
        ms = []
        for _ in range(30):
            vis = c.prepass(p, download=False)
            ms.append(c.last_prepass_ms)
        sms = []
        for _ in range(8):
            c.sort_prepass(download=False)
            sms.append(c.last_sort_prepass_ms)
        c.set_profiling(False)
        ms = np.array(ms[5:])
        b = 96 * total + 100 * vis
        out[name] = {"visible": vis, "kernel_ms_median": float(np.median(ms)), "kernel_ms_min": float(ms.min()),
                     "alg_bytes": b, "GBps": b / np.median(ms) / 1e6, "frac_of_8TBps": b / (np.median(ms) * 1e-3) / 8e12,
                     "sort_prepass_ms_median": float(np.median(sms[2:]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
