#!/usr/bin/env python
"""Single-pass kernel with fewer triangles per wave on mid-size triangles (env M2S_TPW) against the multi-pass pipeline."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, time
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    n, pipe = int(sys.argv[2]), sys.argv[3]
    scene = synth.cube_sphere(n, tex_size=1024)
    c = Converter(0); c.set_pipeline(pipe); c.upload_scene(scene); c.set_max_gaussians(0)
    for _ in range(3): tot = c.convert(1024)
    ts = []
    for _ in range(24):
        t0 = time.perf_counter(); tot = c.convert(1024); ts.append((time.perf_counter() - t0) * 1e3)
    c.set_profiling(True); c.convert(1024); ms = c.last_kernel_ms()
    print(json.dumps({"n": n, "tris": scene.n_triangles, "fpt": round(tot / scene.n_triangles, 1), "pipe": pipe, "ran": c.last_pipeline,
                      "sync_ms": round(float(np.median(ts)), 4), "kernels": {k: round(v, 4) for k, v in ms.items() if v}}))
else:
    for n in (170, 144, 120, 102, 94, 86, 72):
        for pipe, tpw in (("multipass", None), ("team", 64), ("team", 32), ("team", 16), ("team", 8)):
            env = dict(os.environ)
            if tpw: env["M2S_TPW"] = str(tpw)
            r = subprocess.run([sys.executable, __file__, "child", str(n), pipe], capture_output=True, text=True, env=env)
            print("tpw", tpw, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-200:], flush=True)
