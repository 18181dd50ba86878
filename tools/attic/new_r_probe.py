"""Blocking conversions at densities the context has never seen (what a move of the reference's density slider costs), against
repeated ones: python tools/new_r_probe.py [n=289] [R0=1024].  (Round 4 used it to try handing the units of a launch without a run
table out in short runs over ONE chain — no gain: profiles/r04/ab_new_density_chained_runs_negative.jsonl.)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
n = int(sys.argv[1]) if len(sys.argv) > 1 else 289
R0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
scene = synth.cube_sphere(n, tex_size=2048)
c = Converter(0)
c.set_resolution_hint(R0)
c.upload_scene(scene)
for _ in range(5):
    c.convert(R0)
same = []
for _ in range(20):
    t0 = time.perf_counter(); c.convert(R0); same.append((time.perf_counter() - t0) * 1e3)
new = []
for k in range(1, 41):
    R = R0 - 4 * k
    t0 = time.perf_counter(); c.convert(R); new.append((time.perf_counter() - t0) * 1e3 * (R0 / R) ** 2)   # (scaled to R0's record count)
second = []
for k in range(1, 41):
    R = R0 - 4 * k
    t0 = time.perf_counter(); c.convert(R); second.append((time.perf_counter() - t0) * 1e3 * (R0 / R) ** 2)
print(json.dumps({"switches": {k: v for k, v in os.environ.items() if k.startswith("M2S_")}, "same_R_ms": float(np.median(same)),
                  "new_R_ms_scaled": float(np.median(new)), "second_time_at_those_R_ms_scaled": float(np.median(second)),
                  "new_over_same": float(np.median(new) / np.median(same))}))
