"""What the FIRST conversion of a freshly uploaded scene costs (blocking m2s_convert, wall clock), against repeated ones:
python tools/first_call_probe.py [n=289] [R=1024] [reps=6].  With a debug build of the library (make EXTRA=-DM2S_DEBUG_BUILD OUT=../_build_debug; M2S_LIB_PATH) and M2S_DEBUG=1: M2S_NO_WARM=1 (round 3's behaviour: count inside the
first call), M2S_NO_WARM_BANDS=1, M2S_NO_WARM_TOUCH=1 switch the parts of the upload-time preparation off."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter

n = int(sys.argv[1]) if len(sys.argv) > 1 else 289
R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
scene = synth.cube_sphere(n, tex_size=2048)
first, second, steady, warm, up = [], [], [], [], []
# FCP_PRE=other: between the upload and the first conversion, ANOTHER context converts its own scene a few times (same kernel, other
# buffers): separates "the GPU / the kernel's code is cold after an upload" from "this context's buffers are cold"
pre = os.environ.get("FCP_PRE")
other = None
if pre:
    other = Converter(0); other.set_resolution_hint(R); other.upload_scene(scene if pre == "same" else synth.cube_sphere(76, tex_size=2048))
for _ in range(reps):
    c = Converter(0)
    c.set_resolution_hint(R)
    c.upload_scene(scene)
    u = c.last_upload_ms()
    up.append(u["total"]); warm.append(u["warm"])
    if other is not None:
        for _k in range(3): other.convert(R)
    t0 = time.perf_counter(); c.convert(R); first.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); c.convert(R); second.append((time.perf_counter() - t0) * 1e3)
    s = []
    for _ in range(10):
        t0 = time.perf_counter(); c.convert(R); s.append((time.perf_counter() - t0) * 1e3)
    steady.append(float(np.median(s)))
    c.close()
print(json.dumps({"switches": {k: v for k, v in os.environ.items() if k.startswith("M2S_")}, "n": n, "R": R,
                  "first_call_ms": [round(x, 4) for x in first], "second_call_ms": [round(x, 4) for x in second],
                  "steady_sync_ms": [round(x, 4) for x in steady], "upload_ms": [round(x, 2) for x in up], "warm_ms": [round(x, 3) for x in warm],
                  "median": {"first": float(np.median(first[1:])), "second": float(np.median(second[1:])), "steady": float(np.median(steady[1:]))}}))
