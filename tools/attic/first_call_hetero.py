"""First conversion of a freshly uploaded scene that AUTO sends to the multi-pass pipeline (synth.sponza_like), against a repeated one.
usage: python tools/first_call_hetero.py   (M2S_DEBUG=1 M2S_NO_SCRATCH_WARM=1: without the scratch warm-up at upload)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
scene = synth.sponza_like()
firsts, seconds = [], []
for k in range(4):
    c = Converter(0)
    c.set_resolution_hint(1024)
    c.upload_scene(scene)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); c.convert(1024); firsts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); c.convert(1024); seconds.append((time.perf_counter() - t0) * 1e3)
    c.close()
print("first ms", [round(x, 4) for x in firsts], "second ms", [round(x, 4) for x in seconds], "env", os.environ.get("M2S_NO_SCRATCH_WARM"))
