#!/usr/bin/env python
"""Debug tool (-DM2S_TIMING build): when does each XCD start and finish its band of a k_sparse launch?  Prints, per XCD, the span
from its first workgroup's start to its last one's end (ticks of 10 ns, s_memrealtime), the sum of wave-0 lifetimes and the
number of workgroups, for the banded launches of BASELINE config 5 at full size (or a cube-sphere: XS_N, XS_R).
   M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so [M2S_DEBUG=1 M2S_BAND_COST=t,f] python tools/xcd_spans.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mesh2splat_amd import _lib, synth                # noqa: E402
from mesh2splat_amd.converter import Converter        # noqa: E402

if os.environ.get("XS_N"):
    n, R = int(os.environ["XS_N"]), int(os.environ.get("XS_R", 1448))
    scene = synth.cube_sphere(n, tex_size=2048)
else:
    R = 2048
    scene = synth.c5_scene(1021, 4096, 4, cache="/tmp/c5_sphere_1021.npy")
c = Converter(0)
c.set_pipeline("sparse")
c.upload_scene(scene)
c.set_max_gaussians(0)
c.set_profiling(True)
L = _lib.load()
buf = np.zeros(32, np.uint64)


def read():
    assert L.m2s_debug_read_xcd_spans_sparse(buf.ctypes.data_as(C.c_void_p)) == 0
    return buf.copy()


total = c.convert(R)          # without bands (records the workgroups' bases, cuts the bands)
read()
for it in range(int(os.environ.get("XS_ITERS", 3))):
    assert c.convert(R) == total
    ms = c.last_kernel_ms()["fused"]
    b = read().astype(np.float64)
    t0 = b[:8].min()
    start, end, busy, wgs = b[:8] - t0, b[8:16] - t0, b[16:24], b[24:32]
    print(f"launch {it}: kernel {ms:.4f} ms ({c.last_pipeline}); per XCD  end of last workgroup: " + " ".join(f"{e:.0f}" for e in end) +
          f"   (max / mean {end.max() / end.mean():.3f})")
    print("          busy (sum of wave-0 lifetimes): " + " ".join(f"{x / 1e3:.0f}k" for x in busy) + "   workgroups: " + " ".join(f"{int(w)}" for w in wgs))
