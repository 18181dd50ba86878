"""Where the one-shot CLI's "HIP runtime + context" time goes: library load, m2s_create (first HIP call: runtime + device bring-up,
streams, work buffers), m2s_prepare (pinned staging for the upload / for the export).  usage: python tools/init_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t0 = time.perf_counter()
from mesh2splat_amd import _lib   # noqa: E402
L = _lib.load()
t1 = time.perf_counter()
h = C.c_void_p()
assert L.m2s_create(0, C.byref(h)) == 0
t2 = time.perf_counter()
assert L.m2s_prepare(h, 1) == 0
t3 = time.perf_counter()
assert L.m2s_prepare(h, 2) == 0
t4 = time.perf_counter()
h2 = C.c_void_p()
assert L.m2s_create(0, C.byref(h2)) == 0
t5 = time.perf_counter()
print("load library %.1f | m2s_create (first HIP call) %.1f | prepare(upload) %.1f | prepare(export) %.1f | a second m2s_create %.1f ms"
      % tuple((b - a) * 1e3 for a, b in ((t0, t1), (t1, t2), (t2, t3), (t3, t4), (t4, t5))))
