"""Plain memory rates of the box (torch ops on 1 GiB tensors): copy, write-only (fill), read-only (sum) — context for the roofline
fractions (a kernel that mostly WRITES records cannot beat the write-only rate).  usage: python tools/mem_probe.py"""
import json
import time

import torch

n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device="cuda")
b = torch.empty(n, dtype=torch.uint8, device="cuda")
f = torch.empty(n // 4, dtype=torch.float32, device="cuda")


def rate(fn, nbytes, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nbytes * reps / (time.perf_counter() - t0) / 1e9


out = {"copy_GBps_read_plus_write": rate(lambda: b.copy_(a), 2 * n), "write_only_GBps_fill": rate(lambda: f.fill_(1.0), n),
       "read_only_GBps_sum": rate(lambda: f.sum(), n)}
print(json.dumps(out))
