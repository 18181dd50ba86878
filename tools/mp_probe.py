"""Multi-pass workloads of bench.py on their own (mid, c4, hetero [, c2, band]): kernel times, one blocking call, roofline fractions.
    python tools/mp_probe.py [names...]   ->  one JSON line per workload"""
import json
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import torch  # noqa: E402
import bench  # noqa: E402
from mesh2splat_amd.ctl import Ctl  # noqa: E402

names = sys.argv[1:] or ["mid", "c4", "hetero"]
torch.cuda.set_device(0)
ctl = Ctl(0, 1)
pipe = os.environ.get("M2S_PROBE_PIPELINE")      # A/B: force a pipeline setting
if pipe:
    _Rig = bench.Rig
    bench.Rig = lambda *a, **k: _Rig(*a, **dict(k, pipeline=pipe))
for n in names:
    r = bench.extra_workload(torch, ctl, 0, n)
    keep = {k: r[k] for k in ("workload", "gaussians", "pipeline", "kernel_ms", "kernels_total_ms", "blocking_ms", "ms_per_step", "roofline_blocking")}
    keep["frac_kernels"] = r["roofline_whole_conversion"]["frac_of_hbm_peak"]
    keep["overlapped_ms"] = r.get("overlapped", {}).get("ms_per_step")
    print(json.dumps(keep), flush=True)
