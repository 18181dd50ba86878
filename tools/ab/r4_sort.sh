#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4s}
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_dist_local.py tests/test_gpu_persistent.py -m gpu -q -x -k "sort or persistent" > $O/${TAG}_tests.log 2>&1; tail -4 $O/${TAG}_tests.log
python - <<PY
import json, sys, time
sys.path.insert(0, "$R")
import numpy as np, torch
import bench
r = bench.c5_workload(torch, 0, steps=6)
print(json.dumps({k: r[k] for k in ("ms_per_step", "kernels_total_ms", "depth_sort", "upload_s")}, indent=1))
json.dump(r, open("$O/${TAG}_c5.json", "w"), indent=1)
PY
