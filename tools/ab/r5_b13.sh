#!/bin/bash
# round 5, thirteenth batch: depth sort over the bits in which the keys differ (key - min, one radix pass fewer for a bounded scene)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b13}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests -q -m gpu -k "sort or depth or c5" 2>&1 | tail -4 ) | tee $O/${TAG}.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-cold --no-viewer-extra --no-overlap-extra 2>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['extra_workloads']['c5']['depth_sort']))" | tee -a $O/${TAG}.log
grep -v amdgpu.ids $O/${TAG}_err.log | tail -3
