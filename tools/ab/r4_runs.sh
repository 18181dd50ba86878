#!/bin/bash
# XCD runs (r4) against the eight equal-work bands of round 3 (mesh2splat_amd/_build/ab_bands: the library before the change)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4r}
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x > $O/${TAG}_tests.log 2>&1; tail -4 $O/${TAG}_tests.log
B="python bench.py --steps 60 --warmup 5 --no-cold --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5"
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1'.ljust(44), 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f overlapped %s' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused'], (d.get('overlapped') or {}).get('ms_per_step')))"; }
for rep in 1 2; do
  M2S_LIB_PATH=mesh2splat_amd/_build/ab_bands/libm2s_hip.so $B 2>/dev/null | line "bands (round 3)" | tee -a $O/${TAG}_ab.log
  for sh in 3 4 5 6; do
    M2S_DEBUG=1 M2S_RUN_SHIFT=$sh $B 2>/dev/null | line "runs of $((1<<sh)) units" | tee -a $O/${TAG}_ab.log
  done
  M2S_DEBUG=1 M2S_NO_BANDS=1 $B 2>/dev/null | line "plain order (no runs)" | tee -a $O/${TAG}_ab.log
done
M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so python tools/team_timing.py 2>&1 | grep -E "fused|total  |in flight per|XCD|kernel span" | tee $O/${TAG}_timeline.log
