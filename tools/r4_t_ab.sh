#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4x}
cd $R
bash tools/r4_timing.sh $TAG 2>&1 | grep -E "==|total  |strip|barrier|ticket|fused|in flight per|wait"
B="python bench.py --steps 60 --warmup 5 --no-cold --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5"
for rep in 1 2; do
  for v in "" "M2S_DEBUG=1 M2S_NO_PERSIST=1"; do
    env $v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.ljust(32), 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f overlapped %s' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused'], (d.get('overlapped') or {}).get('ms_per_step')))" | tee -a $O/${TAG}_ab.log
  done
done
