#!/bin/bash
# C2 stand-in (one generation of workgroups): number of batches the table aims at, A/B on one box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4c2ab2}
cd $R
B="python bench.py --workload c2 --steps 200 --warmup 10 --no-cold --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5"
for rep in 1 2; do
  for v in "" "M2S_DEBUG=1 M2S_BATCH_TARGET=3072" "M2S_DEBUG=1 M2S_BATCH_TARGET=3072 M2S_BATCH_CT=0" "M2S_DEBUG=1 M2S_BATCH_TARGET=2560" "M2S_DEBUG=1 M2S_BATCH_TARGET=3000"; do
    env $v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.ljust(52), 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused']))" | tee -a $O/${TAG}.log
  done
done
export M2S_LIB_PATH=$R/mesh2splat_amd/_build/timing/libm2s_hip.so
M2S_DEBUG=1 M2S_BATCH_TARGET=3072 TT_DETAIL=1 TT_N=76 TT_R=512 python tools/team_timing.py 2>&1 | tail -8 | tee -a $O/${TAG}.log
