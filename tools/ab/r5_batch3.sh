#!/bin/bash
# round 5, third batch: heaviest-runs-first dispatch (launch_run_order) against mesh order, k_fused3 / k_fused2 / k_sparse
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b3}
cd $R; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1'.ljust(34), 'step %.4f sync %.4f kernel(ev) %s dedicated %.4f' % (d['ms_per_step'], d.get('sync_ms_per_step', 0), {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, (d.get('kernel_ms_dedicated') or {}).get('fused', 0)), d['config'].get('pipeline'))"; }
for rep in 1 2; do
  for V in "lean4_order::" "lean4_meshorder::M2S_NO_RUN_ORDER=1" "fused2_order::M2S_NO_LEAN=1" "fused2_meshorder::M2S_NO_LEAN=1 M2S_NO_RUN_ORDER=1"; do
    IFS=: read name dir envs <<< "$V"
    env M2S_DEBUG=1 $envs timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 --no-cold 2>$O/${TAG}_err.log | line "c3 $name" | tee -a $O/${TAG}.log
  done
done
for V in "order:" "meshorder:M2S_NO_RUN_ORDER=1"; do
  IFS=: read name envs <<< "$V"
  env M2S_DEBUG=1 $envs timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-viewer-extra --no-cold 2>>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for w,c in (d.get('extra_workloads') or {}).items():
    print('extras $name', w, {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ('ms_per_step','kernels_total_ms','pipeline','error') or k.startswith('kernel')})" | tee -a $O/${TAG}.log
done
