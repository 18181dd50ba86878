#!/bin/bash
# A/B on one box: non-temporal loads of the triangle phase's planes; timing probe without the per-fragment attribute loads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4nt}
cd $R
for w in c3 c2; do
for rep in 1 2; do
  for L in . ab_nt ab_noplanes; do
    M2S_LIB_PATH=$R/mesh2splat_amd/_build/$L/libm2s_hip.so python bench.py --workload $w --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cold_path') or {}; print('$w $L'.ljust(24), 'step %.4f kernel(ev) %.4f dedicated %.4f first %.4f cold kernel %.4f' % (d['ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused'], c.get('first_call_ms', 0), (c.get('cold_inputs') or {}).get('kernel_ms', 0)))" | tee -a $O/${TAG}.log
  done
done
done
