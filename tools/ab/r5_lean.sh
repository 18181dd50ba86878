#!/bin/bash
# round 5: the lean team kernel (k_fused3) against k_fused2 on one box.  tools/ab/r5_lean.sh [tag] [workloads]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5lean}; WL=${2:-"c3 c2"}
cd $R; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cold_path') or {}
print('$1'.ljust(30), 'step %.4f sync %.4f kernel(ev) %s dedicated %.4f first %.4f cold %.4f' % (d['ms_per_step'], d.get('sync_ms_per_step', 0), {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, (d.get('kernel_ms_dedicated') or {}).get('fused', 0), c.get('first_call_ms', 0), (c.get('cold_inputs') or {}).get('kernel_ms', 0)), d['config'].get('pipeline'))"; }
for w in $WL; do
for rep in 1 2; do
  for V in "fused2:.:M2S_NO_LEAN=1" "lean4:.:" "lean4s16:ab_l4s16:" "lean5:ab_l5:"; do
    IFS=: read name dir envs <<< "$V"
    [ -f $R/mesh2splat_amd/_build/$dir/libm2s_hip.so ] || continue
    env M2S_DEBUG=1 $envs M2S_LIB_PATH=$R/mesh2splat_amd/_build/$dir/libm2s_hip.so timeout 300 python bench.py --workload $w --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 2>$O/${TAG}_err.log | line "$w $name" | tee -a $O/${TAG}.log
  done
done
done
