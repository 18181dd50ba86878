#!/bin/bash
# round 5, seventh batch: the whole GPU suite on the current tree; mid after the unclipped dense batches; C2 stand-in lean vs team with
# the wave-walked coverage in both; C3 team against _build/base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b7}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) | tee $O/${TAG}_tests.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1'.ljust(26), 'step %.4f sync %.4f kernel(ev) %s dedicated %s' % (d['ms_per_step'], d.get('sync_ms_per_step', 0), {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, {k:round(v,4) for k,v in (d.get('kernel_ms_dedicated') or {}).items() if isinstance(v,float) and v}), d['config'].get('pipeline'))"; }
B="M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so"
for rep in 1 2; do
  for V in "c2_auto:c2::" "c2_lean:c2:--pipeline lean:" "c2_team_base:c2::$B" "c3_team:c3:--pipeline team:" "c3_team_base:c3:--pipeline team:$B" "c3_auto:c3::" "mid_new:mid::" "mid_base:mid::$B" "hetero_new:hetero::"; do
    IFS=: read name wl flags envs <<< "$V"
    env $envs timeout 300 python bench.py --workload $wl $flags --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 --no-cold 2>$O/${TAG}_err.log | line "$name" | tee -a $O/${TAG}.log
  done
done
grep -v amdgpu.ids $O/${TAG}_err.log | tail -5
