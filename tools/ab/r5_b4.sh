#!/bin/bash
# round 5, fourth batch: (1) parity of the 32-bit small-triangle setup + wave-walked coverage (k_fused3) and the 24-bit texel
# addressing (every kernel); (2) config 3 against the library of the previous commit (_build/base); (3) the heterogeneous scene
# under every pipeline setting
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b4}
cd $R; mkdir -p $O
( timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py tests/test_gpu_edge.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -5 ) | tee $O/${TAG}_tests.log
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1'.ljust(26), 'step %.4f sync %.4f kernel(ev) %s dedicated %.4f' % (d['ms_per_step'], d.get('sync_ms_per_step', 0), {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, (d.get('kernel_ms_dedicated') or {}).get('fused', 0)), d['config'].get('pipeline'))"; }
for rep in 1 2; do
  for V in "c3_new:" "c3_base:M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so" "c3_new_team:M2S_DEBUG=1 M2S_NO_LEAN=1" "c3_base_team:M2S_DEBUG=1 M2S_NO_LEAN=1 M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so"; do
    IFS=: read name envs <<< "$V"
    env $envs timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 --no-cold 2>$O/${TAG}_err.log | line "$name" | tee -a $O/${TAG}.log
  done
done
for V in "extras_new:" "extras_base:M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so"; do
  IFS=: read name envs <<< "$V"
  env $envs timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-viewer-extra --no-cold --no-c5 2>>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for w,c in (d.get('extra_workloads') or {}).items():
    print('$name', w, {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ('ms_per_step','kernels_total_ms','pipeline','error') or k.startswith('kernel')})" | tee -a $O/${TAG}.log
done
timeout 300 python tools/hetero_probe.py 2>>$O/${TAG}_err.log | tee $O/${TAG}_hetero.jsonl | cut -c1-400
timeout 300 python tools/hetero_probe.py --combo-only --no-oracle 2>>$O/${TAG}_err.log | tee $O/${TAG}_hetero_combo.jsonl | cut -c1-400
tail -5 $O/${TAG}_err.log
