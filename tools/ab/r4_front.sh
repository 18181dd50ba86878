#!/bin/bash
# A/B on one box: the next strip's front (claim, entry, UV rows) taken before / after the current strip's stores
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4front}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_async.py tests/test_gpu_persistent.py -m gpu -q -x 2>&1 | tail -3
for w in c3 c2; do
for rep in 1 2 3; do
  for L in . ab_late; do
    M2S_LIB_PATH=$R/mesh2splat_amd/_build/$L/libm2s_hip.so python bench.py --workload $w --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cold_path') or {}; print('$w $L'.ljust(24), 'step %.4f kernel(ev) %s dedicated %.4f first %.4f cold kernel %.4f' % (d['ms_per_step'], {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, d['kernel_ms_dedicated']['fused'], c.get('first_call_ms', 0), (c.get('cold_inputs') or {}).get('kernel_ms', 0)))" | tee -a $O/${TAG}.log
  done
done
done
