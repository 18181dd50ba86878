#!/bin/bash
# A/B of the persistent form of k_fused2 (k_fused2p, tickets) against the one-unit-per-workgroup form, same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4ab}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_async.py tests/test_gpu_edge.py tests/test_gpu_sparse.py -m gpu -q -x > $O/${TAG}_tests.log 2>&1; tail -5 $O/${TAG}_tests.log
B="python bench.py --steps 60 --warmup 5 --no-cold --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5"
for rep in 1 2; do
  for v in "" "M2S_DEBUG=1 M2S_NO_PERSIST=1"; do
    env $v $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v'.ljust(32), 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f overlapped %s' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused'], (d.get('overlapped') or {}).get('ms_per_step')))" | tee -a $O/${TAG}_ab.log
  done
done
