#!/bin/bash
# round 5, twelfth batch: k_fused3 without a stack (first conversion), scratch warm-up at upload for multi-pass scenes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b12}
cd $R; mkdir -p $O
for E in "" "M2S_DEBUG=1 M2S_NO_SCRATCH_WARM=1" ""; do env $E timeout 200 python tools/first_call_hetero.py 2>/dev/null | tee -a $O/${TAG}.log; done
timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 2>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); cp=d['cold_path']; r=d['roofline']
print('c3 step %.4f sync %.4f dedicated %s first %.4f (+warm %.4f) second %.4f newR %.4f same %.4f' % (d['ms_per_step'], d['sync_ms_per_step'], round(d['kernel_ms_dedicated']['fused'],4), cp['first_call_ms'], cp['first_call_plus_warm_ms'], cp['second_call_ms'], cp['new_R_ms']['median'], cp['same_R_sync_ms']))" | tee -a $O/${TAG}.log
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py -q -m gpu 2>&1 | tail -3 ) | tee -a $O/${TAG}.log
