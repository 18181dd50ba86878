#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3bt}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_async.py tests/test_gpu_round2.py tests/test_gpu_sparse.py -q -x > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed" $O/${TAG}_tests.log
python tools/mem_probe.py 2>/dev/null | tee $O/${TAG}_mem_probe.json
for v in "" "M2S_DEBUG=1 M2S_NO_BATCH_TABLE=1"; do
  env $v timeout 200 python bench.py --workload c2 --steps 60 --warmup 6 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads --no-overlap-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v c2', d['ms_per_step'], d['kernel_ms'], d['roofline']['frac'], d['kernel_ms_dedicated'])"
done
timeout 300 python bench.py --no-cpu-baseline --no-viewer-extra --no-c5 2>/dev/null > $O/${TAG}_bench.json; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('overlapped'), {k:(v.get('ms_per_step'),v.get('kernel_ms')) for k,v in d.get('extra_workloads',{}).items()})"
