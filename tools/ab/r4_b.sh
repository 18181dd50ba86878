#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4b}
cd $R
timeout 900 python -m pytest tests/test_gpu_dist_stub.py tests/test_gpu_sparse.py tests/test_gpu_dropin.py -m gpu -q -x --durations=8 > $O/${TAG}_tests.log 2>&1; tail -25 $O/${TAG}_tests.log
for v in "" "M2S_DEBUG=1 M2S_NO_WARM_BANDS=1" "M2S_DEBUG=1 M2S_NO_WARM=1"; do
  env $v python tools/first_call_probe.py 289 1024 6 | tee -a $O/${TAG}_first_call.jsonl | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['switches'], d['median'], 'warm', d['warm_ms'][1:3], 'upload', d['upload_ms'][1:3])"
done
