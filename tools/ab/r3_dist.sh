#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3dist}
cd $R
timeout 600 python -m pytest tests/test_gpu_dist_local.py tests/test_gpu_round2.py tests/test_gpu_cli.py tests/test_gpu_async.py -q -x > $O/${TAG}_tests.log 2>&1; tail -4 $O/${TAG}_tests.log
timeout 300 python bench.py --force-dist --no-cpu-baseline --no-viewer-extra --no-c5 --steps 30 > $O/${TAG}_bench_force_dist.json 2> $O/${TAG}_bench_force_dist.err; tail -3 $O/${TAG}_bench_force_dist.err
python -c "
import json; d=json.load(open('$O/${TAG}_bench_force_dist.json')); print(d['value'], d['ms_per_step'], d.get('multi_gpu_bringup'), list(d.get('strong_scaling',{}).keys()), d.get('gather',{}).get('ms_per_step'))"
M2S_BENCH_FORCE_TORCH_EXCHANGE=1 timeout 300 python bench.py --force-dist --no-cpu-baseline --no-viewer-extra --no-c5 --no-extra-workloads --steps 10 > $O/${TAG}_bench_fallback.json 2> $O/${TAG}_bench_fallback.err; tail -2 $O/${TAG}_bench_fallback.err
python -c "
import json; d=json.load(open('$O/${TAG}_bench_fallback.json')); print(d['value'], d['ms_per_step'], d.get('multi_gpu_bringup'))"
