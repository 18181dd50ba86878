#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sc}
cd $R
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_fullsize.py -q -x > $O/${TAG}_tests.log 2>&1; tail -3 $O/${TAG}_tests.log
timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover.json > $O/${TAG}_crossover.log 2>&1; cut -c1-230 $O/${TAG}_crossover.log
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
timeout 300 python tools/c5_full.py $O/${TAG}_c5.json > $O/${TAG}_c5.log 2>&1; grep steady $O/${TAG}_c5.log
TT_N=721 TT_R=1448 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 200 python tools/sparse_timing.py > $O/${TAG}_timing.log 2>&1; tail -22 $O/${TAG}_timing.log
timeout 300 python bench.py --no-cpu-baseline --no-viewer-extra > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; python -c "
import json; d=json.load(open('$O/${TAG}_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], {k:(v.get('ms_per_step'),v.get('roofline_frac')) for k,v in d.get('extra_workloads',{}).items()}, d.get('cold_path'))"
