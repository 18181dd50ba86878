#!/bin/bash
# round 3: XCD bands of equal estimated work (k_pick_bands) — tests, timeline of config 3, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3bands}
cd $R
timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_async.py tests/test_gpu_fullsize.py -q -x > $O/${TAG}_tests.log 2>&1; tail -3 $O/${TAG}_tests.log
M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 200 python tools/team_timing.py > $O/${TAG}_timeline_c3.log 2>&1; cat $O/${TAG}_timeline_c3.log
timeout 600 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; cat $O/${TAG}_bench.json
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
timeout 300 python tools/c5_full.py $O/${TAG}_c5.json > $O/${TAG}_c5.log 2>&1; grep steady $O/${TAG}_c5.log
timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover.json > $O/${TAG}_crossover.log 2>&1; cut -c1-230 $O/${TAG}_crossover.log
