#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sp3}
cd $R
M2S_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_sparse.py -q -s > $O/${TAG}_tests.log 2>&1; grep -E "m2s\]|passed|failed" $O/${TAG}_tests.log | sort | uniq -c | head -20
for v in solo duo; do
  M2S_LIB_PATH=$R/mesh2splat_amd/_build/sp_$v/libm2s_hip.so timeout 600 python -m pytest tests/test_gpu_sparse.py -q > $O/${TAG}_tests_$v.log 2>&1; tail -2 $O/${TAG}_tests_$v.log
  M2S_LIB_PATH=$R/mesh2splat_amd/_build/sp_$v/libm2s_hip.so timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover_$v.json > $O/${TAG}_crossover_$v.log 2>&1; echo $v; cut -c1-230 $O/${TAG}_crossover_$v.log
done
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
for v in solo duo; do
  M2S_LIB_PATH=$R/mesh2splat_amd/_build/sp_$v/libm2s_hip.so timeout 300 python tools/c5_full.py $O/${TAG}_c5_$v.json > $O/${TAG}_c5_$v.log 2>&1; echo $v; grep steady $O/${TAG}_c5_$v.log
done
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_async.py -q -x --durations=12 > $O/${TAG}_tests2.log 2>&1; tail -18 $O/${TAG}_tests2.log
