#!/bin/bash
# round 3: k_sparse with grouped sub-ranges (v2) against the one-range-per-workgroup form (v1, _build/sp_v1) on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sp5}
cd $R
M2S_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_sparse.py -q -s > $O/${TAG}_tests.log 2>&1; grep -E "m2s\]|passed|failed" $O/${TAG}_tests.log | sort | uniq -c | head -20
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
for v in main sp_v1 main sp_v1; do
  L=$R/mesh2splat_amd/_build/libm2s_hip.so; [ $v = main ] || L=$R/mesh2splat_amd/_build/$v/libm2s_hip.so
  echo "== $v"
  M2S_LIB_PATH=$L timeout 300 python tools/c5_full.py $O/${TAG}_c5_$v.json > $O/${TAG}_c5_$v.log 2>&1; grep steady $O/${TAG}_c5_$v.log
done
for v in main sp_v1; do
  L=$R/mesh2splat_amd/_build/libm2s_hip.so; [ $v = main ] || L=$R/mesh2splat_amd/_build/$v/libm2s_hip.so
  echo "== $v"
  M2S_LIB_PATH=$L timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover_$v.json > $O/${TAG}_crossover_$v.log 2>&1; cut -c1-230 $O/${TAG}_crossover_$v.log
done
