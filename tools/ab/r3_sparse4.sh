#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sp4}
cd $R
M2S_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_sparse.py -q -s > $O/${TAG}_tests.log 2>&1; grep -E "m2s\]|passed|failed" $O/${TAG}_tests.log | sort | uniq -c | head -20
timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover.json > $O/${TAG}_crossover.log 2>&1; cut -c1-230 $O/${TAG}_crossover.log
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
timeout 300 python tools/c5_full.py $O/${TAG}_c5.json > $O/${TAG}_c5.log 2>&1; grep steady $O/${TAG}_c5.log
for v in $R/mesh2splat_amd/_build/sp_*; do
  [ -f $v/libm2s_hip.so ] || continue
  M2S_LIB_PATH=$v/libm2s_hip.so timeout 300 python tools/c5_full.py $O/${TAG}_c5_$(basename $v).json > $O/${TAG}_c5_$(basename $v).log 2>&1; echo $(basename $v); grep steady $O/${TAG}_c5_$(basename $v).log
  M2S_LIB_PATH=$v/libm2s_hip.so timeout 300 python tools/sparse_crossover.py > $O/${TAG}_crossover_$(basename $v).log 2>&1; cut -c1-230 $O/${TAG}_crossover_$(basename $v).log
done
TT_N=721 TT_R=1448 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 200 python tools/sparse_timing.py > $O/${TAG}_timing.log 2>&1; cat $O/${TAG}_timing.log
