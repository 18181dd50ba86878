#!/bin/bash
# round 3: the sparse kernel — parity tests, BASELINE config 5 at full size with and without it, density crossover
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sp}
cd $R
timeout 600 python -m pytest tests/test_gpu_sparse.py -x -q > $O/${TAG}_tests.log 2>&1; tail -15 $O/${TAG}_tests.log
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
timeout 300 python tools/c5_full.py $O/${TAG}_c5_sparse.json > $O/${TAG}_c5_sparse.log 2>&1; tail -3 $O/${TAG}_c5_sparse.log
M2S_DEBUG=1 M2S_NO_SPARSE=1 timeout 300 python tools/c5_full.py $O/${TAG}_c5_team.json > $O/${TAG}_c5_team.log 2>&1; tail -2 $O/${TAG}_c5_team.log
for v in $R/mesh2splat_amd/_build/sp_*; do
  [ -f $v/libm2s_hip.so ] || continue
  M2S_LIB_PATH=$v/libm2s_hip.so timeout 300 python tools/c5_full.py $O/${TAG}_c5_$(basename $v).json > $O/${TAG}_c5_$(basename $v).log 2>&1; echo $(basename $v); tail -2 $O/${TAG}_c5_$(basename $v).log
done
timeout 300 python tools/sparse_crossover.py $O/${TAG}_crossover.json > $O/${TAG}_crossover.log 2>&1; cat $O/${TAG}_crossover.log
