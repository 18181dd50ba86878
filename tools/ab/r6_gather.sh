#!/bin/bash
# round 6: non-temporal stores in k_gather_records — depth sort of C5's records, A/B against _build_base (which has plain stores)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_gather}; mkdir -p $O; cd $R
timeout 300 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_parity.py -m gpu -q -x -k "sort" > $O/tests.log 2>&1; tail -2 $O/tests.log
export C5_NO_ORACLE=1 C5_ITERS=6 C5_CACHE=1
for i in 1 2; do for L in _build_base _build; do
  M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python tools/c5_full.py $O/c5_$L.json 2>&1 | grep -i "depth sort" | sed "s/^/$L /" | tee -a $O/ab.log
done; done
