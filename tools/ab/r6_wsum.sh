#!/bin/bash
# round 6: wave_sum through DPP (the tall-triangle loop of k_count_scan) — parity, timeline, A/B against _build_base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_wsum}; mkdir -p $O; cd $R
timeout 900 python -m pytest ${TESTS:-tests/test_gpu_parity.py tests/test_gpu_hetero.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py} -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for w in ${WL:-hetero c4 mid}; do
  M2S_LIB_PATH=$R/mesh2splat_amd/_build_tl/libm2s_hip.so timeout 300 python tools/timeline_probe.py $w $O/tl_$w.json > $O/tl_$w.txt 2>&1; head -${TLHEAD:-22} $O/tl_$w.txt | cut -c1-330
done
for i in 1 2 3; do
for L in ${LIBS:-_build_base _build}; do
  echo "$L $(M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python tools/mp_probe.py ${PROBE:-mid c4 hetero} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'], '+'.join('%.4f' % v for v in d['kernel_ms'].values() if v), 'blk %.4f' % d['blocking_ms'], 'frac %.3f' % d['frac_kernels'], end=' | ')")" | tee -a $O/ab.log
done; done
