#!/bin/bash
# round 6: TriSetup as planes + small slices over the tail of k_emit2's output — parity, timelines per tail size, A/B against _build_base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_tail}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hetero.py tests/test_gpu_edge.py tests/test_gpu_fuzz.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
for tail in ${TAILS:-0 524288 1048576 2097152 4194304}; do
for w in ${WL:-hetero c4 mid}; do
  M2S_EMIT2_TAIL=$tail M2S_LIB_PATH=$R/mesh2splat_amd/_build_tl/libm2s_hip.so timeout 300 python tools/timeline_probe.py $w > $O/tl_${w}_$tail.txt 2>&1
  echo "tail $tail $w: $(grep -E '^\{' $O/tl_${w}_$tail.txt | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k: round(v,4) for k,v in d['kernel_ms'].items() if v})") $(grep -E 'k_count_scan: span|k_emit2:' $O/tl_${w}_$tail.txt | cut -c1-140 | tr '\n' ' ')"
done; done
grep -E "waves alive" $O/tl_hetero_1048576.txt | cut -c1-600
for i in 1 2 3; do
for L in _build_base _build; do
  echo "$L $(M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python tools/mp_probe.py mid c4 hetero 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'], '+'.join('%.4f' % v for v in d['kernel_ms'].values() if v), 'blk %.4f' % d['blocking_ms'], 'frac %.3f' % d['frac_kernels'], end=' | ')")" | tee -a $O/ab.log
done; done
