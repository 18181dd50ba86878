#!/bin/bash
# round 6: row walkers set up with reciprocal-based floor divisions (one reciprocal per edge, no divergent sides) — parity, then A/B
# against the builds before (mesh2splat_amd/_build_base: start of the session; _build_exact: with m2s_exact.h only).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_walker}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py tests/test_gpu_hetero.py tests/test_gpu_fullsize.py tests/test_gpu_edge.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
Q="--no-cpu-baseline --no-c5 --no-viewer-extra --no-cold --no-overlap-extra --no-end-to-end"
for i in 1 2 3; do
for L in ${LIBS:-_build_base _build_exact _build}; do
  M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python bench.py --steps 100 --warmup 10 $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; x=d['extra_workloads']
print('%-12s' % '$L', 'c3 ded %.4f |' % d['kernel_ms_dedicated']['fused'], ' | '.join('%s %s blk %.4f' % (w, '+'.join('%.4f' % v for v in (x[w].get('kernel_ms') or {}).values() if v), x[w].get('blocking_ms') or 0) for w in ('c2','band','mid','c4','hetero')))" | tee -a $O/ab.log
done; done
