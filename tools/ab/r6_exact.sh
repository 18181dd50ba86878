#!/bin/bash
# round 6: short IEEE sequences (m2s_exact.h) — on-device exhaustion, parity, then A/B of the library against the build before them
# (mesh2splat_amd/_build_base: `make OUT=../_build_base` on the parent commit).  tools/ab/r6_exact.sh [tag] [full]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_exact}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_exact_math.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py tests/test_gpu_hetero.py tests/test_gpu_fullsize.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
if [ "$2" = full ]; then (./tests/exact_math/_build/exact_math_check rcp; ./tests/exact_math/_build/exact_math_check sqrt; ./tests/exact_math/_build/exact_math_check divall 600) > $O/exact_math_exhaustive.jsonl 2>&1; cat $O/exact_math_exhaustive.jsonl | cut -c1-250; fi
Q="--no-cpu-baseline --no-c5 --no-viewer-extra --no-cold --no-overlap-extra --no-end-to-end"
for i in 1 2 3; do
for L in mesh2splat_amd/_build_base/libm2s_hip.so mesh2splat_amd/_build/libm2s_hip.so; do
  M2S_LIB_PATH=$R/$L timeout 300 python bench.py --steps 200 --warmup 20 $Q 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$L'.split('/')[1], 'step %.4f sync %.4f kernel %.4f dedicated %.4f' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused']), {k: round(v,4) for k,v in r['workloads'].items() if not k.endswith('_blocking')})" | tee -a $O/ab.log
done; done
