#!/bin/bash
# round 6: non-temporal stores of k_prepass's quads / depths — tests, then viewer passes A/B (bench.py's viewer section) against _build_base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_prepass_nt}; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_gpu_prepass.py -m gpu -q -x > $O/tests.log 2>&1; tail -2 $O/tests.log
for i in 1 2; do for L in ${LIBS:-_build_base _build}; do
  M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-c5 --no-extra-workloads --no-overlap-extra --no-end-to-end --no-cold 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['viewer_passes']; print('$L prepass input %.4f arrival %.4f sort_prepass %.4f frame two calls %.4f fused %.4f' % (v['prepass_input_order']['kernel_ms'], v['prepass_arrival_order']['kernel_ms'], v['sort_prepass_ms'], v['frame']['two_calls_ms'], v['frame']['fused_ms']))" | tee -a $O/ab.log
done; done
