#!/bin/bash
# round 6: per-workgroup base table for scenes of one generation of workgroups — parity, then A/B on the C2 stand-in against _build_base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_c2}; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py tests/test_gpu_async.py tests/test_gpu_round2.py -m gpu -q -x > $O/tests.log 2>&1; tail -3 $O/tests.log
LIBS="${LIBS:-_build_base _build}" PROBE="c2" bash tools/ab/r6_abq.sh ${1:-r6_c2}
for L in ${LIBS:-_build_base _build}; do M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 200 python bench.py --workload c2 --steps 100 --warmup 10 --no-cpu-baseline --no-c5 --no-viewer-extra --no-extra-workloads --no-overlap-extra --no-end-to-end 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$L c2 headline: ms_per_step %.4f kernel %s dedicated %s blocking frac %.3f first_call %.3f new_R %.3f' % (d['ms_per_step'], {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, round(d['kernel_ms_dedicated']['fused'],4), r['frac_blocking'], r['frac_first_call'], r['frac_new_R']))"; done
