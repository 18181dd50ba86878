#!/bin/bash
# quick A/B of library builds on the multi-pass workloads: LIBS="_build_base _build ..." PROBE="c4 hetero" tools/ab/r6_abq.sh tag
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_abq}; mkdir -p $O; cd $R
for i in 1 2 3; do
for L in ${LIBS:-_build_base _build}; do
  echo "$L $(M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 300 python tools/mp_probe.py ${PROBE:-mid c4 hetero} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['workload'], '+'.join('%.4f' % v for v in d['kernel_ms'].values() if v), 'blk %.4f' % d['blocking_ms'], 'frac %.3f' % d['frac_kernels'], end=' | ')")" | tee -a $O/ab.log
done; done
