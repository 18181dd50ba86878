#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3ab2}
cd $R
CASES="289:2048:1024,200:2048:1024,400:2048:1024,289:2048:724"
echo default; timeout 300 python tools/pipe_ab.py $CASES team,sparse 2>/dev/null | tee $O/${TAG}_default.log
for v in sp_nocull sp_nocull32; do echo $v; M2S_LIB_PATH=$R/mesh2splat_amd/_build/$v/libm2s_hip.so timeout 300 python tools/pipe_ab.py $CASES sparse 2>/dev/null | tee $O/${TAG}_$v.log; done
for v in "" cnt3; do
  L="X=1"; [ -n "$v" ] && L="M2S_LIB_PATH=$R/mesh2splat_amd/_build/$v/libm2s_hip.so"
  for w in c4 mid; do env $L timeout 200 python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads --no-overlap-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] $w', round(d['ms_per_step'],4), d['kernel_ms_dedicated'])"; done
done
