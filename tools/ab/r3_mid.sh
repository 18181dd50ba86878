#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3mid}
cd $R
for v in "" midattr "" midattr; do
  L="X=1"; [ -n "$v" ] && L="M2S_LIB_PATH=$R/mesh2splat_amd/_build/$v/libm2s_hip.so"
  echo "== [$v]"
  env $L timeout 300 python tools/pipe_ab.py 289:2048:1024,76:2048:512,721:2048:1448 team,sparse 2>/dev/null
  for w in c4 mid; do env $L timeout 200 python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads --no-overlap-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v] $w', round(d['ms_per_step'],4), {k:round(x,4) for k,x in d['kernel_ms_dedicated'].items() if isinstance(x,float) and x>0})"; done
done 2>&1 | tee $O/${TAG}.log
