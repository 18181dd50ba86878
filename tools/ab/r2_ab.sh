#!/bin/bash
# round-2 A/B: bench workloads under env variants.  usage: tools/ab/r2_ab.sh TAG "ENV=.. ENV2=.." "..." -- workloads...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=$1; shift
VARS=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do VARS+=("$1"); shift; done; shift
for w in "$@"; do for rep in 1 2; do for v in "${VARS[@]}"; do
  env $v python $R/bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra --no-cold 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1])
    print('$w','[$v]', round(d['value']/1e9,2),'B/s', round(d['ms_per_step'],4),'ms', {k:round(x,4) for k,x in d['kernel_ms'].items() if x>0}, d['roofline']['kernel'])
except Exception as e: print('$w','[$v]','FAILED',e)
" | tee -a $O/${TAG}.log
done; done; done
