#!/bin/bash
# what makes the first conversion after an upload slower than the second: another context converting in between (FCP_PRE)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4pre}
cd $R
for rep in 1 2; do
  for v in "" "FCP_PRE=other" "FCP_PRE=same"; do
    env $v python tools/first_call_probe.py 289 1024 6 | tee -a $O/${TAG}.jsonl | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v'.ljust(30), d['median'], 'first', d['first_call_ms'])"
  done
done
