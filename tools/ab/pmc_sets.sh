#!/bin/bash
# generic PMC runner: tools/ab/pmc_sets.sh TAG "set1 counters" "set2 counters" ...   (hard timeout per pass)
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$i -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-extra --no-c5 > $R/gpurun_out/${TAG}_$i.log 2>&1 || echo "pass $i failed/timeout: $set"
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_*/f_counter_collection.csv > $R/gpurun_out/${TAG}_summary.json
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_summary.json'))
for k,v in d.items():
    if 'fused' in k or 'emit' in k: print(k, json.dumps(v, indent=0))"
