#!/bin/bash
# HBM traffic of k_prepass per launch, separate --pmc passes per counter and per append order (MI355X_MICROARCH.md recipe)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for order in input arrival; do
  for set in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=pp_${order}_$(echo $set | tr ' ' '_')
    timeout 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$tag -o f -- python $R/tools/prepass_probe.py 289 1024 $order > $R/gpurun_out/$tag.log 2>&1 || echo "pass failed: $tag"
  done
  python $R/tools/pmc_summary.py $R/gpurun_out/pp_${order}_*/f_counter_collection.csv > $R/gpurun_out/pp_${order}_summary.json
  echo "== $order"; python -c "
import json; d=json.load(open('$R/gpurun_out/pp_${order}_summary.json')); print(json.dumps(d.get('m2s::k_prepass', d), indent=0))"
done
