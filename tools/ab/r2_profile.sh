#!/bin/bash
# round-2 evidence: rocprofv3 kernel traces of the default bench command and of the extra workloads, PMC traffic passes
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r2prof}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_default -o k -- python $R/bench.py --no-cpu-baseline > $O/${TAG}_default.log 2>&1
for w in c4 mid c2; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_$w -o k -- python $R/bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_$w.log 2>&1
done
for w in c3 c4; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_${w}_$i -o f -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_pmc_${w}_$i.log 2>&1 || echo "pass $i failed: $set"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${w}_*/f_counter_collection.csv > $O/${TAG}_pmc_${w}_summary.json
done
