#!/bin/bash
# round-2 baseline: bench lines + kernel traces + PMC traffic for the mid-size-triangle workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r2base}
cd /tmp && export TMPDIR=/tmp
for w in c4 mid c2 c3; do
  python $R/bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra > $O/${TAG}_$w.json 2> $O/${TAG}_$w.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_prof_$w -o k -- python $R/bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra > $O/${TAG}_prof_$w.log 2>&1
done
for w in c4 mid; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_${w}_$i -o f -- python $R/bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-viewer-extra > $O/${TAG}_pmc_${w}_$i.log 2>&1 || echo "pass $i failed: $set"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${w}_*/f_counter_collection.csv > $O/${TAG}_pmc_${w}_summary.json
done
ls $O | head -50
