#!/bin/bash
# PMC passes over the multi-pass pipeline on the mid-size mesh (k_count_scan, k_emit2): traffic, VALU, waits
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r4pmcmid}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_WAVES" "TCP_PENDING_STALL_CYCLES TD_TD_BUSY TA_TA_BUSY TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$i -o f -- python $R/bench.py --workload mid --steps 20 --warmup 2 --no-cpu-baseline --no-overlap-extra --no-c5 --no-cold --no-extra-workloads --no-viewer-extra --sync-steps > $R/gpurun_out/${TAG}_$i.log 2>&1 || echo "pass $i failed/timeout: $set"
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_*/f_counter_collection.csv > $R/gpurun_out/${TAG}_summary.json
python -c "
import json; d=json.load(open('$R/gpurun_out/${TAG}_summary.json'))
for k,v in d.items():
    if 'count_scan' in k or 'emit2' in k: print(k, json.dumps({a:round(b) for a,b in v.items()}))"
