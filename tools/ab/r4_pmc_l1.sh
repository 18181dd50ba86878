#!/bin/bash
# what saturates in k_fused2's steady state?  texture-addresser / vector-L1 (TA, TCP, TD) busy and stall counters, config 3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4pmc}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TD|TCC|SQ|SQC|GRBM)_[A-Za-z0-9_]+" | sort -u > $O/${TAG}_counters.txt; wc -l $O/${TAG}_counters.txt
grep -E "^(TA_|TCP_|TD_)" $O/${TAG}_counters.txt | tr '\n' ' ' | head -c 6000; echo
SETS=("TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum TA_FLAT_WAVEFRONTS_sum" "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_$i -o f -- python $R/bench.py --steps 20 --warmup 3 --sync-steps --no-overlap-extra --no-c5 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_$i.log 2>&1 || echo "pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_*/f_counter_collection.csv > $O/${TAG}_summary.json 2>/dev/null
python - <<PY
import json
d = json.load(open("$O/${TAG}_summary.json"))
for k, v in d.items():
    if "fused2" in k: print(k, json.dumps({c: round(x) for c, x in v.items()}, indent=0))
PY
