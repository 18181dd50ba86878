#!/bin/bash
# PMC passes for the bench workload: few counters per pass, hard timeout per pass (an over-subscribed
# counter set makes rocprofv3 abort and then hang in finalisation).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-pmcC}; shift
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum" "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_$i -o f -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-extra --no-c5 "$@" > $R/gpurun_out/${TAG}_$i.log 2>&1 || echo "pass $i failed/timeout: $set"
done
python $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_*/f_counter_collection.csv > $R/gpurun_out/${TAG}_summary.json
