#!/bin/bash
# round-3 "before" evidence: PMC passes on BASELINE config 5 at full size (the sparse regime had none) and on the C2 stand-in,
# the team kernel's per-workgroup cycle accounting for C2 (-DM2S_TIMING build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3base}
cd /tmp && export TMPDIR=/tmp
export C5_NO_ORACLE=1 C5_ITERS=3 C5_CACHE=1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_c5_trace -o k -- python $R/tools/c5_full.py $O/${TAG}_c5.json > $O/${TAG}_c5_trace.log 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_c5_$i -o f -- python $R/tools/c5_full.py $O/${TAG}_c5_pmc$i.json > $O/${TAG}_pmc_c5_$i.log 2>&1 || echo "c5 pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_c5_*/f_counter_collection.csv > $O/${TAG}_pmc_c5_summary.json
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_c2_$i -o f -- python $R/bench.py --workload c2 --steps 3 --warmup 1 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_pmc_c2_$i.log 2>&1 || echo "c2 pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_c2_*/f_counter_collection.csv > $O/${TAG}_pmc_c2_summary.json
cd $R
if [ -f mesh2splat_amd/_build/timing/libm2s_hip.so ]; then
  TT_N=76 TT_R=512 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 120 python tools/team_timing.py > $O/${TAG}_team_timing_c2.log 2>&1
  TT_N=289 TT_R=1024 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 120 python tools/team_timing.py > $O/${TAG}_team_timing_c3.log 2>&1
fi
ls $O | head -50
