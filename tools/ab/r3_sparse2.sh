#!/bin/bash
# round 3: sparse kernel — tests again, cycle accounting, PMC on config 5
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3sp2}
cd $R
timeout 600 python -m pytest tests/test_gpu_sparse.py -q > $O/${TAG}_tests.log 2>&1; tail -5 $O/${TAG}_tests.log
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_async.py -q -x > $O/${TAG}_tests2.log 2>&1; tail -3 $O/${TAG}_tests2.log
for nr in "721 1448" "721 512" "1021 2048"; do set -- $nr
  TT_N=$1 TT_R=$2 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 200 python tools/sparse_timing.py > $O/${TAG}_timing_$1_$2.log 2>&1; cat $O/${TAG}_timing_$1_$2.log
done
cd /tmp && export TMPDIR=/tmp
export C5_NO_ORACLE=1 C5_ITERS=3 C5_CACHE=1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_c5_$i -o f -- python $R/tools/c5_full.py $O/${TAG}_c5_pmc$i.json > $O/${TAG}_pmc_c5_$i.log 2>&1 || echo "c5 pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_c5_*/f_counter_collection.csv > $O/${TAG}_pmc_c5_summary.json
python -c "
import json; d=json.load(open('$O/${TAG}_pmc_c5_summary.json'))
for k,v in d.items():
    if 'sparse' in k or 'fused' in k: print(k, json.dumps(v))"
