#!/bin/bash
# where k_fused2's instructions are: PMC instruction counts of the full kernel and of two truncated builds
#   probe1: no strips (triangle phase + TriShade + expansion)      probe2: triangle phase only (load, GS setup, raster setup, coverage, scans, look-back)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4valu}
cd /tmp && export TMPDIR=/tmp
for w in c3 c2; do
for v in . probe1 probe2; do
  M2S_LIB_PATH=$R/mesh2splat_amd/_build/$v/libm2s_hip.so timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/${TAG}_${w}_$v -o f -- python $R/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-overlap-extra --no-c5 --no-cold --no-extra-workloads --no-viewer-extra --sync-steps > $O/${TAG}_${w}_$v.log 2>&1 || echo "failed $w $v"
  python $R/tools/pmc_summary.py $O/${TAG}_${w}_$v/f_counter_collection.csv | python -c "
import json,sys; d=json.load(sys.stdin).get('m2s::k_fused2',{}); print('$w $v'.ljust(12), {k:round(x) for k,x in d.items()})" | tee -a $O/${TAG}.log
done
done
