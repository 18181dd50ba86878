#!/bin/bash
# round-3 closing evidence, one box: the whole GPU suite, smoke(), the default bench line; rocprofv3 --kernel-trace --stats of the
# same bench command; separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ sets) for config 3 (k_fused2), config 5 at full size
# (k_sparse) and the C2 stand-in; sha256 of the library the counters were taken on
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3fin}
cd $R
sha256sum mesh2splat_amd/_build/libm2s_hip.so | cut -c1-16 > $O/${TAG}_binary_sha.txt; cat $O/${TAG}_binary_sha.txt
bash tools/ab/r3_check.sh ${TAG}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_bench -o k -- python $R/bench.py > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace_bench.err || echo "trace of bench failed"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_c3 -o k -- python $R/bench.py --no-overlap-extra --no-c5 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_trace_c3.json 2> $O/${TAG}_trace_c3.err || echo "trace of c3 failed"
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM")
for w in c3 c2; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_${w}_$i -o f -- python $R/bench.py --workload $w --steps 20 --warmup 3 --sync-steps --no-overlap-extra --no-c5 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_pmc_${w}_$i.log 2>&1 || echo "$w pass $i failed: $set"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${w}_*/f_counter_collection.csv > $O/${TAG}_pmc_${w}_summary.json
done
export C5_NO_ORACLE=1 C5_ITERS=6 C5_CACHE=1
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_c5_$i -o f -- python $R/tools/c5_full.py $O/${TAG}_c5_pmc$i.json > $O/${TAG}_pmc_c5_$i.log 2>&1 || echo "c5 pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_c5_*/f_counter_collection.csv > $O/${TAG}_pmc_c5_summary.json
for w in c3 c2 c5; do python - <<PY
import json
d = json.load(open("$O/${TAG}_pmc_${w}_summary.json"))
for k, v in d.items():
    if any(x in k for x in ("fused2", "sparse", "emit2", "count_scan")): print("$w", k, {c: round(x) for c, x in v.items()})
PY
done
for t in bench c3; do f=$(ls $O/${TAG}_trace_$t/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -8 $f; done
