#!/bin/bash
# round 6: instruction counts of k_count_scan / k_emit2 for two builds on the multi-pass workloads
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_pmc_count}; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for L in ${LIBS:-_build_base _build}; do
for w in ${WL:-c4 hetero}; do
  M2S_LIB_PATH=$R/mesh2splat_amd/$L/libm2s_hip.so timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/${L}_$w -o f -- python $R/tools/mp_probe.py $w > $O/${L}_$w.log 2>&1
  python $R/tools/pmc_summary.py $O/${L}_$w/f_counter_collection.csv > $O/${L}_$w.json
  python - <<PY
import json
d = json.load(open("$O/${L}_$w.json"))
for k, v in d.items():
    if "count_scan" in k or "emit2" in k: print("$L $w", k.split("(")[0][-14:], {c: round(x) for c, x in v.items()})
PY
done; done
