#!/bin/bash
# round-4 closing evidence, one box: the whole GPU suite, smoke(), the default bench line; rocprofv3 --kernel-trace --stats of the
# same bench command; separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ sets) for config 3 (k_fused2), config 5 at full size
# (k_sparse) and the C2 stand-in; sha256 of the library the counters were taken on
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4fin}
cd $R
sha256sum mesh2splat_amd/_build/libm2s_hip.so | cut -c1-16 > $O/${TAG}_binary_sha.txt; cat $O/${TAG}_binary_sha.txt
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -2 $O/${TAG}_bench.err; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('frac_dedicated_sample'), (d.get('overlapped') or {}).get('ms_per_step'), {k:(round(v.get('ms_per_step',0),4),v.get('kernel_ms'),v.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak')) for k,v in d.get('extra_workloads',{}).items()}); print({k: d['roofline'].get(k) for k in ('frac_step','frac_first_call','frac_new_R','frac_cold_inputs')}, d['extra_workloads'].get('c5',{}).get('depth_sort',{}).get('repeat_ms'))"
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

# the command line: what the first conversion of a model costs inside the one-shot converter (config 3 as a .glb)
cd $R
python - <<PY > $O/${TAG}_cli.log 2>&1
import os, subprocess, sys, json
sys.path.insert(0, "$R")
from mesh2splat_amd import gltf_io, synth
glb = "/tmp/c3_cli.glb"
gltf_io.write_glb(synth.cube_sphere(289, tex_size=2048), glb, indexed=False)
for fmt in (2, 1):
    r = subprocess.run(["$R/mesh2splat_amd/_build/mesh2splat", glb, "/tmp/c3_cli.ply", "--density", "1024", "--format", str(fmt), "--timing"], capture_output=True, text=True, timeout=300)
    print(r.stdout[-1500:], r.stderr[-300:])
PY
tail -12 $O/${TAG}_cli.log
# the GPU suite last (its CPU side — the oracle — can take several minutes on a busy box)
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed|^[0-9.]+s (call|setup)" $O/${TAG}_tests.log | head -12
