#!/bin/bash
# round-6 closing evidence, one box.  part 1 (default): sha of the library, smoke(), the default bench line, rocprofv3 --kernel-trace --stats of
# the same bench command and of config 3 alone, the whole GPU suite.  part 2 (`pmc`): separate --pmc passes (FETCH_SIZE / WRITE_SIZE / two
# SQ sets) for config 3 (k_fused3), the C2 stand-in (k_fused2), the heterogeneous scene (k_count_scan + k_emit2) and config 5 (k_sparse).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r6fin}; PART=${2:-main}
cd $R; mkdir -p $O
sha256sum mesh2splat_amd/_build/libm2s_hip.so | cut -c1-16 > $O/${TAG}_binary_sha.txt; cat $O/${TAG}_binary_sha.txt
if [ "$PART" = main ]; then
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 500 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -2 $O/${TAG}_bench.err; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); r=d['roofline']
print('value %.4g ms_per_step %.4f sync %.4f kernel %s dedicated %s' % (d['value'], d['ms_per_step'], d['sync_ms_per_step'], {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, {k:round(v,4) for k,v in d['kernel_ms_dedicated'].items() if isinstance(v,float) and v}))
print({k: (round(r[k],4) if isinstance(r.get(k),float) else r.get(k)) for k in ('frac','frac_dedicated_sample','frac_step','frac_blocking','frac_first_call','frac_first_call_including_warm','frac_new_R','frac_cold_inputs','write_only_frac')})
print({k: round(v,4) for k,v in r['workloads'].items()})
for w,c in d['extra_workloads'].items(): print(w, c.get('pipeline'), round(c.get('ms_per_step',0),4), round(c.get('blocking_ms',0),4) if c.get('blocking_ms') else None, c.get('kernel_ms') or c.get('kernels_total_ms'), (c.get('depth_sort') or {}).get('repeat_ms'))
print('cpu', d.get('cpu_baseline'))"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_bench -o k -- python $R/bench.py > $O/${TAG}_trace_bench.json 2> $O/${TAG}_trace_bench.err || echo "trace of bench failed"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_trace_c3 -o k -- python $R/bench.py --no-overlap-extra --no-c5 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_trace_c3.json 2> $O/${TAG}_trace_c3.err || echo "trace of c3 failed"
for t in bench c3; do f=$(ls $O/${TAG}_trace_$t/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -9 $f | cut -c1-170; done
cd $R
# the driver's launch line for N > 1 (torchrun sets RANK / WORLD_SIZE / MASTER_*), here with both ranks on this one GPU through the RCCL stand-in
D=$(mktemp -d); mkdir -p $D/objs
M2S_RCCL_PATH=$R/tests/stub_rccl/_build/librccl_stub.so M2S_STUB_RCCL_DIR=$D/objs M2S_STUB_RCCL_LOG=$D/log M2S_STUB_RCCL_TIMEOUT=120 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --one-device --workload c2 --steps 5 --warmup 1 --no-extra-workloads --no-strong-scaling > $O/${TAG}_torchrun2.json 2> $O/${TAG}_torchrun2.err; python -c "
import json; d=json.loads([l for l in open('$O/${TAG}_torchrun2.json') if l.startswith('{')][-1]); print('torchrun x2 (stand-in):', d['n_gpus'], d['exchange_transport'][:20], d['scale_record']['per_rank_gaussians'], round(d['scale_record']['bringup_ms'],1), d['multi_gpu_bringup']['errors'])" || tail -5 $O/${TAG}_torchrun2.err
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed|^[0-9.]+s (call|setup)" $O/${TAG}_tests.log | head -10
else
cd /tmp && export TMPDIR=/tmp
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM")
for w in c3 c2 hetero; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_${w}_$i -o f -- python $R/bench.py --workload $w --steps 20 --warmup 3 --sync-steps --no-overlap-extra --no-c5 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads > $O/${TAG}_pmc_${w}_$i.log 2>&1 || echo "$w pass $i failed: $set"
  done
  python $R/tools/pmc_summary.py $O/${TAG}_pmc_${w}_*/f_counter_collection.csv > $O/${TAG}_pmc_${w}_summary.json
done
export C5_NO_ORACLE=1 C5_ITERS=6 C5_CACHE=1
i=0
for set in "${SETS[@]:0:3}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_c5_$i -o f -- python $R/tools/c5_full.py $O/${TAG}_c5_pmc$i.json > $O/${TAG}_pmc_c5_$i.log 2>&1 || echo "c5 pass $i failed: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_c5_*/f_counter_collection.csv > $O/${TAG}_pmc_c5_summary.json
for w in c3 c2 hetero c5; do python - <<PY
import json
d = json.load(open("$O/${TAG}_pmc_${w}_summary.json"))
for k, v in d.items():
    if any(x in k for x in ("fused", "sparse", "emit2", "count_scan")): print("$w", k, {c: round(x) for c, x in v.items()})
PY
done
fi
