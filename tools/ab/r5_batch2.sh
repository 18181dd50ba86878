#!/bin/bash
# round 5, second batch: spin waits, full-strip staging at three workgroups per CU, s_setprio, and PMC passes over k_fused3
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b2}
cd $R; mkdir -p $O
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d.get('cold_path') or {}
print('$1'.ljust(30), 'step %.4f sync %.4f (median %.4f) kernel(ev) %s dedicated %.4f' % (d['ms_per_step'], d.get('sync_ms_per_step', 0), d['sync_ms_stats']['median'], {k:round(v,4) for k,v in d['kernel_ms'].items() if v}, (d.get('kernel_ms_dedicated') or {}).get('fused', 0)), d['config'].get('pipeline'))"; }
for rep in 1 2; do
  for V in "fused2:.:M2S_NO_LEAN=1" "lean4:.:" "lean4_nospin:.:M2S_NO_SPIN=1" "lean3s64:ab_l3s64:" "lean4prio:ab_prio:"; do
    IFS=: read name dir envs <<< "$V"
    [ -f $R/mesh2splat_amd/_build/$dir/libm2s_hip.so ] || continue
    env M2S_DEBUG=1 $envs M2S_LIB_PATH=$R/mesh2splat_amd/_build/$dir/libm2s_hip.so timeout 300 python bench.py --workload c3 --steps 100 --warmup 10 --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 --no-cold 2>$O/${TAG}_err.log | line "c3 $name" | tee -a $O/${TAG}.log
  done
done
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM" "FETCH_SIZE WRITE_SIZE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $O/${TAG}_pmc_$i -o f -- python $R/bench.py --workload c3 --steps 3 --warmup 1 --no-cpu-baseline --no-overlap-extra --no-c5 --no-extra-workloads --no-viewer-extra --no-cold > $O/${TAG}_pmc_$i.log 2>&1 || echo "pass $i failed/timeout: $set"
done
python $R/tools/pmc_summary.py $O/${TAG}_pmc_*/f_counter_collection.csv > $O/${TAG}_pmc_summary.json
python -c "
import json; d=json.load(open('$O/${TAG}_pmc_summary.json'))
for k,v in d.items():
    if 'fused' in k: print(k, json.dumps(v))" | tee -a $O/${TAG}.log
rm -rf $O/${TAG}_pmc_[0-9]
