#!/bin/bash
# round 5, sixth batch: fine blocks (per-256-triangle choice between the output-partitioned and a triangle-partitioned emitter inside
# k_emit2's launch): parity of everything that runs the multi-pass pipeline, then mid / C4 / the heterogeneous scene against _build/base
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b6}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_hetero.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py tests/test_gpu_async.py tests/test_gpu_sparse.py -q -m gpu 2>&1 | tail -12 ) | tee $O/${TAG}_tests.log
( timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3 or c4" 2>&1 | tail -4 ) | tee -a $O/${TAG}_tests.log
for V in "extras_new:" "extras_base:M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so"; do
  IFS=: read name envs <<< "$V"
  env $envs timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-viewer-extra --no-cold --no-c5 2>>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for w,c in (d.get('extra_workloads') or {}).items():
    print('$name', w, {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ('ms_per_step','blocking_ms','kernels_total_ms','pipeline','error','roofline_blocking') or k.startswith('kernel')}, round(c.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak',0),4))" | tee -a $O/${TAG}.log
done
timeout 300 python tools/hetero_probe.py --settings auto,multipass 2>>$O/${TAG}_err.log | tee $O/${TAG}_hetero.jsonl | cut -c1-420
for RR in 512 2048; do timeout 300 python tools/hetero_probe.py --settings auto,multipass,team --R $RR --no-oracle 2>>$O/${TAG}_err.log | tee -a $O/${TAG}_hetero.jsonl | cut -c1-420; done
grep -v amdgpu.ids $O/${TAG}_err.log | tail -5
