#!/bin/bash
# round 5, eleventh batch: small triangles of the multi-pass pipeline through the coverage mask (no row walker): k_count_scan, k_emit2's row
# starts, emit_fine_block.  parity, then band / mid / C4 / heterogeneous scene against the commit before (_build/base)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b11}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_hetero.py tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_fuzz.py tests/test_gpu_edge.py tests/test_gpu_async.py -q -m gpu 2>&1 | tail -8 ) | tee $O/${TAG}_tests.log
( timeout 300 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3 or c4" 2>&1 | tail -3 ) | tee -a $O/${TAG}_tests.log
for rep in 1 2; do
for V in "extras_new:" "extras_base:M2S_LIB_PATH=$R/mesh2splat_amd/_build/base/libm2s_hip.so"; do
  IFS=: read name envs <<< "$V"
  env $envs timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-viewer-extra --no-cold --no-c5 --no-overlap-extra 2>>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for w,c in (d.get('extra_workloads') or {}).items():
    if w in ('mid','c4','hetero'): print('$name', w, {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ('blocking_ms','kernels_total_ms') or k.startswith('kernel_ms')}, round(c.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak',0),4))" | tee -a $O/${TAG}.log
done
done
grep -v amdgpu.ids $O/${TAG}_err.log | tail -5
