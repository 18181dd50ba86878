#!/bin/bash
# round 5, eighth batch: bench.py's multi-rank path on mesh2splat_amd/ctl.py (stub transport, world 2/4/8, --dry-scale), the band
# test (team kernel in batches of 40), strided row walker in k_count_scan; extras
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b8}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_dist_stub.py tests/test_gpu_dist_local.py tests/test_gpu_hetero.py tests/test_gpu_round2.py -q -m gpu 2>&1 | tail -15 ) | tee $O/${TAG}_tests.log
timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 --no-cpu-baseline --no-viewer-extra --no-cold --no-c5 2>>$O/${TAG}_err.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for w,c in (d.get('extra_workloads') or {}).items():
    print('extras', w, {k:(round(v,4) if isinstance(v,float) else v) for k,v in c.items() if k in ('ms_per_step','blocking_ms','kernels_total_ms','pipeline','error','roofline_blocking') or k.startswith('kernel')}, round(c.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak',0),4))
print('workloads', d['roofline'].get('workloads'))" | tee -a $O/${TAG}.log
timeout 300 python bench.py --gpus 2 --dry-scale --workload c2 --steps 20 --warmup 3 --no-extra-workloads 2>>$O/${TAG}_err.log | tail -1 > $O/${TAG}_dry_scale_2.json; python -c "
import json; d=json.load(open('$O/${TAG}_dry_scale_2.json')); print('dry-scale 2:', d['scale_record']); print({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ('ms_per_step','value','merged_identical_on_all_ranks')}) for k, v in d.get('strong_scaling', {}).get('c3', {}).items() if k in ('gaussians','no_gather','gather')})" | tee -a $O/${TAG}.log
grep -v amdgpu.ids $O/${TAG}_err.log | tail -8
