#!/bin/bash
# the multi-pass pipeline on two lanes: tests, then the bench's extra workloads with their overlapped throughput
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4mp2}
cd $R
timeout 600 python -m pytest tests/test_gpu_async.py -m gpu -q -x 2>&1 | tail -5
for rep in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-cold --no-viewer-extra 2>$O/${TAG}.err | tee $O/${TAG}_bench.json | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c3 step %.4f overlapped %.4f' % (d['ms_per_step'], d['overlapped']['ms_per_step']))
for k,v in d['extra_workloads'].items():
    print(k, v.get('pipeline'), 'step %.4f' % v['ms_per_step'], 'kernels', {a:round(b,4) for a,b in v['kernel_ms'].items()}, 'frac %.3f' % v['roofline_whole_conversion']['frac_of_hbm_peak'], 'overlapped', {a:(round(b,4) if isinstance(b,float) else b) for a,b in v['overlapped'].items() if a!='what'})
"
done
tail -3 $O/${TAG}.err
