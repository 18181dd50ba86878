#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3tex}
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_sparse.py tests/test_gpu_fullsize.py -q -x > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed" $O/${TAG}_tests.log
timeout 300 python tools/pipe_ab.py 289:2048:1024,400:2048:1024,721:2048:1448 team,sparse 2>/dev/null | tee $O/${TAG}_ab.log
timeout 300 python bench.py --no-cpu-baseline --no-viewer-extra --no-c5 2>/dev/null > $O/${TAG}_bench.json; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['kernel_ms_dedicated'], (d.get('overlapped') or {}).get('ms_per_step'), {k:(v.get('ms_per_step'),v.get('kernel_ms')) for k,v in d.get('extra_workloads',{}).items()}, d['cold_path']['cold_inputs'])"
for w in c4 mid; do M2S_LIB_PATH=$R/mesh2splat_amd/_build/emit4/libm2s_hip.so timeout 200 python bench.py --workload $w --steps 40 --warmup 4 --no-cpu-baseline --no-viewer-extra --no-cold --no-extra-workloads --no-overlap-extra 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[emit2 at 4 waves] $w', round(d['ms_per_step'],4), d['kernel_ms_dedicated'])"; done
export C5_NO_ORACLE=1 C5_ITERS=5 C5_CACHE=1
timeout 300 python tools/c5_full.py $O/${TAG}_c5.json > $O/${TAG}_c5.log 2>&1; grep steady $O/${TAG}_c5.log
TT_N=289 TT_R=1024 M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so timeout 120 python tools/team_timing.py 2>/dev/null | tee $O/${TAG}_team_timing_c3.log
