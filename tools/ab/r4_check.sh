#!/bin/bash
# round 4: the whole GPU suite, smoke(), the default bench line, on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4chk}
cd $R
sha256sum mesh2splat_amd/_build/libm2s_hip.so | cut -c1-16 > $O/${TAG}_binary_sha.txt; cat $O/${TAG}_binary_sha.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=12 > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed|Error|^[0-9.]+s (call|setup)" $O/${TAG}_tests.log | head -24; grep -B5 -A40 "^___" $O/${TAG}_tests.log | head -120
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; tail -3 $O/${TAG}_bench.err; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], (d.get('overlapped') or {}).get('ms_per_step'), {k:(round(v.get('ms_per_step',0),4),v.get('kernel_ms'),v.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak')) for k,v in d.get('extra_workloads',{}).items()}); print(json.dumps(d.get('cold_path'), indent=0)[:1500])"
