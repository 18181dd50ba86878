#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r3chk}
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $O/${TAG}_tests.log 2>&1; grep -E "passed|failed|^[0-9.]+s (call|setup)" $O/${TAG}_tests.log | head -16
python __graft_entry__.py smoke 2>&1 | tail -1
timeout 400 python bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; python -c "
import json; d=json.loads(open('$O/${TAG}_bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_binary_sha'), (d.get('overlapped') or {}).get('ms_per_step'), {k:(round(v.get('ms_per_step',0),4),v.get('kernel_ms'),v.get('roofline_whole_conversion',{}).get('frac_of_hbm_peak')) for k,v in d.get('extra_workloads',{}).items()}, d['cpu_baseline'].get('value'), d['cpu_baseline'].get('kind'))"
