#!/bin/bash
# compare N builds on the bench workload: tools/ab/abn.sh lib1 lib2 ...   (blocking steps, events on every launch)
for i in 1 2; do
for L in "$@"; do
  M2S_LIB_PATH=$L timeout 120 python bench.py --steps 48 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L'.split('/')[-2], round(d['value']/1e9,3),'B/s', round(d['ms_per_step'],4),'ms', {k:round(v,4) for k,v in d['kernel_ms'].items() if v>0})"
done; done
