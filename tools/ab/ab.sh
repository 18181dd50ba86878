#!/bin/bash
# A/B two builds of the library on the bench workload: tools/ab/ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
for i in 1 2; do
for L in $A $B; do
  M2S_LIB_PATH=$L python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', round(d['value']/1e9,3),'B/s', round(d['ms_per_step'],4),'ms', {k:round(v,4) for k,v in d['kernel_ms'].items() if v>0})"
done; done
