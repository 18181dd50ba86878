#!/bin/bash
# compare builds of k_prepass: tools/ab/prepass_ab.sh lib1 lib2 ...
for L in "$@"; do
  echo "== $L"; M2S_LIB_PATH=$L timeout 100 python tools/prepass_probe.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:(round(v['kernel_ms_median'],4), v['visible']) for k,v in d.items() if isinstance(v,dict)})"
done
