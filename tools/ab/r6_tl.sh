#!/bin/bash
# round 6: per-workgroup timelines of the multi-pass kernels (tools/timeline_probe.py on the -DM2S_TIMELINE build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/${1:-r6_tl}; mkdir -p $O; cd $R
for w in ${WL:-hetero c4 mid}; do
  M2S_LIB_PATH=$R/mesh2splat_amd/_build_tl/libm2s_hip.so timeout 300 python tools/timeline_probe.py $w $O/tl_$w.json > $O/tl_$w.txt 2>&1; cat $O/tl_$w.txt | cut -c1-400
done
