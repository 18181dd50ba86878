#!/bin/bash
# mid-size triangles: the team kernel with smaller batches (more, shorter workgroups) against the multi-pass pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4teammid}
cd $R
for L in . ab_s6k ab_s12k; do
  echo "== $L" | tee -a $O/${TAG}.log
  M2S_LIB_PATH=$R/mesh2splat_amd/_build/$L/libm2s_hip.so python tools/pipe_ab.py 94:2048:1024,76:2048:1024,140:2048:1024,200:2048:1024,289:2048:1024 team,multipass,auto 2>&1 | grep -v amdgpu.ids | tee -a $O/${TAG}.log
done
