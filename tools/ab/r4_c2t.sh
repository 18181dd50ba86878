#!/bin/bash
# per-workgroup cycle accounting of k_fused2 on the C2 stand-in and on C3 (timing build)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4c2t}
cd $R
export M2S_LIB_PATH=$R/mesh2splat_amd/_build/timing/libm2s_hip.so
echo "== c2" | tee $O/${TAG}.log; TT_DETAIL=1 TT_N=76 TT_R=512 python tools/team_timing.py 2>&1 | tee -a $O/${TAG}.log
