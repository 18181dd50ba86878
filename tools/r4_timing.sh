#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4t}
cd $R
export M2S_LIB_PATH=mesh2splat_amd/_build/timing/libm2s_hip.so
echo "== persistent" | tee $O/${TAG}_timing.log; python tools/team_timing.py 2>&1 | tee -a $O/${TAG}_timing.log
echo "== one unit per workgroup" | tee -a $O/${TAG}_timing.log; M2S_DEBUG=1 M2S_NO_PERSIST=1 python tools/team_timing.py 2>&1 | tee -a $O/${TAG}_timing.log
