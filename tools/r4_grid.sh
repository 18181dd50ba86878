#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r4g}
cd $R
B="python bench.py --steps 40 --warmup 5 --no-cold --no-extra-workloads --no-cpu-baseline --no-viewer-extra --no-c5 --no-overlap-extra"
for g in 512 768 1024 1536; do
    M2S_DEBUG=1 M2S_PERSIST_GRID=$g $B 2>$O/${TAG}_err_$g.log | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('grid $g', 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused']))" | tee -a $O/${TAG}_grid.log
    grep "k_fused2p" $O/${TAG}_err_$g.log | head -1
done
M2S_DEBUG=1 M2S_NO_PERSIST=1 $B 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no persist', 'step %.4f sync %.4f kernel(ev) %.4f dedicated %.4f' % (d['ms_per_step'], d['sync_ms_per_step'], d['kernel_ms']['fused'], d['kernel_ms_dedicated']['fused']))" | tee -a $O/${TAG}_grid.log
