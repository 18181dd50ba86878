#!/bin/bash
# round 5, ninth batch: persistent k_count_scan for scenes of more than 1024 blocks; the 11-18 band under auto / multipass / team
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r5b9}
cd $R; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_hetero.py tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_async.py tests/test_gpu_edge.py -q -m gpu 2>&1 | tail -12 ) | tee $O/${TAG}_tests.log
timeout 300 python tools/band_probe.py 2>>$O/${TAG}_err.log | tee $O/${TAG}_band.jsonl
timeout 300 python tools/hetero_probe.py --settings auto --no-oracle 2>>$O/${TAG}_err.log | tee $O/${TAG}_hetero.jsonl | cut -c1-420
timeout 300 python tools/hetero_probe.py --settings auto --no-oracle --R 2048 2>>$O/${TAG}_err.log | tee -a $O/${TAG}_hetero.jsonl | cut -c1-420
grep -v amdgpu.ids $O/${TAG}_err.log | tail -8
