#!/bin/bash
# round-2 final evidence: full GPU suite, smoke, rocprofv3 trace of the C3-only bench command and of the default command
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${1:-r2final}
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|error" > $O/${TAG}_tests.log
python __graft_entry__.py smoke > $O/${TAG}_smoke.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_c3only -o k -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-cold --no-extra-workloads --no-viewer-extra --no-overlap-extra > $O/${TAG}_c3only.json 2> $O/${TAG}_c3only.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_default -o k -- python $R/bench.py > $O/${TAG}_default.json 2> $O/${TAG}_default.err
