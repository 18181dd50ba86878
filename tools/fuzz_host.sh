#!/bin/bash
# Mutation fuzzing of the host-side file readers under ASan + UBSan (CPU only).  usage: tools/fuzz_host.sh [iterations] [jobs]
set -e
R=$(cd "$(dirname "$0")/.." && pwd); W=${TMPDIR:-/tmp}/m2s_fuzz; IT=${1:-20000}; J=${2:-8}
mkdir -p $W/seeds
C=$R/mesh2splat_amd/csrc
g++ -std=c++17 -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
    -o $W/fuzz_host $R/tools/fuzz_host.cpp $C/m2s_png.cpp $C/m2s_jpeg.cpp $C/m2s_gltf.cpp $C/m2s_host_api.cpp $C/m2s_ply.cpp -lpthread -lz
python3 $R/tools/fuzz_seeds.py $W/seeds
SEEDS="$W/seeds/*.png $W/seeds/*.jpg $R/tests/golden/ref_host/*.glb $R/tests/golden/ref_host/*.ply"
pids=()
for j in $(seq 1 $J); do
  ASAN_OPTIONS=detect_leaks=1:allocator_may_return_null=1:max_allocation_size_mb=2048 $W/fuzz_host $IT $((0x4D325300 + j)) $SEEDS > $W/out_$j.log 2>&1 &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
tail -n 3 $W/out_*.log
exit $rc
