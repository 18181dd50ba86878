#!/bin/bash
# batch mode of the CLI: 6 copies of a mid-size .glb, sequential single-file runs vs --batch
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; CLI=$R/mesh2splat_amd/_build/mesh2splat
D=$(mktemp -d /tmp/m2s_batch_XXXX); mkdir -p $D/in $D/out $D/out2
python - <<PY
import sys; sys.path.insert(0,"$R")
from mesh2splat_amd import gltf_io, synth
for i in range(6):
    gltf_io.write_glb(synth.cube_sphere(120 + 4*i, tex_size=1024, seed=100+i), "$D/in/m%d.glb" % i)
PY
s=$(date +%s.%N)
for i in 0 1 2 3 4 5; do $CLI $D/in/m$i.glb $D/out2/m$i.ply --density 1024 --format 2 > /dev/null; done
e=$(date +%s.%N); python3 -c "print(\"sequential single-file runs: %.3f s for 6 files\" % ($e - $s))"
s=$(date +%s.%N)
$CLI --batch $D/in --out $D/out --density 1024 --format 2
e=$(date +%s.%N); python3 -c "print(\"batch: %.3f s for 6 files\" % ($e - $s))"
for i in 0 1 2 3 4 5; do cmp $D/out/m$i.ply $D/out2/m$i.ply && echo "m$i identical"; done
rm -rf $D
