#!/bin/bash
# what the first conversion of a PROCESS costs in the one-shot converter (config 3 as a .glb), with and without the code-object preload
TAG=${1:-r4cli}
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
python - <<PY > $O/${TAG}.log 2>&1
import os, subprocess, sys
sys.path.insert(0, "$R")
from mesh2splat_amd import gltf_io, synth
glb = "/tmp/c3_cli.glb"
gltf_io.write_glb(synth.cube_sphere(289, tex_size=2048), glb, indexed=False)
for name, env in (("preload", {}), ("no preload", {"M2S_DEBUG": "1", "M2S_NO_PRELOAD": "1"}), ("preload", {}), ("no preload", {"M2S_DEBUG": "1", "M2S_NO_PRELOAD": "1"})):
    for fmt in (2,):
        e = dict(os.environ); e.update(env)
        r = subprocess.run(["$R/mesh2splat_amd/_build/mesh2splat", glb, "/tmp/c3_cli.ply", "--density", "1024", "--format", str(fmt), "--timing"], capture_output=True, text=True, timeout=300, env=e)
        print(name, "|", r.stdout.strip().splitlines()[-1][:400], r.stderr[-300:])
PY
cat $O/${TAG}.log
