#!/usr/bin/env python
"""Debug probe: multi-pass kernel times on the C4 stand-in at several resolutions, and on one big mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
for name, scene in [("c4 64 meshes", synth.sphere_grid(4, n=18, tex_size=256)), ("one mesh n=204", synth.cube_sphere(204, tex_size=256))]:
    c = Converter(0); c.set_pipeline("multipass"); c.upload_scene(scene); c.set_max_gaussians(0)
    for R in (64, 256, 1024, 2048):
        c.convert(R); c.set_profiling(True); n = c.convert(R); ms = c.last_kernel_ms(); c.set_profiling(False)
        print(name, scene.n_triangles, "R", R, "frags", n, {k: round(v, 4) for k, v in ms.items() if v})
    c.close()
