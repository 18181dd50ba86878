"""Generates the golden fixtures in this directory from the CPU oracle:

    python tests/golden/make_golden.py

The reference has no golden vectors for the conversion path and its GL implementation cannot run
here, so these freeze OUR pinned semantics: any later change of the oracle that
alters a record shows up as a diff of these files.  Inputs are rebuilt deterministically from
mesh2splat_amd.synth by the functions in SCENES.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from mesh2splat_amd import synth  # noqa: E402

SCENES = {
    "quad_R16": lambda: (synth.unit_quad(), 16),
    "sphere_n4_R32_tex16": lambda: (synth.cube_sphere(4, tex_size=16), 32),
    "soup40_R24_tex8": lambda: (synth.random_soup(40, seed=5, textures=synth.procedural_textures(8, 3)), 24),
    "grid2_n3_R40": lambda: (synth.sphere_grid(2, n=3, tex_size=8), 40),
}

if __name__ == "__main__":
    from oracle import oracle
    oracle.build()
    for name, mk in SCENES.items():
        scene, R = mk()
        total, rec, keys = oracle.convert(scene, R, cap=0, want_keys=True)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), total=np.int64(total), records=rec, keys=keys)
        print(name, total, rec.shape)
