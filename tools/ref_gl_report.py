#!/usr/bin/env python
"""Measured differences between the oracle's pinned fixed-function semantics and a real OpenGL implementation (Mesa llvmpipe)
running the reference's own host code and unmodified conversion shaders (oracle/_ref/ref_gl_check).  Writes
profiles/r02/ref_gl_llvmpipe.json.  CPU only (no GPU): python tools/ref_gl_report.py [--c3]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import refgl  # noqa: E402
from mesh2splat_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def main():
    oracle.build()
    rep = {"what": "reference path (SceneManager::loadModel -> ConversionPass::execute -> converter{VS,GS,FS}.glsl, all unmodified) on Mesa "
                   "llvmpipe vs oracle/m2s_oracle.c; records matched by nearest position; deviations are absolute",
           "scenes": {}}
    tex64 = synth.procedural_textures(64)
    scenes = [("K-1 unit quad R=64", synth.unit_quad(), 64, {}),
              ("cube-sphere n=6, 64^2 maps, R=128 (magnified)", synth.cube_sphere(6, tex_size=64), 128, {}),
              ("cube-sphere n=6, 64^2 maps, R=40 (mip levels blended)", synth.cube_sphere(6, tex_size=64), 40, {}),
              ("cube-sphere n=6, R=40, llvmpipe fp32 sampler", synth.cube_sphere(6, tex_size=64), 40, {"float_sampler": True}),
              ("2x2x2 sphere grid (cumulative bbox), R=96", synth.sphere_grid(2, n=4, tex_size=16), 96, {}),
              ("300 random triangles R=200", synth.random_soup(300, seed=5), 200, {}),
              ("2000 small random triangles R=1000", synth.random_soup(2000, seed=11, tri_size=0.02), 1000, {}),
              ("textured quad 64^2 maps R=256 (K-7, lambda < 0), fp32 sampler", synth.unit_quad(tex64), 256, {"float_sampler": True}),
              ("textured quad 64^2 maps R=256 (K-7, lambda < 0), default sampler", synth.unit_quad(tex64), 256, {}),
              ("textured quad 64^2 maps R=24 (K-7, 0 < lambda < 4)", synth.unit_quad(tex64), 24, {}),
              ("textured quad 256^2 maps R=8 (K-7, lambda > 4: clamped to level 4)", synth.unit_quad(synth.procedural_textures(256)), 8, {})]
    for name, scene, R, kw in scenes:
        r = refgl.compare(scene, R, oracle, **kw)
        if r is None:
            print("no GL context on this machine")
            return 1
        rep["scenes"][name] = r
        print(name, "| count", r["gl_counter"], "vs", r["oracle_counter"], "| pixels only GL / only oracle", r["pixels_only_gl"], r["pixels_only_oracle"],
              "| colour max", r.get("color", {}).get("max_abs"))
    g = refgl.run(synth.unit_quad(tex64), 16, want_mips=True)
    chain, offs, n = oracle.build_mips(tex64["baseColorTexture"])
    rep["glGenerateMipmap_vs_pinned_box_filter"] = {}
    for l, m in enumerate(g["mips"]):
        o = chain[int(offs[l]): int(offs[l]) + m.shape[0] * m.shape[1]].reshape(m.shape)
        d = np.abs(m.astype(int) - o.astype(int))
        rep["glGenerateMipmap_vs_pinned_box_filter"][f"level{l}"] = {"size": list(m.shape[:2]), "max_byte_diff": int(d.max()), "fraction_of_bytes_differing": float((d > 0).mean())}
    if "--c3" in sys.argv:
        t0 = time.time()
        scene = synth.colocated_spheres(1, 289, 2048)
        total = oracle.convert(scene, 1024, cap=0, count_only=True)[0]
        g = refgl.run(scene, 1024)
        rep["C3 full workload (1 002 252 triangles, 3 x 2048^2 maps, R = 1024)"] = {
            "gl_counter": g["counter"], "oracle_counter": int(total), "coverage_fragments_gl": len(g["coverage"]),
            "execute_ms_llvmpipe": g["info"]["execute_ms"], "cores": os.cpu_count(), "wall_s_incl_coverage_pass": time.time() - t0}
        print("C3:", rep["C3 full workload (1 002 252 triangles, 3 x 2048^2 maps, R = 1024)"])
    out = os.path.join(ROOT, "profiles", "r02", "ref_gl_llvmpipe.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", out)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
