#!/usr/bin/env python
"""Where does the 2.6e-2 between the oracle's pinned minification filter and Mesa llvmpipe come from?  

The reference's loader, ConversionPass::execute and UNMODIFIED shaders on llvmpipe (oracle/_ref/ref_gl_check) against the oracle,
on a 2 x 2 of
    textures : hash noise (as everywhere else)   |  the same maps with every byte rounded to a multiple of 64 — every 2x2 average
               through level 3 is then an exact integer, so glGenerateMipmap's rounding cannot differ from the pinned rule
    sampler  : llvmpipe's default (RGBA8 texels blended with 8-bit weights, one LOD per 2x2 quad)
               | GALLIVM_PERF=no_aos_sampling,no_quad_lod (fp32 filter, per-pixel LOD)
If the tie-free / fp32 cell agrees to ~1e-6, the pinned LOD formula and trilinear blend ARE what a real GL computes wherever GL is
deterministic, and the 2.6e-2 decomposes into mip rounding + 8-bit weights.  Writes profiles/r03/ref_gl_decomposition.json.
CPU only: python tools/ref_gl_decompose.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import refgl  # noqa: E402
from mesh2splat_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def tie_free(tex: dict) -> dict:
    """every byte -> the nearest lower multiple of 64 (alpha stays 255 -> 192: also a multiple of 64)"""
    return {k: (v & np.uint8(0xC0)) for k, v in tex.items()}


def scenes(tex):
    q = synth.unit_quad(tex)
    s = synth.cube_sphere(6, tex_size=0)
    s.meshes[0].textures = dict(tex)
    yield "textured quad 64^2 maps, R=24 (lambda 1.4: levels 1+2)", q, 24
    yield "textured quad 64^2 maps, R=12 (lambda 2.4: levels 2+3)", q, 12
    yield "cube-sphere n=6, 64^2 maps, R=40 (per-triangle lambda 0..2)", s, 40


def mip_agreement(tex):
    g = refgl.run(synth.unit_quad(tex), 16, want_mips=True)
    chain, offs, n = oracle.build_mips(tex["baseColorTexture"])
    out = {}
    for l, m in enumerate(g["mips"]):
        o = chain[int(offs[l]): int(offs[l]) + m.shape[0] * m.shape[1]].reshape(m.shape)
        d = np.abs(m.astype(int) - o.astype(int))
        out[f"level{l}"] = {"max_byte_diff": int(d.max()), "fraction_of_bytes_differing": float((d > 0).mean())}
    return out


def main():
    if not refgl.available():
        print("oracle/_ref/ref_gl_check not built")
        return 1
    oracle.build()
    base = synth.procedural_textures(64)
    rep = {"what": __doc__.split("\n\n")[1], "cells": {}, "glGenerateMipmap_vs_pinned": {}}
    for lod_mode, lname in ((0, "oracle as pinned: lambda = log2(rho)"), (1, "oracle with llvmpipe's lambda = 0.5 fast_log2(rho^2) (diagnostic switch)")):
        oracle.lib().orc_debug_set_lod_mode(lod_mode)
        for tname, tex in (("noise", base), ("multiples_of_64", tie_free(base))):
            if lod_mode == 0:
                rep["glGenerateMipmap_vs_pinned"][tname] = mip_agreement(tex)
            for sname, flags in (("default_sampler", False), ("fp32_sampler_per_pixel_lod", True)):
                cell = {}
                for name, scene, R in scenes(tex):
                    r = refgl.compare(scene, R, oracle, float_sampler=flags)
                    if r is None:
                        print("no GL context on this machine")
                        return 1
                    assert r["gl_counter"] == r["oracle_counter"], name
                    cell[name] = {f: r[f]["max_abs"] for f in ("color", "normal", "pbr", "position")} | {"mean_abs_color": r["color"]["mean_abs"]}
                rep["cells"][f"{lname} / {tname} / {sname}"] = cell
                print(f"lod_mode {lod_mode} {tname:16s} {sname:28s}", {k[:22]: (round(v["color"], 7), round(v["pbr"], 7), round(v["normal"], 7)) for k, v in cell.items()})
    oracle.lib().orc_debug_set_lod_mode(0)
    out = os.path.join(ROOT, "profiles", "r03", "ref_gl_decomposition.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", out)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
