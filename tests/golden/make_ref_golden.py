#!/usr/bin/env python
"""Regenerates tests/golden/ref_host/: outputs of the REFERENCE's own host code (oracle/_ref/ref_host_check,
built by `make -C oracle ref` from /root/reference) on small synthetic inputs, so that the comparison with the
real reference also runs where /root/reference does not exist.

    python tests/golden/make_ref_golden.py

Inputs (.glb written by mesh2splat_amd.gltf_io.write_glb, records.bin from the oracle) and the reference's
outputs (*.scene.bin = dump of SceneManager::loadModel's vertex upload / bbox / material / textures,
ref_fmt*.ply = parsers::savePlyVector, ref_read_fmt*.bin = parsers::loadPlyFile, glsl_*.dump.bin = what the
reference's converter{VS,GS,FS}.glsl computed when run as C++ through glm by oracle/_ref/ref_glsl_check,
prepass_*.out.bin = what GaussiansPrepass::execute + gaussianSplattingPrepassCS.glsl produced for the parameter sets
of tests/prepass_cases.py on prepass_records.bin, by oracle/_ref/ref_prepass_check) are all committed."""
import json
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import refhost  # noqa: E402
from mesh2splat_amd import gltf_io, synth  # noqa: E402
from oracle import oracle  # noqa: E402

OUT = os.path.join(HERE, "ref_host")


def scene_cases():
    q = np.array([0.1, 0.5, -0.2, 0.8])
    q /= np.linalg.norm(q)
    trs = [dict(translation=(1, 2, 3), rotation=q, scale=(2, 0.5, 1.5)), dict(scale=(1, -1, 1)), dict(translation=(-0.0, 4, 0)),
           dict(matrix=[1, 0, 0, 0, 0, 0, 1, 0, 0, -1, 0, 0, 5, 6, 7, 1])] + [{}] * 4
    yield "trs_nested", synth.sphere_grid(2, n=2, tex_size=8), dict(node_trs=trs, nested=True)
    yield "flat_nonindexed", synth.cube_sphere(2), dict(with_normals=False, with_tangents=False, indexed=False)
    yield "no_uv_u8", synth.cube_sphere(2), dict(with_uvs=False, index_type="u8")
    mixed = synth.sphere_grid(2, n=2, tex_size=8)
    mixed.meshes[1].textures.pop("normalTexture", None)
    mixed.meshes[2].textures.clear()
    mixed.meshes[3].base_color = (0.2, 0.4, 0.6, 0.8)
    yield "mixed_materials_u32", mixed, dict(index_type="u32")


def jpeg_cases():
    import io
    from PIL import Image
    rng = np.random.default_rng(3)
    y, x = np.mgrid[0:40, 0:56]
    a = np.stack([128 + 100 * np.sin(x / 7.0) * np.cos(y / 11.0), 128 + 90 * np.cos(x / 5.0 + y / 9.0), 128 + 80 * np.sin((x + y) / 13.0)], -1)
    img = np.clip(a + rng.normal(0, 6, a.shape), 0, 255).astype(np.uint8)
    for name, kw in (("jpeg_baseline_420", dict(quality=80, subsampling=2)), ("jpeg_progressive_422", dict(quality=70, subsampling=1, progressive=True))):
        b = io.BytesIO()
        Image.fromarray(img).save(b, "JPEG", **kw)
        yield name, b.getvalue()


def glsl_cases():
    """(name, scene, R, FS samples per triangle) for the shader-level fixtures."""
    yield "sphere", synth.cube_sphere(4, tex_size=16), 64, 3
    yield "soup", synth.random_soup(150, seed=21, textures=synth.procedural_textures(8, 2)), 64, 3
    plain = synth.random_soup(100, seed=22)
    plain.meshes[0].base_color = (0.25, 0.5, 0.75, 0.5)
    yield "soup_untextured", plain, 64, 2


def pipeline_cases():
    """(name, scene, R, write_glb kwargs) for the whole-pass fixtures (kept small: they are committed)."""
    mixed = synth.sphere_grid(2, n=2, tex_size=8)
    mixed.meshes[1].textures.pop("normalTexture", None)
    mixed.meshes[2].textures.clear()
    mixed.meshes[3].base_color = (0.2, 0.4, 0.6, 0.8)
    q = np.array([0.3, -0.1, 0.2, 0.9])
    q /= np.linalg.norm(q)
    trs = [dict(translation=(0.5, 0, 0), rotation=q), dict(scale=(1, -1, 1)), {}, dict(scale=(0.5, 2, 1))] + [{}] * 4
    yield "mixed_trs", mixed, 16, dict(node_trs=trs, nested=True)
    yield "soup", synth.random_soup(60, seed=9, textures=synth.procedural_textures(8, 2)), 32, dict(indexed=False)


def authored_cases():
    """(name, scene, flags, trs): .glb files written by the REFERENCE's tiny_gltf + stb_image_write (refhost.write_glb_by_tinygltf),
    not by this repository's gltf_io: the loader then reads assets it did not generate (PNG streams of another encoder, tiny_gltf's
    own JSON layout, its buffer packing)."""
    q = np.array([0.2, -0.4, 0.1, 0.88])
    q /= np.linalg.norm(q)
    grid = synth.sphere_grid(2, n=2, tex_size=16)
    grid.meshes[2].textures.pop("metallicRoughnessTexture", None)
    grid.meshes[5].textures.clear()
    grid.meshes[6].base_color = (0.9, 0.3, 0.5, 0.7)
    trs = [((0.5, -1, 2), tuple(q), (1.5, 0.75, 2)), ((0, 0, 0), (0, 0, 0, 1), (1, -1, 1))] + [((k * 0.25, 0, -k), (0, 0, 0, 1), (1, 1, 1)) for k in range(6)]
    yield "authored_grid_trs", grid, 8, trs
    yield "authored_interleaved_u16", synth.cube_sphere(3, tex_size=32), 1 | 2, None
    yield "authored_nonindexed", synth.random_soup(40, seed=5, textures=synth.procedural_textures(8, 4)), 4, None


def write_authored(tmp):
    for name, scene, flags, trs in authored_cases():
        glb = os.path.join(OUT, name + ".glb")
        refhost.write_glb_by_tinygltf(scene, glb, tmp, flags=flags, trs=trs)
        refhost.load_scene(glb, tmp)
        shutil.copy(os.path.join(tmp, "ref_scene.bin"), os.path.join(OUT, name + ".scene.bin"))


def sample_records():
    scene = synth.random_soup(24, seed=11, textures=synth.procedural_textures(16, 2))
    scene.meshes[0].base_color = (1.0, 0.9, 0.8, 1.0)
    _, rec, _ = oracle.convert(scene, 40, cap=0)
    rec = rec[:120].copy()
    rec[::3, 7] = 1.0            # opaque -> opacity +inf
    rec[1::3, 7] = 0.37
    rec[5, 4:7] = (1.5, -0.25, 0.5)
    rec[7, 12:15] = (0.0, 0.0, -1.0)
    rec[8, 12:15] = (-0.3, 0.2, -0.6)
    return rec


def main():
    assert refhost.available(), "build oracle/_ref first: make -C oracle ref"
    tmp = tempfile.mkdtemp()
    if "--only-authored" in sys.argv:        # (adds the tiny_gltf-authored assets without touching the other fixtures)
        write_authored(tmp)
        return
    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    write_authored(tmp)
    for name, scene, kw in scene_cases():
        glb = os.path.join(OUT, name + ".glb")
        gltf_io.write_glb(scene, glb, **kw)
        refhost.load_scene(glb, tmp)
        shutil.copy(os.path.join(tmp, "ref_scene.bin"), os.path.join(OUT, name + ".scene.bin"))
    # JPEG-textured scenes (decoded by stb_image in the reference)
    for name, jpg in jpeg_cases():
        glb = os.path.join(OUT, name + ".glb")
        gltf_io.write_glb(synth.cube_sphere(2, tex_size=8), glb, png_override={"baseColorTexture": jpg, "metallicRoughnessTexture": jpg})
        refhost.load_scene(glb, tmp)
        shutil.copy(os.path.join(tmp, "ref_scene.bin"), os.path.join(OUT, name + ".scene.bin"))
    rec = sample_records()
    rec.tofile(os.path.join(OUT, "records.bin"))
    sm = np.float32(0.65) / np.float32(40)
    for fmt in (0, 1, 2):
        ply = os.path.join(OUT, f"ref_fmt{fmt}.ply")
        refhost.write_ply(rec, ply, fmt, sm, tmp)
        if fmt in (0, 1):
            refhost.read_ply(ply, tmp)
            shutil.copy(os.path.join(tmp, "ref_plyread.bin"), os.path.join(OUT, f"ref_read_fmt{fmt}.bin"))
    # the reference's conversion SHADERS run as C++ (oracle/ref_glsl_check.cpp): keep what they produced
    assert refhost.glsl_available()
    for name, scene, R, samples in glsl_cases():
        dump = os.path.join(OUT, f"glsl_{name}.dump.bin")
        rep = refhost.run_glsl_check(scene, R, samples, tmp, dump_path=dump)
        shutil.copy(os.path.join(tmp, "glsl_scene.bin"), os.path.join(OUT, f"glsl_{name}.scene.bin"))
        with open(os.path.join(OUT, f"glsl_{name}.report.json"), "w") as f:
            json.dump(dict(R=R, samples=samples, **rep), f, indent=1)
    # the reference's WHOLE path (loadModel -> ConversionPass::execute -> shaders -> exportPly) on the software GL
    assert refhost.pipeline_available()
    for name, scene, R, kw in pipeline_cases():
        glb = os.path.join(OUT, f"pipe_{name}.glb")
        gltf_io.write_glb(scene, glb, **kw)
        ply = os.path.join(OUT, f"pipe_{name}_R{R}.ply") if name == "soup" else None   # SceneManager::exportPly, format 1
        refhost.run_pipeline(glb, R, tmp, ply_path=ply, fmt=1, std=0.65, out_path=os.path.join(OUT, f"pipe_{name}_R{R}.records.bin"))
    # the reference's viewer prepass (GaussiansPrepass.cpp + gaussianSplattingPrepassCS.glsl) on the software GL
    assert refhost.prepass_available()
    import prepass_cases
    prec = np.concatenate([prepass_cases.base_records(oracle, 3, 12), prepass_cases.hostile_records(128)])
    prec.tofile(os.path.join(OUT, "prepass_records.bin"))
    for name, p in prepass_cases.cases():
        refhost.run_prepass(p, prec, tmp)
        shutil.copy(os.path.join(tmp, "prepass_out.bin"), os.path.join(OUT, f"prepass_{name}.out.bin"))
    print("wrote", sorted(os.listdir(OUT)), sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")


if __name__ == "__main__":
    main()
