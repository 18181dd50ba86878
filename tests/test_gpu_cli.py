"""-m gpu: the headless CLI (.glb -> HIP conversion -> .ply) end to end, against the oracle and the
Python path."""
import os
import subprocess

import numpy as np
import pytest

from mesh2splat_amd import _lib, gltf_io, synth
from mesh2splat_amd.converter import ConversionPass, RenderContext, SceneManager
from mesh2splat_amd.scene import resolution_from_quality

pytestmark = pytest.mark.gpu
EXE = os.path.join(os.path.dirname(_lib.LIB_PATH), "mesh2splat")


@pytest.mark.parametrize("fmt", [0, 1, 2])
def test_cli_matches_python_path_and_oracle(tmp_path, hiplib, oracle, fmt):
    scene = synth.sphere_grid(2, n=5, tex_size=32)
    glb, out = str(tmp_path / "s.glb"), str(tmp_path / "cli.ply")
    gltf_io.write_glb(scene, glb)
    r = subprocess.run([EXE, glb, out, "--density", "96", "--format", str(fmt), "--std", "0.65", "--timing"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "8 mesh(es)" in r.stdout and "Gaussians" in r.stdout
    # same scene through the Python mirror of the reference interface
    loaded = gltf_io.load_glb(glb)
    ctx = RenderContext(loaded, resolutionTarget=96, gaussianStd=0.65)
    ConversionPass().execute(ctx)
    ref_ply = str(tmp_path / "py.ply")
    SceneManager(ctx).exportPly(ref_ply, fmt)
    assert open(out, "rb").read() == open(ref_ply, "rb").read()
    # and the oracle agrees on what went into the file
    total, orec, _ = oracle.convert(loaded, 96)
    assert ctx.numberOfGaussians == total
    if fmt in (0, 1):
        got, _ = gltf_io.read_ply(out)
        assert got.shape[0] == min(total, oracle.reference_cap(96, 8))
        assert np.allclose(got[:, 0:3], orec[:, 0:3], rtol=1e-4, atol=1e-6)


def test_cli_defaults_are_the_guis(tmp_path, hiplib):
    """No --density: R = int(16 + 0.5*(1024-16)) = 520 (ImGuiUI.cpp:512, main.cpp:26)."""
    glb, out = str(tmp_path / "q.glb"), str(tmp_path / "q.ply")
    gltf_io.write_glb(synth.unit_quad(), glb)
    r = subprocess.run([EXE, glb, out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert resolution_from_quality(0.5, 1024) == 520 and "density 520 -> 270400 Gaussians" in r.stdout


@pytest.mark.parametrize("fmt", [0, 2])
def test_cli_sharded_code_path_with_one_rank(tmp_path, hiplib, fmt):
    """`--gpus N` with N = 1 (--force-sharded): fork after the CPU-only load, shard plan, RCCL communicator and counter
    exchange through libm2s_hip.so, the slice writer (and, with --gather, the record exchange to rank 0 + whole-file export)
    — the same file, byte for byte, as the single-process path.  Also a scene above the reference's cap (global cap semantics)."""
    scene = synth.sphere_grid(2, n=5, tex_size=32)
    glb = str(tmp_path / "s.glb")
    gltf_io.write_glb(scene, glb)
    for density, cap in (("96", []), ("96", ["--cap", "20000"])):
        one = str(tmp_path / "one.ply")
        r = subprocess.run([EXE, glb, one, "--density", density, "--format", str(fmt)] + cap, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for extra in ([], ["--gather"]):
            out = str(tmp_path / "sharded.ply")
            r = subprocess.run([EXE, glb, out, "--density", density, "--format", str(fmt), "--force-sharded", "--timing"] + cap + extra,
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr + r.stdout
            assert open(out, "rb").read() == open(one, "rb").read(), (fmt, cap, extra)


def test_cli_batch_equals_single_runs(tmp_path, hiplib):
    """--batch: loader thread | upload + convert | exporter thread over two alternating contexts; every output identical to
    the one-file command's."""
    ind, outd, single = tmp_path / "in", tmp_path / "out", tmp_path / "single"
    for d in (ind, outd, single):
        d.mkdir()
    for i in range(5):
        gltf_io.write_glb(synth.cube_sphere(8 + 3 * i, tex_size=32, seed=7 + i), str(ind / f"m{i}.glb"))
    r = subprocess.run([EXE, "--batch", str(ind), "--out", str(outd), "--density", "200", "--format", "1"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "5 file(s)" in r.stdout and "meshes/s" in r.stdout
    for i in range(5):
        r = subprocess.run([EXE, str(ind / f"m{i}.glb"), str(single / f"m{i}.ply"), "--density", "200", "--format", "1"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        assert (outd / f"m{i}.ply").read_bytes() == (single / f"m{i}.ply").read_bytes()


@pytest.mark.parametrize("ranks", [2, 3])
def test_cli_several_ranks_on_one_gpu(tmp_path, hiplib, ranks):
    """`--gpus N --one-device`: N forked PROCESSES, each with its own context on the same GPU, its own fragment-balanced
    triangle range and its own rows of the one output file (concurrent writers) — the N > 1 logic end to end on the one GPU this
    box has: shard plan, per-rank conversion with the cap lifted, counter exchange, global cap, slice offsets.  Same bytes as
    the single-process run."""
    scene = synth.sphere_grid(2, n=6, tex_size=32)
    glb = str(tmp_path / "s.glb")
    gltf_io.write_glb(scene, glb)
    for fmt, cap in ((0, []), (2, []), (1, ["--cap", "30000"])):
        one, many = str(tmp_path / "one.ply"), str(tmp_path / "many.ply")
        r = subprocess.run([EXE, glb, one, "--density", "128", "--format", str(fmt)] + cap, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        r = subprocess.run([EXE, glb, many, "--density", "128", "--format", str(fmt), "--gpus", str(ranks), "--one-device", "--timing"] + cap,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr + r.stdout
        assert f"{ranks} GPUs" in r.stdout and r.stdout.count("[rank ") == ranks
        assert open(many, "rb").read() == open(one, "rb").read(), (fmt, cap)


@pytest.mark.parametrize("ranks", [2, 3])
def test_cli_ranks_as_threads_with_gather(tmp_path, hiplib, ranks):
    """`--gpus N --threads [--gather] --one-device`: the ranks are threads of ONE process (the reference's shape); with --gather the
    blocks travel to rank 0 over the in-process transport (m2s_dist_local_id) and rank 0 writes the whole file — the C++ consumer
    of the same code the RCCL transport runs, with more than one rank, on the one GPU this box has.  Same bytes as one process."""
    scene = synth.sphere_grid(2, n=6, tex_size=32)
    glb = str(tmp_path / "s.glb")
    gltf_io.write_glb(scene, glb)
    for fmt, cap in ((0, []), (2, ["--cap", "30000"])):
        one = str(tmp_path / "one.ply")
        r = subprocess.run([EXE, glb, one, "--density", "128", "--format", str(fmt)] + cap, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for extra in ([], ["--gather"]):
            many = str(tmp_path / "many.ply")
            r = subprocess.run([EXE, glb, many, "--density", "128", "--format", str(fmt), "--gpus", str(ranks), "--threads", "--one-device", "--timing"] + cap + extra,
                               capture_output=True, text=True, timeout=300)
            assert r.returncode == 0, r.stderr + r.stdout
            assert r.stdout.count("[rank ") == ranks
            if extra:
                assert "in-process transport" in r.stdout
            assert open(many, "rb").read() == open(one, "rb").read(), (fmt, cap, extra)
