#!/usr/bin/env python
"""End-to-end ms/mesh of the headless CLI on the C3 workload (reported next to the conversion-only metric):
.glb parse + texture decode, upload, conversion, download + .ply write, for the three export formats."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mesh2splat_amd import gltf_io, synth  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 289
    tex = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    cli = os.path.join(ROOT, "mesh2splat_amd", "_build", "mesh2splat")
    out = {"workload": f"cube-sphere n={n}, 3 x {tex}^2 maps (PNG), R={R}"}
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        glb = os.path.join(tmp, "c3.glb")
        t0 = time.perf_counter()
        gltf_io.write_glb(synth.cube_sphere(n, tex_size=tex), glb)
        out["glb_bytes"] = os.path.getsize(glb)
        out["write_glb_s (test tooling, not part of the path)"] = time.perf_counter() - t0
        for fmt in (0, 1, 2):
            ply = os.path.join(tmp, f"out{fmt}.ply")
            best = None
            for _ in range(2):
                w0 = time.perf_counter()
                r = subprocess.run([cli, glb, ply, "--density", str(R), "--format", str(fmt), "--timing"], capture_output=True, text=True, timeout=600)
                wall = (time.perf_counter() - w0) * 1e3
                if r.returncode != 0:
                    raise SystemExit(r.stderr)
                m = re.search(r"load ([\d.]+) ms \(HIP runtime \+ context ([\d.]+) ms, on a second thread; waited ([\d.]+) ms for it\) \| upload ([\d.]+) ms "
                              r"\(geometry ([\d.]+), textures ([\d.]+), allocations ([\d.]+)\) \| convert ([\d.]+) ms.*\| export ([\d.]+) ms \| total ([\d.]+) ms", r.stdout)
                t = dict(zip(("load_ms", "hip_init_ms_overlapped", "waited_for_init_ms", "upload_ms", "upload_geometry_ms", "upload_textures_ms",
                              "upload_alloc_ms", "convert_first_call_ms", "export_ms", "total_ms"), map(float, m.groups())))
                t["wall_ms_whole_process"] = wall
                if best is None or t["total_ms"] < best["total_ms"]:
                    best = t
            best["ply_bytes"] = os.path.getsize(ply)
            g = re.search(r"(\d+) Gaussians", r.stdout)
            out[f"format{fmt}"] = best
            out["stdout_tail"] = r.stdout.strip().splitlines()[0][:200]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
