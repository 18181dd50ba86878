"""A low-polygon scene at a high density: N tilted patches of 6 x 6 cells (72 triangles each, every one taller than 128 pixel rows at
R = 2048) — the wave-counted triangles of k_count_scan, dozens per wave.  Kernel times of the multi-pass pipeline, the counter against
the team kernel's.   python tools/tall_probe.py [patches=12] [R=2048]"""
import json
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from mesh2splat_amd import synth  # noqa: E402
from mesh2splat_amd.converter import Converter  # noqa: E402
from mesh2splat_amd.scene import Mesh, Scene  # noqa: E402


def scene_of(n_patches, seed=7):
    rng = np.random.default_rng(seed)
    meshes = []
    for i in range(n_patches):
        k = i % 3
        ax = np.eye(3)
        U, V = ax[(k + 1) % 3], ax[(k + 2) % 3]
        tilt = rng.normal(scale=0.08, size=3)
        o = rng.uniform(0.0, 0.45, 3)
        v = synth.patch_vertices(6, 6, o, 0.5 * (U + tilt), 0.5 * (V - tilt))
        meshes.append(Mesh(name=f"patch_{i}", vertices=v, base_color=(1.0, 1.0, 1.0, 1.0), textures={}))
    return Scene(meshes)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    scene = scene_of(n)
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    c.set_pipeline("multipass")
    for _ in range(3):
        tot = c.convert(R)
    c.set_profiling(True)
    ms = []
    for _ in range(7):
        assert c.convert(R) == tot
        ms.append(c.last_kernel_ms())
    cnt = c.download_triangle_counts()
    med = {k: float(np.median([m[k] for m in ms])) for k in ("count", "emit")}
    print(json.dumps({"patches": n, "R": R, "triangles": int(c.num_triangles), "gaussians": int(tot),
                      "kernel_ms": med, "max_count_per_triangle": int(cnt.max())}))
