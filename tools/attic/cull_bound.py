"""Upper bound for a pre-cull pass on sub-pixel meshes: kernel time of the single-pass kernel on the whole mesh vs on a mesh
made of only the triangles that produce fragments (same bounding box, same order)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
from mesh2splat_amd.scene import Mesh, Scene
for n, R in ((721, 1024), (1021, 2048)):
    scene = synth.cube_sphere(n, tex_size=2048)
    c = Converter(0); c.upload_scene(scene); c.set_max_gaussians(0); c.set_profiling(True)
    total = c.convert(R)
    ms = []
    for _ in range(10):
        c.convert(R); ms.append(c.last_kernel_ms()["fused"])
    cnt = c.download_triangle_counts()
    c.close()
    keep = np.repeat(cnt > 0, 3)
    m0 = scene.meshes[0]
    m = Mesh(name="kept", vertices=m0.vertices[keep], textures=m0.textures)
    s2 = Scene([m])
    s2.meshes[0].bbox_min, s2.meshes[0].bbox_max = scene.meshes[0].bbox_min.copy(), scene.meshes[0].bbox_max.copy()
    c = Converter(0); c.upload_scene(s2); c.set_max_gaussians(0); c.set_profiling(True)
    t2 = c.convert(R)
    ms2 = []
    for _ in range(10):
        c.convert(R); ms2.append(c.last_kernel_ms()["fused"])
    print("n", n, "R", R, "T", scene.n_triangles, "N", total, "fused", round(float(np.median(ms)), 4), "| survivors", s2.n_triangles,
          f"({s2.n_triangles / scene.n_triangles:.2f})", "N", t2, "fused", round(float(np.median(ms2)), 4), c.last_pipeline, flush=True)
    c.close()
