"""Shared inputs of the prepass tests (reference pin, golden generator, GPU parity): deterministic records and a list of
named parameter sets covering every branch of gaussianSplattingPrepassCS.glsl."""
import numpy as np

import camera
from mesh2splat_amd import synth
from mesh2splat_amd.prepass import PrepassParams

SEED = 0x4D32535F50524550


def base_records(oracle, n: int = 12, R: int = 48) -> np.ndarray:
    """Records of a real conversion (textured cube-sphere) -> realistic scales / rotations / normals / pbr."""
    scene = synth.cube_sphere(n, tex_size=32)
    _, rec, _ = oracle.convert(scene, R, cap=0)
    return np.ascontiguousarray(rec, np.float32).reshape(-1, 24)


def hostile_records(n: int = 512) -> np.ndarray:
    """Random records with the things a loaded .ply can contain: translucent alpha, unnormalised quaternions, equal and
    zero scales, huge scales, positions behind and far outside the frustum, non-finite values."""
    rng = np.random.default_rng(SEED)
    r = np.zeros((n, 24), np.float32)
    r[:, 0:3] = rng.uniform(-3, 3, (n, 3))
    r[:, 3] = 1
    r[:, 4:8] = rng.uniform(0, 1, (n, 4))
    r[:, 8:11] = np.exp(rng.uniform(-9, 1, (n, 3)))
    r[:, 12:15] = rng.normal(size=(n, 3))
    r[:, 16:20] = rng.normal(size=(n, 4))
    r[:, 20:22] = rng.uniform(0, 1, (n, 2))
    r[:, 23] = 1
    r[::7, 8:11] = r[::7, 8:9]                     # equal scales (min-index ties, format 1)
    r[5::31, 8:11] = 0                             # degenerate
    r[9::37, 8:11] = 1e4                           # quad axes hit the 1024 px clamp
    r[11::41, 16:20] = 0                           # zero quaternion
    r[3::53, 0:3] = np.nan
    r[4::59, 0:3] = np.inf
    r[6::61, 8] = np.nan
    r[::2, 7] = 1.0                                # opaque half (depth test applies at alpha > .95)
    return r


def default_camera(res=(640, 360)):
    view = camera.look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1))
    proj = camera.perspective(45.0, res[0] / res[1], 0.01, 100.0)
    return view, proj


def depth_image(res, seed=3) -> np.ndarray:
    """Window-space depth with structure on the scale of the splats: some occlude, some do not."""
    rng = np.random.default_rng(seed)
    h, w = res[1] // 4, res[0] // 4
    coarse = rng.uniform(0.97, 1.0, (h // 8 + 1, w // 8 + 1)).astype(np.float32)
    d = np.kron(coarse, np.ones((8, 8), np.float32))[:h, :w]
    return np.ascontiguousarray(d)


def cases():
    """-> list of (name, PrepassParams)"""
    res = (640, 360)
    view, proj = default_camera(res)
    out = []

    def add(name, **kw):
        p = PrepassParams(view_mat=view, proj_mat=proj, renderer_resolution=res, resolution_target=48, **kw)
        out.append((name, p))

    add("colour")                                                       # format 0, mode 0, identity model
    add("depth_mode", render_mode=1)
    add("normal_mode", render_mode=2)
    add("geometry_mode", render_mode=3)
    add("mode6", render_mode=6)
    add("mode4_black", render_mode=4)
    add("ply_classic", format=1, render_mode=2)                        # shortest-axis normal
    add("ply_classic_pbr", format=1, ply_has_pbr=True, render_mode=2)
    add("ply_compressed", format=2, render_mode=2)                     # normal stays (1,0,0,0)
    add("format3", format=3, gaussian_std=0.9)
    add("depth_test", perform_mesh_depth_test=True, mesh_depth=depth_image(res))
    add("depth_test_format1_ignored", perform_mesh_depth_test=True, mesh_depth=depth_image(res), format=1)
    add("model_trs", model_mat=camera.trs((0.3, -0.2, 0.1), (1, 2, 3), 37.0, (1.5, 0.7, 1.2)), render_mode=2)
    add("wide_std", gaussian_std=4.0)
    # camera inside the object looking out: many behind the eye, near-plane culls
    p = PrepassParams(view_mat=camera.look_at((0.2, 0.1, 0.0), (1, 0.3, 0.2)), proj_mat=camera.perspective(70.0, 1.0, 0.01, 100.0),
                      renderer_resolution=(512, 512), resolution_target=48)
    out.append(("inside", p))
    return out
