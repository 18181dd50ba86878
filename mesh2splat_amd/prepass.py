"""Parameters of the viewer prepass (`Converter.prepass`), named after the RenderContext fields the reference's
GaussiansPrepass::execute reads (GaussiansPrepass.cpp:18-33; RenderContext.hpp:34-111)."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np


def _eye() -> np.ndarray:
    return np.eye(4, dtype=np.float32)


@dataclass
class PrepassParams:
    """Matrices are 4x4 float32 arrays in glm's memory order: m[c] is COLUMN c (i.e. the transpose of the usual numpy
    row-major math matrix)."""
    view_mat: np.ndarray = field(default_factory=_eye)       # RenderContext::viewMat   (u_worldToView)
    proj_mat: np.ndarray = field(default_factory=_eye)       # RenderContext::projMat   (u_viewToClip)
    model_mat: np.ndarray = field(default_factory=_eye)      # RenderContext::modelMat  (u_modelToWorld)
    renderer_resolution: tuple = (1280, 720)                 # RenderContext::rendererResolution (glm::ivec2)
    near_plane: float = 0.01                                 # renderer.cpp:186-187
    far_plane: float = 100.0
    gaussian_std: float = 0.65                               # u_stdDev = gaussianStd / resolutionTarget
    resolution_target: int = 1024
    render_mode: int = 0                                     # 0 colour, 1 depth, 2 normal, 3 geometry (random), 6 as 0
    format: int = 0                                          # 0 mesh2splat, 1 classic 3DGS .ply, 2 compressed PBR
    ply_has_pbr: bool = False
    perform_mesh_depth_test: bool = False
    arrival_order: bool = False                              # True: survivors in arrival order (the reference's atomic append)
    mesh_depth: object = None                                # (h, w) float32 window-space depth, row 0 = bottom: numpy array
                                                             # (host) or a CUDA torch tensor (used in place)


class PrepassParamsC(C.Structure):
    """== m2s_prepass_params (include/m2s.h)."""
    _fields_ = [("world_to_view", C.c_float * 16), ("view_to_clip", C.c_float * 16), ("model_to_world", C.c_float * 16),
                ("resolution", C.c_int32 * 2), ("near_far", C.c_float * 2), ("gaussian_std", C.c_float),
                ("resolution_target", C.c_uint32), ("render_mode", C.c_int32), ("format", C.c_uint32),
                ("ply_has_pbr", C.c_uint32), ("depth_test_mesh", C.c_uint32),
                ("depth", C.c_void_p), ("depth_w", C.c_uint32), ("depth_h", C.c_uint32), ("depth_on_device", C.c_uint32), ("arrival_order", C.c_uint32)]


def to_c(p: PrepassParams):
    """-> (PrepassParamsC, keep-alive list)"""
    c = PrepassParamsC()
    keep = []
    for name, m in (("world_to_view", p.view_mat), ("view_to_clip", p.proj_mat), ("model_to_world", p.model_mat)):
        a = np.ascontiguousarray(m, np.float32).reshape(16)
        getattr(c, name)[:] = a.tolist()
    c.resolution[:] = [int(p.renderer_resolution[0]), int(p.renderer_resolution[1])]
    c.near_far[:] = [float(np.float32(p.near_plane)), float(np.float32(p.far_plane))]
    c.gaussian_std = float(np.float32(p.gaussian_std))
    c.resolution_target = int(p.resolution_target)
    c.render_mode = int(p.render_mode)
    c.format = int(p.format)
    c.ply_has_pbr = 1 if p.ply_has_pbr else 0
    c.depth_test_mesh = 1 if p.perform_mesh_depth_test else 0
    c.depth_on_device = 0
    c.arrival_order = 1 if p.arrival_order else 0
    if p.mesh_depth is not None and hasattr(p.mesh_depth, "data_ptr"):      # torch tensor on the device
        d = p.mesh_depth.contiguous().float()
        keep.append(d)
        c.depth = d.data_ptr()
        c.depth_h, c.depth_w = int(d.shape[0]), int(d.shape[1])
        c.depth_on_device = 1
    elif p.mesh_depth is not None:
        d = np.ascontiguousarray(p.mesh_depth, np.float32)
        keep.append(d)
        c.depth = d.ctypes.data
        c.depth_h, c.depth_w = d.shape
    else:
        c.depth = None
        c.depth_w = c.depth_h = 0
    return c, keep
