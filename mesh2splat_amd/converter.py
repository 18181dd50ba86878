"""Host-side mirror of the reference's operator interface for the conversion path.

Reference                                             | here
------------------------------------------------------+--------------------------------------------
IRenderPass / ConversionPass::execute(RenderContext&) | ConversionPass.execute(RenderContext)
  (RenderPass.hpp:11-29, ConversionPass.cpp:9-68)     |
RenderContext fields the pass reads / writes          | RenderContext (same names)
  (RenderContext.hpp:64,73,83,86,90)                  |
SceneManager::exportPly(outputFile, exportFormat)     | SceneManager.exportPly
  (SceneManager.cpp:651-678)                          |
parsers::savePlyVector                                | write_ply

All compute happens in the HIP library behind include/m2s.h; this file only marshals arguments.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from . import _lib
from .scene import RECORD_FLOATS, TEXTURE_SLOTS, Scene


def marshal_scene(scene: Scene):
    """Scene -> (m2s_mesh[] for the C ABI, objects that must stay alive while it is used)."""
    n = scene.n_meshes
    arr = (_lib.MeshC * max(1, n))()
    keep = []
    for i, m in enumerate(scene.meshes):
        v = np.ascontiguousarray(m.vertices, np.float32)
        keep.append(v)
        arr[i].vertices = v.ctypes.data
        arr[i].n_vertices = v.shape[0]
        arr[i].stride_floats = v.shape[1]
        for k in range(3):
            arr[i].bbox_min[k] = float(m.bbox_min[k])
            arr[i].bbox_max[k] = float(m.bbox_max[k])
        for k in range(4):
            arr[i].base_color[k] = float(m.base_color[k])
        for k, key in enumerate(TEXTURE_SLOTS):
            t = m.textures.get(key)
            if t is None:
                continue
            keep.append(t)
            arr[i].tex[k].rgba8 = t.ctypes.data
            arr[i].tex[k].width = t.shape[1]
            arr[i].tex[k].height = t.shape[0]
    return arr, keep


class Converter:
    """Thin owner of one m2s_ctx (one per host thread / per GPU rank)."""

    def __init__(self, device: int = 0):
        self._L = _lib.load()
        h = C.c_void_p()
        st = self._L.m2s_create(int(device), C.byref(h))
        if st != _lib.M2S_OK:
            raise _lib.M2SError(st, self._L.m2s_last_error(None).decode())
        self._h = h
        self.device = int(device)
        self._keep = None

    # -- helpers ------------------------------------------------------------------------------------
    def _check(self, st: int):
        if st != _lib.M2S_OK:
            raise _lib.M2SError(st, self._L.m2s_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.m2s_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- scene ------------------------------------------------------------------------------------
    def set_triangle_range(self, first: int, count: Optional[int]):
        self._check(self._L.m2s_set_triangle_range(self._h, int(first), (1 << 64) - 1 if count is None else int(count)))

    PREPARE_UPLOAD, PREPARE_EXPORT, PREPARE_KERNELS = 1, 2, 4

    def prepare(self, flags: int = 7):
        """m2s_prepare: staging buffers (upload / export) and the kernels' code objects now, not inside the first upload /
        export / conversion of the process (the command line does this on a second thread while it parses the file)."""
        self._check(self._L.m2s_prepare(self._h, int(flags)))

    def set_resolution_hint(self, R: int):
        """The resolutionTarget the next upload_scene prepares for (0: the last R converted at, else 1024)."""
        self._check(self._L.m2s_set_resolution_hint(self._h, int(R)))

    def upload_scene(self, scene: Scene):
        arr, keep = marshal_scene(scene)
        self._check(self._L.m2s_upload_scene(self._h, arr, scene.n_meshes))
        del keep

    def last_upload_ms(self) -> dict:
        """Wall-clock breakdown of the last upload_scene (ms)."""
        ms = (C.c_float * 4)()
        self._check(self._L.m2s_last_upload_ms(self._h, ms))
        return {"total": float(ms[0]), "geometry": float(ms[1]), "textures": float(ms[2]), "alloc": float(ms[3]),
                "warm": float(self._L.m2s_last_warm_ms(self._h))}

    # -- the pass ---------------------------------------------------------------------------------
    def set_max_gaussians(self, cap: int):
        """-1 reference formula (default), 0 unlimited, >0 explicit."""
        self._check(self._L.m2s_set_max_gaussians(self._h, int(cap)))

    def convert(self, R: int) -> int:
        total = C.c_uint64()
        self._check(self._L.m2s_convert(self._h, int(R), C.byref(total)))
        return int(total.value)

    def convert_into(self, R: int, device_ptr: int, capacity_records: int, stream: int = 0) -> int:
        total = C.c_uint64()
        self._check(self._L.m2s_convert_into(self._h, int(R), C.c_void_p(device_ptr), int(capacity_records),
                                             C.c_void_p(stream), C.byref(total)))
        return int(total.value)

    def submit(self, R: int, device_ptr: int = 0, capacity_records: int = 0, stream: int = 0) -> None:
        """Enqueue one conversion without waiting (m2s_convert_submit); device_ptr == 0 uses the context's buffer."""
        self._check(self._L.m2s_convert_submit(self._h, int(R), C.c_void_p(device_ptr or None), int(capacity_records),
                                               C.c_void_p(stream or None)))

    def wait(self) -> int:
        """Counter of the oldest submitted conversion, once it has finished (m2s_convert_wait)."""
        total = C.c_uint64()
        self._check(self._L.m2s_convert_wait(self._h, C.byref(total)))
        return int(total.value)

    @property
    def num_stored(self) -> int:
        return int(self._L.m2s_num_stored(self._h))

    @property
    def num_triangles(self) -> int:
        return int(self._L.m2s_num_triangles(self._h))

    @property
    def device_records(self) -> int:
        return int(self._L.m2s_device_records(self._h) or 0)

    def download(self) -> np.ndarray:
        n = self.num_stored
        out = np.empty((n, RECORD_FLOATS), np.float32)
        self._check(self._L.m2s_download(self._h, out.ctypes.data, n))
        return out

    def download_triangle_counts(self) -> np.ndarray:
        n = self.num_triangles
        out = np.empty(n, np.uint32)
        self._check(self._L.m2s_download_triangle_counts(self._h, out.ctypes.data, n))
        return out

    def export_ply(self, path: str, fmt: int = 0, gaussian_std: float = 0.65):
        self._check(self._L.m2s_export_ply(self._h, os.fsencode(path), int(fmt), float(gaussian_std)))

    def export_ply_slice(self, path: str, fmt: int, gaussian_std: float, first_row: int, n_rows: int, total_rows: int):
        """This rank's rows of a .ply that several ranks write together (m2s_export_ply_slice)."""
        self._check(self._L.m2s_export_ply_slice(self._h, os.fsencode(path), int(fmt), float(gaussian_std), int(first_row),
                                                 int(n_rows), int(total_rows)))

    def reserve_records(self, n: int) -> int:
        """Room for n records in the context-owned pool; its device address (m2s_reserve_records)."""
        p = C.c_void_p()
        self._check(self._L.m2s_reserve_records(self._h, int(n), C.byref(p)))
        return int(p.value or 0)

    def set_records(self, device_ptr: int, n: int, R: int):
        """Device-resident records from elsewhere (e.g. the merged buffer of a multi-GPU exchange) become the context's
        current records, zero copy (m2s_set_records)."""
        self._check(self._L.m2s_set_records(self._h, C.c_void_p(device_ptr or None), int(n), int(R)))

    def set_keep_positions(self, on: bool):
        """Say before converting that the records will be depth-sorted: the sparse kernel then also writes the 16-byte position plane."""
        self._check(self._L.m2s_set_keep_positions(self._h, 1 if on else 0))

    @property
    def positions_ready(self) -> bool:
        """the 16-byte position plane of the current records exists (left by their conversion or by an earlier depth sort)"""
        return bool(self._L.m2s_positions_ready(self._h))

    def sort_by_depth(self, world_to_view, download: bool = True):
        """RadixSortPass: sort the last conversion's records by floatBitsToUint(view-space z); returns them (or, with
        download=False, their number: the sorted records stay on the device).
        world_to_view: 4x4, applied as M @ (P,1) (stored column-major for the ABI, like glm)."""
        m = np.ascontiguousarray(np.asarray(world_to_view, np.float32).T.reshape(16))   # column-major
        n = C.c_uint64()
        self._check(self._L.m2s_sort_by_depth(self._h, m.ctypes.data_as(C.POINTER(C.c_float)), C.byref(n)))
        if not download:
            return int(n.value)
        out = np.empty((n.value, RECORD_FLOATS), np.float32)
        self._check(self._L.m2s_download_sorted(self._h, out.ctypes.data, n.value))
        return out

    def download_sorted(self) -> np.ndarray:
        """The context's sorted buffer (after sort_by_depth(download=False) or RcclExchange.sort_by_depth)."""
        n = int(self._L.m2s_num_sorted(self._h))
        out = np.empty((n, RECORD_FLOATS), np.float32)
        if n:
            self._check(self._L.m2s_download_sorted(self._h, out.ctypes.data, n))
        return out

    @property
    def last_sort_ms(self) -> float:
        return float(self._L.m2s_last_sort_ms(self._h))

    @property
    def last_sort_stage_ms(self) -> dict:
        """keys | radix sort | gather of the last sort_by_depth (profiling on)."""
        ms = (C.c_float * 3)()
        self._check(self._L.m2s_last_sort_stage_ms(self._h, ms))
        return {"keys": float(ms[0]), "radix_sort": float(ms[1]), "gather": float(ms[2])}

    def upload_records(self, records: np.ndarray):
        """Renderer::updateGaussianBuffer after LoadPly: (n, 24) float32 host records become the context's current records."""
        r = np.ascontiguousarray(records, np.float32).reshape(-1, RECORD_FLOATS)
        self._check(self._L.m2s_upload_records(self._h, r.ctypes.data, r.shape[0]))

    def prepass(self, params, records=None, download: bool = True):
        """GaussiansPrepass: cull + covariance projection of the last conversion's records (or of `records`, a CUDA torch
        tensor of shape (n, 24) float32).  `params`: mesh2splat_amd.prepass.PrepassParams.
        -> (visible, quads (visible,24) f32, depths (visible,) f32) in input order; with download=False only `visible`
        (the results stay on the device: device_quads / device_prepass_depths)."""
        from . import prepass as _pp
        pc, keep = _pp.to_c(params)
        ptr, n = None, 0
        if records is not None:
            if not (hasattr(records, "data_ptr") and records.is_cuda and records.is_contiguous()):
                raise ValueError("records must be a contiguous CUDA tensor (n, 24) float32")
            ptr, n = records.data_ptr(), int(records.shape[0])
            if n == 0:                       # (a NULL pointer would mean "the last conversion" to the C entry point)
                return (0, np.empty((0, 24), np.float32), np.empty(0, np.float32)) if download else 0
        vis = C.c_uint64()
        self._check(self._L.m2s_prepass(self._h, C.byref(pc), ptr, n, C.byref(vis)))
        del keep
        if not download:
            return vis.value
        quads = np.empty((vis.value, 24), np.float32)
        depths = np.empty(vis.value, np.float32)
        self._check(self._L.m2s_download_prepass(self._h, quads.ctypes.data, depths.ctypes.data, vis.value))
        return vis.value, quads, depths

    @property
    def device_quads(self) -> int:
        return int(self._L.m2s_device_quads(self._h) or 0)

    @property
    def device_prepass_depths(self) -> int:
        return int(self._L.m2s_device_prepass_depths(self._h) or 0)

    @property
    def last_prepass_ms(self) -> float:
        return float(self._L.m2s_last_prepass_ms(self._h))

    def sort_prepass(self, download: bool = True):
        """RadixSortPass on the last prepass: quads ordered by the raw bits of their view-space depth (ascending, stable).
        -> (n, 24) float32 quads, or just n with download=False."""
        n = C.c_uint64()
        self._check(self._L.m2s_sort_prepass(self._h, C.byref(n)))
        if not download:
            return n.value
        out = np.empty((n.value, 24), np.float32)
        self._check(self._L.m2s_download_sorted_quads(self._h, out.ctypes.data, n.value))
        return out

    def prepass_sorted(self, params, download: bool = True):
        """GaussiansPrepass + RadixSortPass of one frame in one pass over the context's records (m2s_prepass_sorted): the depth sort first,
        as a permutation; the prepass through it.  -> (visible, 24) float32 quads in depth order == prepass() then sort_prepass(), byte
        for byte; or just `visible` with download=False."""
        from . import prepass as _pp
        pc, keep = _pp.to_c(params)
        vis = C.c_uint64()
        self._check(self._L.m2s_prepass_sorted(self._h, C.byref(pc), C.byref(vis)))
        del keep
        if not download:
            return vis.value
        out = np.empty((vis.value, 24), np.float32)
        self._check(self._L.m2s_download_sorted_quads(self._h, out.ctypes.data, vis.value))
        return out

    @property
    def last_sort_prepass_ms(self) -> float:
        return float(self._L.m2s_last_sort_prepass_ms(self._h))

    def set_pipeline(self, name: str):
        """'auto' (single-pass kernel or multi-pass pipeline, chosen per scene and R), 'multipass', or the single-pass
        kernel forced in one of its forms: 'team' (k_fused2, workgroup-cooperative), 'lean' (k_fused3), 'sparse' (k_sparse)."""
        self._check(self._L.m2s_set_pipeline(self._h, {"auto": 0, "multipass": 1, "team": 3, "sparse": 4, "lean": 5}[name]))

    def set_async_lanes(self, lanes: int):
        """2: context-owned submissions alternate between two streams / chains / record buffers and overlap."""
        self._check(self._L.m2s_set_async_lanes(self._h, int(lanes)))

    @property
    def last_pipeline(self) -> str:
        """What the last conversion ran: 'multipass', 'team' (k_fused2), 'lean' (k_fused3) or 'sparse' (k_sparse)."""
        return {0: "none", 1: "multipass", 3: "team", 4: "sparse", 5: "lean"}[self._L.m2s_last_pipeline(self._h)]

    # -- measurement --------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._check(self._L.m2s_set_profiling(self._h, 1 if on else 0))

    def last_kernel_ms(self) -> dict:
        ms = (C.c_float * len(_lib.KERNEL_NAMES))()
        self._check(self._L.m2s_last_kernel_ms(self._h, ms))
        return {k: float(ms[i]) for i, k in enumerate(_lib.KERNEL_NAMES)}


def write_ply(path: str, records: np.ndarray, fmt: int, scale_multiplier: float):
    """parsers::savePlyVector (parsers.cpp:631-651) on a host array of 96-byte records."""
    r = np.ascontiguousarray(records, np.float32)
    if r.ndim != 2 or r.shape[1] != RECORD_FLOATS:
        raise ValueError("records must be (n, 24) float32")
    st = _lib.load().m2s_write_ply(os.fsencode(path), r.ctypes.data, r.shape[0], int(fmt), float(scale_multiplier))
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, f"could not write {path}")


def write_ply_slice(path: str, records: np.ndarray, fmt: int, scale_multiplier: float, first_row: int, total_rows: int):
    """One of several writers of one .ply (m2s_write_ply_slice): rows [first_row, first_row + len(records))."""
    r = np.ascontiguousarray(records, np.float32).reshape(-1, RECORD_FLOATS)
    st = _lib.load().m2s_write_ply_slice(os.fsencode(path), r.ctypes.data, r.shape[0], int(fmt), float(scale_multiplier),
                                         int(first_row), int(total_rows))
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, f"could not write {path}")


# ---------------------------------------------------------------------------------------------------
# reference-shaped interface
# ---------------------------------------------------------------------------------------------------
class RenderContext:
    """The slice of RenderContext (RenderContext.hpp:28-124) that the conversion path touches."""

    def __init__(self, scene: Optional[Scene] = None, resolutionTarget: int = 520, gaussianStd: float = 0.65,
                 device: int = 0):
        self.scene = scene                        # dataMeshAndGlMesh + meshToTextureData
        self.resolutionTarget = int(resolutionTarget)  # main.cpp:26 default quality 0.5 -> 520
        self.gaussianStd = float(gaussianStd)     # main.cpp:26
        self.numberOfGaussians = 0                # written by ConversionPass::execute
        self.device = int(device)
        self.converter: Optional[Converter] = None  # owns gaussianBuffer (the SSBO equivalent)
        self._uploaded_scene = None


class IRenderPass:
    """RenderPass.hpp:11-29."""

    def __init__(self):
        self._enabled = False

    def execute(self, context: RenderContext):
        raise NotImplementedError

    def isEnabled(self) -> bool:
        return self._enabled

    def setIsEnabled(self, enabled: bool):
        self._enabled = bool(enabled)


class ConversionPass(IRenderPass):
    """ConversionPass::execute (ConversionPass.cpp:9-68): synchronous; leaves the Gaussians in the
    context's device buffer and their count in context.numberOfGaussians (not clamped to the cap)."""

    def execute(self, context: RenderContext):
        if context.scene is None:
            raise ValueError("RenderContext.scene is not set (SceneManager::loadModel has not run)")
        if context.converter is None:
            context.converter = Converter(context.device)
        if context._uploaded_scene is not context.scene:
            context.converter.set_resolution_hint(context.resolutionTarget)   # the upload prepares for the conversion below
            context.converter.upload_scene(context.scene)
            context._uploaded_scene = context.scene
        context.numberOfGaussians = context.converter.convert(context.resolutionTarget)


class SceneManager:
    """The export half of SceneManager (SceneManager.cpp:651-678)."""

    def __init__(self, context: RenderContext):
        self.renderContext = context

    def exportPly(self, outputFile: str, exportFormat: int = 0):
        ctx = self.renderContext
        if ctx.converter is None:
            raise RuntimeError("nothing converted yet")
        ctx.converter.export_ply(outputFile, exportFormat, ctx.gaussianStd)
