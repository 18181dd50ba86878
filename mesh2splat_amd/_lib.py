"""ctypes binding of the C ABI (include/m2s.h) implemented by mesh2splat_amd/_build/libm2s_hip.so.

There is deliberately NO fallback: if the HIP library is missing or no device is usable the
product path raises.  (The CPU oracle under oracle/ is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("M2S_LIB_PATH") or os.path.join(_HERE, "_build", "libm2s_hip.so")  # env override: A/B builds

M2S_OK = 0
STATUS_NAMES = {0: "M2S_OK", 1: "M2S_ERR_INVALID", 2: "M2S_ERR_NO_DEVICE", 3: "M2S_ERR_HIP", 4: "M2S_ERR_OOM",
                5: "M2S_ERR_CAPACITY", 6: "M2S_ERR_IO", 7: "M2S_ERR_STATE"}
KERNEL_NAMES = ("count", "scan", "offsets", "emit", "fused")  # M2S_K_*

# every symbol include/m2s.h declares (tests check the .so exports all of them)
EXPORTS = (
    "m2s_abi_version", "m2s_create", "m2s_destroy", "m2s_last_error", "m2s_set_triangle_range", "m2s_upload_scene",
    "m2s_set_max_gaussians", "m2s_convert", "m2s_convert_into", "m2s_convert_submit", "m2s_convert_wait", "m2s_last_pipeline", "m2s_set_keep_positions", "m2s_positions_ready", "m2s_debug_set_launch_counter", "m2s_set_async_lanes", "m2s_num_stored", "m2s_device_records", "m2s_download",
    "m2s_download_triangle_counts", "m2s_write_ply", "m2s_export_ply", "m2s_set_profiling", "m2s_last_kernel_ms",
    "m2s_num_triangles", "m2s_set_pipeline", "m2s_load_glb", "m2s_free_host_scene", "m2s_host_scene_num_meshes",
    "m2s_host_scene_meshes", "m2s_host_scene_mesh_name", "m2s_host_scene_warnings", "m2s_read_ply", "m2s_free_records",
    "m2s_io_last_error", "m2s_sort_by_depth", "m2s_device_sorted_records", "m2s_download_sorted", "m2s_last_sort_ms",
    "m2s_upload_records", "m2s_prepass", "m2s_device_quads", "m2s_device_prepass_depths", "m2s_download_prepass", "m2s_last_prepass_ms",
    "m2s_sort_prepass", "m2s_prepass_sorted", "m2s_device_sorted_quads", "m2s_download_sorted_quads", "m2s_last_sort_prepass_ms",
    "m2s_last_upload_ms", "m2s_write_ply_slice", "m2s_export_ply_slice",
    "m2s_dist_unique_id", "m2s_dist_create", "m2s_dist_destroy", "m2s_dist_rank", "m2s_dist_world", "m2s_dist_last_error",
    "m2s_dist_shard_ranges", "m2s_dist_all_gather_counts", "m2s_dist_publish_count", "m2s_dist_collect_counts",
    "m2s_dist_clamp_to_cap", "m2s_dist_gather_records", "m2s_dist_wait", "m2s_set_records", "m2s_reserve_records", "m2s_prepare",
    "m2s_dist_local_id", "m2s_dist_sort_by_depth", "m2s_device_sorted_keys", "m2s_num_sorted", "m2s_last_resolution",
    "m2s_set_resolution_hint", "m2s_last_warm_ms", "m2s_dist_transport", "m2s_last_sort_stage_ms",
)


class M2SError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class Texture(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32)]


class MeshC(C.Structure):
    _fields_ = [("vertices", C.c_void_p), ("n_vertices", C.c_uint32), ("stride_floats", C.c_uint32),
                ("bbox_min", C.c_float * 3), ("bbox_max", C.c_float * 3), ("base_color", C.c_float * 4),
                ("tex", Texture * 3)]


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "m2s.h")]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        r = subprocess.run(["make", "-C", CSRC, "all"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode:
            raise RuntimeError("building libm2s_hip.so failed:\n" + r.stdout)
    return LIB_PATH


_lib = None


def _preload_torch_hip_runtime():
    """If PyTorch-ROCm is installed, load ITS libamdhip64 first (without importing torch).

    torch wheels bundle their own HIP runtime (same SONAME libamdhip64.so.7 as /opt/rocm's).  A process must
    not end up with two HIP runtimes: if this library pulled in /opt/rocm's copy first, a later `import torch`
    would find no GPU.  Loading torch's copy up front makes both sides share one runtime regardless of import
    order; the stand-alone C++ CLI simply uses /opt/rocm's."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass   # best effort: without torch there is nothing to reconcile


def load():
    """Load the library; raise loudly if it is missing (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the conversion pass)")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i64 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int64
    sigs = {
        "m2s_abi_version": (u32, []),
        "m2s_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "m2s_destroy": (None, [vp]),
        "m2s_last_error": (C.c_char_p, [vp]),
        "m2s_set_triangle_range": (C.c_int, [vp, u64, u64]),
        "m2s_upload_scene": (C.c_int, [vp, C.POINTER(MeshC), u32]),
        "m2s_set_max_gaussians": (C.c_int, [vp, i64]),
        "m2s_convert": (C.c_int, [vp, u32, C.POINTER(u64)]),
        "m2s_convert_into": (C.c_int, [vp, u32, vp, u64, vp, C.POINTER(u64)]),
        "m2s_convert_submit": (C.c_int, [vp, u32, vp, u64, vp]),
        "m2s_convert_wait": (C.c_int, [vp, C.POINTER(u64)]),
        "m2s_upload_records": (C.c_int, [vp, vp, u64]),
        "m2s_prepass": (C.c_int, [vp, vp, vp, u64, C.POINTER(u64)]),
        "m2s_device_quads": (vp, [vp]),
        "m2s_device_prepass_depths": (vp, [vp]),
        "m2s_download_prepass": (C.c_int, [vp, vp, vp, u64]),
        "m2s_last_prepass_ms": (C.c_float, [vp]),
        "m2s_sort_prepass": (C.c_int, [vp, C.POINTER(u64)]),
        "m2s_prepass_sorted": (C.c_int, [vp, vp, C.POINTER(u64)]),
        "m2s_device_sorted_quads": (vp, [vp]),
        "m2s_download_sorted_quads": (C.c_int, [vp, vp, u64]),
        "m2s_last_sort_prepass_ms": (C.c_float, [vp]),
        "m2s_last_pipeline": (C.c_int, [vp]),
        "m2s_set_keep_positions": (C.c_int, [vp, C.c_int]),
        "m2s_positions_ready": (C.c_int, [vp]),
        "m2s_debug_set_launch_counter": (C.c_int, [vp, u32]),
        "m2s_set_async_lanes": (C.c_int, [vp, C.c_int]),
        "m2s_num_stored": (u64, [vp]),
        "m2s_device_records": (vp, [vp]),
        "m2s_download": (C.c_int, [vp, vp, u64]),
        "m2s_download_triangle_counts": (C.c_int, [vp, vp, u64]),
        "m2s_write_ply": (C.c_int, [C.c_char_p, vp, u64, u32, C.c_float]),
        "m2s_export_ply": (C.c_int, [vp, C.c_char_p, u32, C.c_float]),
        "m2s_set_profiling": (C.c_int, [vp, C.c_int]),
        "m2s_last_kernel_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "m2s_num_triangles": (u64, [vp]),
        "m2s_set_pipeline": (C.c_int, [vp, C.c_int]),
        "m2s_load_glb": (C.c_int, [C.c_char_p, C.POINTER(vp)]),
        "m2s_free_host_scene": (None, [vp]),
        "m2s_host_scene_num_meshes": (u32, [vp]),
        "m2s_host_scene_meshes": (C.POINTER(MeshC), [vp]),
        "m2s_host_scene_mesh_name": (C.c_char_p, [vp, u32]),
        "m2s_host_scene_warnings": (C.c_char_p, [vp]),
        "m2s_read_ply": (C.c_int, [C.c_char_p, C.POINTER(vp), C.POINTER(u64), C.POINTER(C.c_int)]),
        "m2s_free_records": (None, [vp]),
        "m2s_io_last_error": (C.c_char_p, []),
        "m2s_sort_by_depth": (C.c_int, [vp, C.POINTER(C.c_float), C.POINTER(u64)]),
        "m2s_device_sorted_records": (vp, [vp]),
        "m2s_download_sorted": (C.c_int, [vp, vp, u64]),
        "m2s_last_sort_ms": (C.c_float, [vp]),
        "m2s_last_upload_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "m2s_write_ply_slice": (C.c_int, [C.c_char_p, vp, u64, u32, C.c_float, u64, u64]),
        "m2s_export_ply_slice": (C.c_int, [vp, C.c_char_p, u32, C.c_float, u64, u64, u64]),
        "m2s_dist_unique_id": (C.c_int, [vp]),
        "m2s_dist_create": (C.c_int, [C.c_int, vp, C.c_int, C.c_int, C.POINTER(vp)]),
        "m2s_dist_destroy": (None, [vp]),
        "m2s_dist_rank": (C.c_int, [vp]),
        "m2s_dist_world": (C.c_int, [vp]),
        "m2s_dist_last_error": (C.c_char_p, [vp]),
        "m2s_dist_shard_ranges": (C.c_int, [C.POINTER(MeshC), u32, u32, C.c_int, C.POINTER(u64), C.POINTER(u64)]),
        "m2s_dist_all_gather_counts": (C.c_int, [vp, u64, C.POINTER(u64), C.POINTER(u64)]),
        "m2s_dist_publish_count": (C.c_int, [vp, u64]),
        "m2s_dist_collect_counts": (C.c_int, [vp, C.POINTER(u64), C.POINTER(u64)]),
        "m2s_dist_clamp_to_cap": (None, [C.POINTER(u64), C.c_int, u64, C.POINTER(u64)]),
        "m2s_dist_gather_records": (C.c_int, [vp, vp, C.POINTER(u64), vp, C.c_int, vp]),
        "m2s_dist_wait": (C.c_int, [vp, vp]),
        "m2s_set_records": (C.c_int, [vp, vp, u64, u32]),
        "m2s_reserve_records": (C.c_int, [vp, u64, C.POINTER(vp)]),
        "m2s_prepare": (C.c_int, [vp, u32]),
        "m2s_dist_local_id": (C.c_int, [C.c_int, vp]),
        "m2s_dist_sort_by_depth": (C.c_int, [vp, vp, C.POINTER(C.c_float), C.POINTER(u64), C.POINTER(u64)]),
        "m2s_device_sorted_keys": (vp, [vp]),
        "m2s_num_sorted": (u64, [vp]),
        "m2s_last_resolution": (u32, [vp]),
        "m2s_set_resolution_hint": (C.c_int, [vp, u32]),
        "m2s_last_warm_ms": (C.c_float, [vp]),
        "m2s_dist_transport": (C.c_char_p, [vp]),
        "m2s_last_sort_stage_ms": (C.c_int, [vp, C.POINTER(C.c_float)]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L
