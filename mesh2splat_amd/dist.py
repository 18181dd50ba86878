"""Multi-GPU host logic: one process per GPU, triangle-range shards, RCCL over xGMI behind the C ABI (m2s_dist_*).

The conversion path shards naturally: every triangle is independent (no depth test, no blending,
framebuffer unused: ConversionPass.cpp:45-48); the only shared state of the reference is the append
cursor, which becomes a per-rank counter plus ONE exchange of the counters (offsets of each rank's
block in the merged splat buffer).  Concatenating the per-rank blocks in rank order reproduces the
single-GPU output bit for bit because the device emits records in canonical (triangle, y, x) order.

The record all-gather itself is optional (a sharded consumer, e.g. a per-rank .ply slice writer, only
needs the offsets).  This module binds the C ABI (RcclExchange) and keeps the host-side plan helpers; there is exactly ONE
implementation of the exchange — csrc/m2s_dist.cpp.  (The torch.distributed transcription that the CPU-only gloo tests use
to check plan, offsets, cap clamp and sort schedule lives with the tests: tests/torch_twin.py.)
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from .scene import Scene


def estimate_fragments(scene: Scene, R: int) -> np.ndarray:
    """Cheap per-triangle estimate of the fragment count: area of the triangle projected on its dominant
    axis plane, in pixels of the R x R viewport (what the rasteriser covers up to boundary effects)."""
    out = []
    for m in scene.meshes:
        v = m.vertices.reshape(-1, 3, m.stride)[:, :, 0:3].astype(np.float64)
        n = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])          # 2 * area * normal
        an = np.abs(n)
        ext = (np.asarray(m.bbox_max, np.float64) - np.asarray(m.bbox_min, np.float64))
        first = (an[:, 0] > an[:, 1]) & (an[:, 0] > an[:, 2])
        second = ~first & (an[:, 1] > an[:, 2])
        third = ~first & ~second
        rng = np.where(first, np.maximum(ext[1], ext[2]), np.where(second, np.maximum(ext[0], ext[2]),
                                                                   np.maximum(ext[0], ext[1])))
        proj2 = np.where(first, an[:, 0], np.where(second, an[:, 1], an[:, 2]))   # 2 * projected area
        with np.errstate(divide="ignore", invalid="ignore"):
            px = 0.5 * proj2 / (rng * rng) * float(R) * float(R)
        out.append(np.nan_to_num(px, nan=0.0, posinf=0.0).astype(np.float32))
        del third
    return np.concatenate(out) if out else np.zeros(0, np.float32)


def shard_ranges(weights: np.ndarray, world: int, per_triangle_cost: float = 0.25) -> List[Tuple[int, int]]:
    """Cut [0, T) into `world` contiguous (first, count) ranges with ~equal sum of
    (estimated fragments + per_triangle_cost): emitting costs per fragment, setup costs per triangle."""
    T = int(len(weights))
    if world <= 1 or T == 0:
        return [(0, T)] + [(T, 0)] * (max(world, 1) - 1)
    cost = np.cumsum(np.asarray(weights, np.float64) + per_triangle_cost)
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cost, cost[-1] * r / world, side="left")))
    cuts.append(T)
    cuts = np.maximum.accumulate(np.minimum(cuts, T))
    return [(int(a), int(b - a)) for a, b in zip(cuts[:-1], cuts[1:])]


def shard_ranges_native(scene: Scene, R: int, world: int) -> List[Tuple[int, int]]:
    """The same plan from the C ABI (m2s_dist_shard_ranges): what the C++ command line and any non-Python host use."""
    import ctypes as C
    from . import _lib
    from .converter import marshal_scene
    L = _lib.load()
    arr, keep = marshal_scene(scene)
    first = (C.c_uint64 * world)()
    count = (C.c_uint64 * world)()
    st = L.m2s_dist_shard_ranges(arr, scene.n_meshes, int(R), int(world), first, count)
    del keep
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, L.m2s_dist_last_error(None).decode())
    return [(int(first[r]), int(count[r])) for r in range(world)]


class RcclExchange:
    """The multi-GPU exchange through the C ABI (m2s_dist_*: RCCL opened by libm2s_hip.so itself, no torch.distributed in
    the data path).  `bootstrap` hands rank 0's 128-byte communicator id to the other ranks: a callable
    bytes-or-None -> bytes (e.g. a torch.distributed / MPI broadcast, a file, a pipe)."""

    def __init__(self, device: int, rank: int, world: int, bootstrap):
        import ctypes as C
        from . import _lib
        self._C, self._lib, self._L = C, _lib, _lib.load()
        ident = None
        if rank == 0 and not getattr(bootstrap, "local_group", False) and not getattr(bootstrap, "provides_id", False):
            buf = (C.c_uint8 * 128)()
            st = self._L.m2s_dist_unique_id(buf)
            if st != _lib.M2S_OK:
                raise _lib.M2SError(st, self._L.m2s_dist_last_error(None).decode())
            ident = bytes(buf)
        ident = bootstrap(ident)
        assert len(ident) == 128
        h = C.c_void_p()
        st = self._L.m2s_dist_create(int(device), C.c_char_p(ident), int(rank), int(world), C.byref(h))
        if st != _lib.M2S_OK:
            raise _lib.M2SError(st, self._L.m2s_dist_last_error(None).decode())
        self._h, self.rank, self.world = h, int(rank), int(world)

    @staticmethod
    def unique_id() -> bytes:
        """m2s_dist_unique_id (rank 0): the 128 bytes the other ranks need.  Separate from the constructor so that a launcher can
        tell the other ranks about a failure HERE before anybody enters a collective (hand the id to the constructor through a
        bootstrap callable with the attribute provides_id = True)."""
        import ctypes as C
        from . import _lib
        L = _lib.load()
        buf = (C.c_uint8 * 128)()
        st = L.m2s_dist_unique_id(buf)
        if st != _lib.M2S_OK:
            raise _lib.M2SError(st, L.m2s_dist_last_error(None).decode())
        return bytes(buf)

    def _check(self, st):
        if st != self._lib.M2S_OK:
            raise self._lib.M2SError(st, self._L.m2s_dist_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.m2s_dist_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_gather_counts(self, total: int):
        counts = (self._C.c_uint64 * self.world)()
        offs = (self._C.c_uint64 * (self.world + 1))()
        self._check(self._L.m2s_dist_all_gather_counts(self._h, int(total), counts, offs))
        return [int(x) for x in counts], [int(x) for x in offs]

    def publish_count(self, total: int):
        self._check(self._L.m2s_dist_publish_count(self._h, int(total)))

    def collect_counts(self):
        counts = (self._C.c_uint64 * self.world)()
        offs = (self._C.c_uint64 * (self.world + 1))()
        self._check(self._L.m2s_dist_collect_counts(self._h, counts, offs))
        return [int(x) for x in counts], [int(x) for x in offs]

    def gather_records(self, d_mine: int, counts: Sequence[int], d_merged: int, root: int = -1, stream: int = 0):
        c = (self._C.c_uint64 * self.world)(*[int(x) for x in counts])
        self._check(self._L.m2s_dist_gather_records(self._h, self._C.c_void_p(d_mine or None), c, self._C.c_void_p(d_merged or None),
                                                    int(root), self._C.c_void_p(stream or None)))


    def gather_records_t(self, mine, counts: Sequence[int], merged, root: int = -1, stream: int = 0):
        """Tensor form (what bench.py calls, so that the torch.distributed stand-in below is interchangeable)."""
        self.gather_records(mine.data_ptr(), counts, merged.data_ptr() if merged is not None else 0, root, stream)

    def wait(self, stream: int = 0):
        self._check(self._L.m2s_dist_wait(self._h, self._C.c_void_p(stream or None)))

    def sort_by_depth(self, converter, world_to_view):
        """m2s_dist_sort_by_depth: sample sort of the ranks' current records (collective).  -> (n, offset): this rank's slice
        of the globally sorted sequence is `converter`'s sorted buffer (converter.download_sorted())."""
        import numpy as np
        m = np.ascontiguousarray(np.asarray(world_to_view, np.float32).T.reshape(16))   # column-major, like glm
        n, off = self._C.c_uint64(), self._C.c_uint64()
        self._check(self._L.m2s_dist_sort_by_depth(self._h, converter._h, m.ctypes.data_as(self._C.POINTER(self._C.c_float)),
                                                   self._C.byref(n), self._C.byref(off)))
        return int(n.value), int(off.value)

    @property
    def transport(self) -> str:
        """m2s_dist_transport: "in-process", "rccl" or "rccl:<M2S_RCCL_PATH>"."""
        return self._L.m2s_dist_transport(self._h).decode()


def local_bootstrap(ident: bytes):
    """bootstrap argument of RcclExchange for a rank of an in-process group (see local_group_id)."""
    def hand_over(_):
        return ident
    hand_over.local_group = True      # rank 0 must not ask RCCL for an id
    return hand_over


def local_group_id(world: int) -> bytes:
    """m2s_dist_local_id: the id of an in-process group (ranks = threads of this process, one Converter each).  Hand the
    same bytes to every rank's RcclExchange(device, rank, world, local_bootstrap(ident))."""
    import ctypes as C
    from . import _lib
    L = _lib.load()
    buf = (C.c_uint8 * 128)()
    st = L.m2s_dist_local_id(int(world), buf)
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, L.m2s_dist_last_error(None).decode())
    return bytes(buf)


def even_ranges(T: int, world: int) -> List[Tuple[int, int]]:
    cuts = [T * r // world for r in range(world + 1)]
    return [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]


def offsets_from_counts(counts: Sequence[int]) -> List[int]:
    off = [0]
    for c in counts:
        off.append(off[-1] + int(c))
    return off


def clamp_to_cap(counts: Sequence[int], cap: int) -> List[int]:
    """Global u_maxGaussians semantics for a sharded run: the merged buffer keeps the first `cap`
    records in rank order (cap 0 = unlimited); returns how many records each rank contributes."""
    if not cap:
        return [int(c) for c in counts]
    out, left = [], int(cap)
    for c in counts:
        k = min(int(c), left)
        out.append(k)
        left -= k
    return out
