"""Multi-GPU host logic: one process per GPU, triangle-range shards, RCCL over xGMI via torch.distributed.

The conversion path shards naturally: every triangle is independent (no depth test, no blending,
framebuffer unused: ConversionPass.cpp:45-48); the only shared state of the reference is the append
cursor, which becomes a per-rank counter plus ONE exchange of the counters (offsets of each rank's
block in the merged splat buffer).  Concatenating the per-rank blocks in rank order reproduces the
single-GPU output bit for bit because the device emits records in canonical (triangle, y, x) order.

The record all-gather itself is optional (a sharded consumer, e.g. a per-rank .ply slice writer, only
needs the offsets).  backend "nccl" IS RCCL on ROCm; the same code runs on "gloo" CPU tensors, which
is how the N>1 logic is tested without GPUs.
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

    kind = "C ABI (m2s_dist_*: RCCL opened by libm2s_hip.so)"


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


class TorchExchange:
    """The same exchange through torch.distributed (backend nccl = RCCL).  NOT the product path: bench.py falls back to it when
    the C-ABI communicator cannot be created on some rank (so that a scaling run still yields its line), and says so."""

    kind = "torch.distributed fallback (the C-ABI communicator could not be created)"

    def __init__(self, rank: int, world: int, device: str = "cuda"):
        import collections
        import torch
        self.rank, self.world, self.device = int(rank), int(world), device
        self._torch = torch
        self._pending = collections.deque()

    def close(self):
        pass

    def all_gather_counts(self, total: int):
        while self._pending:
            self.collect_counts()
        self.publish_count(total)
        return self.collect_counts()

    def publish_count(self, total: int):
        import torch.distributed as dist
        torch = self._torch
        mine = torch.tensor([int(total)], dtype=torch.int64, device=self.device)
        allc = torch.zeros(self.world, dtype=torch.int64, device=self.device)
        self._pending.append((dist.all_gather_into_tensor(allc, mine, async_op=True), allc, mine))

    def collect_counts(self):
        work, allc, _ = self._pending.popleft()
        work.wait()
        counts = [int(x) for x in allc.tolist()]
        return counts, offsets_from_counts(counts)

    def gather_records_t(self, mine, counts: Sequence[int], merged, root: int = -1, stream: int = 0):
        import torch.distributed as dist
        off = offsets_from_counts(counts)
        me, W = self.rank, self.world
        receives = root < 0 or root == me
        if receives and counts[me]:
            merged[off[me]: off[me + 1]].copy_(mine[: counts[me]])
        ops = []
        for step in range(1, W):
            dst, src = (me + step) % W, (me - step) % W
            if counts[me] and (root < 0 or root == dst):
                ops.append(dist.P2POp(dist.isend, mine[: counts[me]], dst))
            if counts[src] and receives:
                ops.append(dist.P2POp(dist.irecv, merged[off[src]: off[src + 1]], src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()


def even_ranges(T: int, world: int) -> List[Tuple[int, int]]:
    cuts = [T * r // world for r in range(world + 1)]
    return [(a, b - a) for a, b in zip(cuts[:-1], cuts[1:])]


def all_gather_counts(local_total: int, device=None):
    """The one mandatory exchange: every rank learns every rank's counter. Returns a list of ints."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    allc = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)
    return [int(x) for x in allc.tolist()]


def offsets_from_counts(counts: Sequence[int]) -> List[int]:
    off = [0]
    for c in counts:
        off.append(off[-1] + int(c))
    return off


def all_gather_records(local, counts: Sequence[int], mode: str = "direct"):
    """Concatenate per-rank record blocks (n_r, 24) in rank order on every rank (an all-gather-v; RCCL has no native one).

    mode "direct" (default): every rank sends its block straight to every other rank and receives each peer's block at its
    final offset in the merged buffer — exact sizes, no padding, no compaction copy, and on xGMI's all-to-all point-to-point
    links every link carries exactly one block (grouped isend / irecv = one RCCL group call).
    mode "padded": blocks padded to the largest count for a single all_gather_into_tensor, then compacted."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    assert len(counts) == world and local.shape[0] == counts[rank]
    if mode == "direct":
        off = offsets_from_counts(counts)
        merged = torch.empty((off[-1],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = local.contiguous()
        merged[off[rank]: off[rank + 1]].copy_(local)
        ops = []
        for step in range(1, world):                      # peer order staggered per rank: no two ranks start on the same peer
            dst, src = (rank + step) % world, (rank - step) % world
            if int(counts[rank]):
                ops.append(dist.P2POp(dist.isend, local, dst))
            if int(counts[src]):
                ops.append(dist.P2POp(dist.irecv, merged[off[src]: off[src + 1]], src))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        return merged
    nmax = max(1, max(int(c) for c in counts))
    send = local
    if local.shape[0] != nmax:
        send = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    recv = torch.empty((world * nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send.contiguous())
    if all(int(c) == nmax for c in counts):
        return recv
    return torch.cat([recv[r * nmax: r * nmax + int(counts[r])] for r in range(world)], dim=0)


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


# ---------------------------------------------------------------------------------------------------
# distributed depth sort of the merged splat buffer (SURVEY 8 f-2 at N GPUs; BASELINE config 5): sample sort
# ---------------------------------------------------------------------------------------------------
def depth_keys(records, world_to_view):
    """RadixSortPass keys of (n, 24) records: the raw bits of view-space z = row 2 of world_to_view * (P, 1), as int64 in
    [0, 2^32).  world_to_view: 16 floats, column-major (glm).  Same association as the device key kernel
    (m2s_sort.hip k_depth_keys): ((v02*x + v12*y) + v22*z) + v32, every operation rounded separately."""
    import torch
    v = [float(np.float32(x)) for x in np.asarray(world_to_view, np.float32).reshape(16)]
    x, y, z = records[:, 0], records[:, 1], records[:, 2]
    zz = ((x * v[2] + y * v[6]) + z * v[10]) + v[14]
    return zz.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF


def _stable_sort(keys, payload):
    import torch
    k, order = torch.sort(keys, stable=True)
    return k, payload.index_select(0, order)


def sample_sort(keys, payload, local_sort=None, samples_per_rank: int = 256):
    """Globally stable sort of (key, payload) pairs spread over the ranks of the default process group.

    keys: (n_r,) int64 tensor; payload: (n_r, ...) tensor on the same device.  Afterwards rank r holds the r-th
    contiguous slice of the sequence obtained by stably sorting the rank-major concatenation of all inputs (ties keep
    (source rank, local position) order) — i.e. concatenating the results in rank order IS the single-GPU result.

    Sample sort, one exchange: local sort -> `samples_per_rank` evenly spaced keys per rank, all-gathered -> world-1
    splitters -> every key range goes to one rank (equal keys never straddle ranks) with ONE all-to-all of counts and
    ONE all-to-all-v of keys and of payloads (RCCL point-to-point over xGMI: every rank talks to every other directly,
    which is the pattern the links are built for) -> local stable sort of the received runs.
    `local_sort(keys, payload) -> (keys, payload)` must be stable; default: torch.sort (rocPRIM radix sort on the GPU)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    local_sort = local_sort or _stable_sort
    keys, payload = local_sort(keys, payload)
    if world == 1:
        return keys, payload
    n = int(keys.shape[0])
    dev = keys.device
    s = int(samples_per_rank)
    big = torch.iinfo(torch.int64).max
    mine = torch.full((s + 1,), big, dtype=torch.int64, device=dev)
    take = min(s, n)
    if take:
        pos = ((torch.arange(take, device=dev, dtype=torch.int64) + 1) * n) // (take + 1)
        mine[:take] = keys[pos.clamp_(max=n - 1)]
    mine[s] = take
    allm = torch.empty(world * (s + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allm, mine)
    allm = allm.view(world, s + 1)
    valid = torch.cat([allm[r, : int(allm[r, s])] for r in range(world)])
    valid, _ = torch.sort(valid)
    m = int(valid.shape[0])
    if m:
        cut = (torch.arange(1, world, device=dev, dtype=torch.int64) * m) // world
        splitters = valid[cut.clamp_(max=m - 1)]
    else:
        splitters = torch.full((world - 1,), big, dtype=torch.int64, device=dev)
    # keys < splitters[0] -> rank 0; splitters[j-1] <= key < splitters[j] -> rank j
    bounds = torch.searchsorted(keys, splitters, right=False)
    edges = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), bounds.to(torch.int64),
                       torch.tensor([n], dtype=torch.int64, device=dev)])
    send_counts = (edges[1:] - edges[:-1]).contiguous()
    recv_counts = torch.empty_like(send_counts)
    dist.all_to_all_single(recv_counts, send_counts)
    sc, rc = [int(x) for x in send_counts.tolist()], [int(x) for x in recv_counts.tolist()]
    rk = torch.empty(sum(rc), dtype=keys.dtype, device=dev)
    dist.all_to_all_single(rk, keys.contiguous(), output_split_sizes=rc, input_split_sizes=sc)
    rp = torch.empty((sum(rc),) + tuple(payload.shape[1:]), dtype=payload.dtype, device=dev)
    dist.all_to_all_single(rp, payload.contiguous(), output_split_sizes=rc, input_split_sizes=sc)
    return local_sort(rk, rp)      # runs arrive in source-rank order, each sorted: a stable sort keeps ties in that order
