"""TEST INFRASTRUCTURE: a torch.distributed transcription of the multi-GPU exchange of csrc/m2s_dist.cpp.

The product's exchange (m2s_dist_*: RCCL behind the C ABI) needs GPUs; the CPU-only suite still has to cover the N > 1 LOGIC —
shard plan, counter offsets, global cap, the all-pairs record schedule, the sample sort's splitters and cut points — with
world_size > 1, which it does with these functions on gloo CPU tensors (tests/test_dist_gloo.py).  Nothing outside tests/
imports this module: bench.py and the command line use the C ABI only and fail loudly when its communicator cannot be created
(round 3 kept this code in the package as a fall-back of bench.py; a first multi-GPU run could then have measured it instead
of the product — VERDICT r3).  The multi-PROCESS path of the C++ code itself is covered on the GPU by tests/test_gpu_dist_stub.py.
"""
from typing import List, Sequence

import numpy as np

from mesh2splat_amd.dist import offsets_from_counts


class TorchExchange:
    """The same exchange through torch.distributed: same schedule as m2s_dist_gather_records (every rank sends its block to
    every peer / to one root, staggered peer order, exact sizes)."""

    transport = "torch.distributed (test transcription)"

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


def all_gather_counts(local_total: int, device=None):
    """The one mandatory exchange: every rank learns every rank's counter. Returns a list of ints."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    allc = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine)
    return [int(x) for x in allc.tolist()]


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
