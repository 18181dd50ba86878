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


def all_gather_records(local, counts: Sequence[int]):
    """Concatenate per-rank record blocks (n_r, 24) in rank order on every rank.
    RCCL has no all-gather-v: blocks are padded to the largest count for a single
    all_gather_into_tensor (one collective, large message), then compacted."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    assert len(counts) == world and local.shape[0] == counts[dist.get_rank()]
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
