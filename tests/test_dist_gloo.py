"""CPU, world_size 2 (gloo): the N>1 host logic — shard plan, counter exchange, record all-gather.
Device compute is replaced by the oracle here (tests may use it); on GPUs the same functions run on
RCCL with the HIP path producing each rank's block."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mesh2splat_amd import dist as m2d
from mesh2splat_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, R, cap, result_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle
        scene = synth.sphere_grid(2, n=4, tex_size=16)
        ranges = m2d.shard_ranges(m2d.estimate_fragments(scene, R), world)
        first, count = ranges[rank]
        total, rec, _ = oracle.convert(scene, R, cap=0, tri_first=first, tri_count=count)
        counts = m2d.all_gather_counts(total)
        assert counts[rank] == total
        keep = m2d.clamp_to_cap(counts, cap)
        merged = m2d.all_gather_records(torch.from_numpy(rec[: keep[rank]]), keep)
        np.save(os.path.join(result_dir, f"merged_{rank}.npy"), merged.numpy())
        np.save(os.path.join(result_dir, f"counts_{rank}.npy"), np.asarray(counts))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cap", [(2, 0), (2, 5000), (3, 0)])
def test_sharded_convert_matches_single(tmp_path, oracle, world, cap):
    R = 64
    port = _free_port()
    mp.spawn(_worker, args=(world, port, R, cap, str(tmp_path)), nprocs=world, join=True)
    scene = synth.sphere_grid(2, n=4, tex_size=16)
    total, full, _ = oracle.convert(scene, R, cap=cap)
    for r in range(world):
        merged = np.load(tmp_path / f"merged_{r}.npy")
        counts = np.load(tmp_path / f"counts_{r}.npy")
        assert counts.sum() == total
        assert merged.shape == full.shape
        assert np.array_equal(merged.view(np.uint32), full.view(np.uint32))


def test_shard_ranges_are_contiguous_and_balanced(oracle):
    scene = synth.sphere_grid(2, n=6)
    R = 128
    est = m2d.estimate_fragments(scene, R)
    exact = oracle.count_per_triangle(scene, R).astype(np.float64)
    assert abs(est.sum() - exact.sum()) / exact.sum() < 0.05        # the estimate tracks the rasteriser
    for world in (1, 2, 4, 8):
        rg = m2d.shard_ranges(est, world)
        assert len(rg) == world and rg[0][0] == 0 and sum(c for _, c in rg) == scene.n_triangles
        for (a, ca), (b, _) in zip(rg[:-1], rg[1:]):
            assert a + ca == b
        loads = [exact[a:a + c].sum() + 0.25 * c for a, c in rg]
        assert max(loads) <= 1.25 * (sum(loads) / world) + 64
    assert m2d.even_ranges(10, 3) == [(0, 3), (3, 3), (6, 4)]
    assert m2d.shard_ranges(np.zeros(0, np.float32), 4) == [(0, 0)] * 4
    assert m2d.offsets_from_counts([3, 0, 5]) == [0, 3, 3, 8]
    assert m2d.clamp_to_cap([3, 4, 5], 6) == [3, 3, 0] and m2d.clamp_to_cap([3, 4], 0) == [3, 4]
