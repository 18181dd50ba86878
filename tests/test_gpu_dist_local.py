"""The multi-GPU exchange with MORE THAN ONE RANK, on the one GPU this box has: the in-process transport of m2s_dist.cpp
(m2s_dist_local_id: ranks are threads, one context each; all-gathers through the group's table, record transfers through
hipMemcpyPeerAsync) runs the same shard plan, counter exchange, record exchange and sample sort as the RCCL transport — only
the four RCCL calls are replaced.  Everything is compared with ONE context converting / sorting the whole scene."""
import threading

import numpy as np
import pytest

import camera
from mesh2splat_amd import dist as m2d
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter

pytestmark = pytest.mark.gpu


def _run_ranks(world, body):
    """body(rank, exchange) on `world` threads; returns the per-rank results, re-raises the first failure."""
    ident = m2d.local_group_id(world)
    out, err = [None] * world, [None] * world

    def main(rank):
        ex = None
        try:
            ex = m2d.RcclExchange(0, rank, world, m2d.local_bootstrap(ident))
            out[rank] = body(rank, ex)
        except BaseException as e:      # noqa: BLE001 (reported below)
            err[rank] = e
        finally:
            if ex is not None:
                ex.close()

    threads = [threading.Thread(target=main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    for e in err:
        if e is not None:
            raise e
    assert all(not t.is_alive() for t in threads)
    return out


def _whole(scene, R):
    c = Converter(0)
    c.upload_scene(scene)
    c.set_max_gaussians(0)
    total = c.convert(R)
    rec = c.download()
    return c, total, rec


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_conversion_counts_and_gather(hiplib, world):
    """Triangle-range shards -> per-rank conversion -> counter exchange (blocking and pipelined) -> record exchange to all
    ranks and to one root: every rank's merged buffer is bit-identical to the single-context conversion."""
    import torch
    scene = synth.cube_sphere(20, tex_size=32)
    R = 144
    c0, total, rec = _whole(scene, R)
    c0.close()
    plan = m2d.shard_ranges_native(scene, R, world)
    assert sum(n for _, n in plan) == scene.n_triangles

    def body(rank, ex):
        c = Converter(0)
        c.set_triangle_range(*plan[rank])
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        mine = c.convert(R)
        counts, offs = ex.all_gather_counts(mine)
        assert counts[rank] == mine and offs[-1] == total
        for k in range(4):                       # pipelined counters keep their order
            ex.publish_count(mine + k)
        for k in range(4):
            assert ex.collect_counts()[0][rank] == mine + k
        merged = torch.zeros((total, 24), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ex.gather_records(c.device_records, counts, merged.data_ptr(), -1, 0)
        ex.wait(0)
        everybody = merged.cpu().numpy()
        rooted = torch.zeros((total, 24), dtype=torch.float32, device="cuda") if rank == world - 1 else None
        torch.cuda.synchronize()
        ex.gather_records(c.device_records, counts, rooted.data_ptr() if rooted is not None else 0, world - 1, 0)
        ex.wait(0)
        res = (counts, everybody, rooted.cpu().numpy() if rooted is not None else None)
        c.close()
        return res

    results = _run_ranks(world, body)
    for rank, (counts, everybody, rooted) in enumerate(results):
        assert sum(counts) == total
        assert np.array_equal(everybody.view(np.uint32), rec.view(np.uint32)), f"rank {rank}: merged buffer differs"
        if rooted is not None:
            assert np.array_equal(rooted.view(np.uint32), rec.view(np.uint32)), "root's merged buffer differs"


@pytest.mark.parametrize("world,scene_kind", [(2, "sphere"), (3, "sphere"), (4, "sphere"), (3, "ties"), (4, "tiny")])
def test_distributed_depth_sort_equals_one_gpu(hiplib, world, scene_kind):
    """m2s_dist_sort_by_depth: the ranks' slices, in rank order, are bit-identical to m2s_sort_by_depth of the merged buffer on
    one context — including the order of equal keys ("ties": a flat quad seen head-on, thousands of identical depths), ranks
    that end up with nothing, and inputs smaller than the sample count ("tiny")."""
    if scene_kind == "sphere":
        scene, R = synth.cube_sphere(20, tex_size=32), 144
        view = camera.look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1))
    elif scene_kind == "ties":
        scene, R = synth.unit_quad(), 96
        view = camera.look_at((0.5, 0.5, 3.0), (0.5, 0.5, 0.0))      # every Gaussian of the quad has the same view-space z
    else:
        scene, R = synth.unit_quad(), 6
        view = camera.look_at((0.3, 0.2, 2.0), (0.5, 0.5, 0.0))
    c0, total, rec = _whole(scene, R)
    want = c0.sort_by_depth(view)
    c0.close()
    plan = m2d.shard_ranges_native(scene, R, world)

    def body(rank, ex):
        c = Converter(0)
        c.set_triangle_range(*plan[rank])
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        c.convert(R)
        n, off = ex.sort_by_depth(c, view)
        got = c.download_sorted()
        assert len(got) == n
        c.close()
        return n, off, got

    results = _run_ranks(world, body)
    assert sum(n for n, _, _ in results) == total
    run = 0
    for rank, (n, off, got) in enumerate(results):
        assert off == run, f"rank {rank}: offset {off}, expected {run}"
        assert np.array_equal(got.view(np.uint32), want[run:run + n].view(np.uint32)), f"rank {rank}: slice differs"
        run += n


def test_group_with_a_missing_rank_can_be_created_and_destroyed(hiplib):
    """Creation does not wait for the other ranks, destruction does not either (the barriers of the exchanges themselves are
    bounded: a rank that never arrives turns into an error on the others, not a hang)."""
    ident = m2d.local_group_id(2)
    ex = m2d.RcclExchange(0, 0, 2, m2d.local_bootstrap(ident))
    assert ex.world == 2 and ex.rank == 0
    ex.close()


@pytest.mark.parametrize("world,bad_rank", [(2, 1), (3, 0)])
def test_a_rank_that_fails_locally_takes_every_rank_out_of_the_sort_together(hiplib, world, bad_rank):
    """a collective is entered by every rank or by none.  One rank's context holds no records (its local sort fails
    with M2S_ERR_STATE); it still contributes to the first all-gather — with a non-zero status word — and EVERY rank returns an
    error from m2s_dist_sort_by_depth instead of waiting for it.  The group stays usable: the same ranks then sort for real."""
    scene, R = synth.cube_sphere(20, tex_size=32), 144
    view = camera.look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1))
    c0, total, _ = _whole(scene, R)
    want = c0.sort_by_depth(view)
    c0.close()
    plan = m2d.shard_ranges_native(scene, R, world)

    def body(rank, ex):
        c = Converter(0)
        c.set_triangle_range(*plan[rank])
        c.upload_scene(scene)
        c.set_max_gaussians(0)
        if rank != bad_rank:
            c.convert(R)                      # the bad rank never converts: nothing to sort
        with pytest.raises(RuntimeError) as e:
            ex.sort_by_depth(c, view)
        first = str(e.value)
        if rank == bad_rank:
            c.convert(R)
        n, off = ex.sort_by_depth(c, view)    # second attempt, every rank has records now
        got = c.download_sorted()
        c.close()
        return first, n, off, got

    results = _run_ranks(world, body)
    for rank, (first, _, _, _) in enumerate(results):
        if rank != bad_rank:
            assert f"rank {bad_rank} reported a failure" in first, first
    assert sum(n for _, n, _, _ in results) == total
    run = 0
    for rank, (_, n, off, got) in enumerate(results):
        assert off == run
        assert np.array_equal(got.view(np.uint32), want[run:run + n].view(np.uint32)), f"rank {rank}: slice differs"
        run += n
