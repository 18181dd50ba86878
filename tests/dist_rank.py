"""One RANK of the multi-process tests of tests/test_gpu_dist_stub.py: a separate Python process with its own HIP runtime, its own
context (m2s_create) and its own communicator (m2s_dist_create over RCCL — in these tests the stand-in of tests/stub_rccl, which
lets the ranks share one GPU).  No torch, no torch.distributed: the 128-byte id travels through a file, like any side channel.

    python tests/dist_rank.py <rank> <world> <work dir> <scenario>

Scenarios (results go to <work dir>/rank<r>.npz; the parent compares them with ONE context converting / sorting the whole scene):
  gather   shard plan -> convert -> counter exchange (blocking, then four in flight) -> records to every rank -> records to one root
  sort     shard plan -> convert -> m2s_dist_sort_by_depth (sample sort: three all-gathers + one all-pairs exchange)
  empty    as gather, but rank 0's triangle range is empty (it sends nothing and still receives)
Exit code 0: fine; 1: an m2s call failed (message on stderr) — what a dead peer must turn into, instead of a hang."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from mesh2splat_amd import _lib, synth                     # noqa: E402
from mesh2splat_amd import dist as m2d                     # noqa: E402
from mesh2splat_amd.converter import Converter             # noqa: E402


def scene_and_view(scenario):
    import camera
    if scenario == "sort_ties":
        return synth.unit_quad(), 96, camera.look_at((0.5, 0.5, 3.0), (0.5, 0.5, 0.0))
    return synth.cube_sphere(20, tex_size=32), 144, camera.look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1))


def merged_buffer(total):
    """A context that only holds a merged buffer (what the command line's rank 0 does): room in its record pool."""
    sink = Converter(0)
    return sink, sink.reserve_records(max(total, 1))


def read_back(sink, ptr, total, R):
    sink.set_records(ptr, total, R)
    out = sink.download() if total else np.zeros((0, 24), np.float32)
    sink.close()
    return out


def main():
    rank, world, work, scenario = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    scene, R, view = scene_and_view(scenario)
    id_file = os.path.join(work, "id.bin")

    def bootstrap(ident):                                   # rank 0 publishes the id, the others wait for the file
        if rank == 0:
            with open(id_file + ".tmp", "wb") as f:
                f.write(ident)
            os.replace(id_file + ".tmp", id_file)
            return ident
        t0 = time.time()
        while not os.path.exists(id_file):
            if time.time() - t0 > 60:
                raise RuntimeError("rank 0 never published the communicator id")
            time.sleep(0.01)
        with open(id_file, "rb") as f:
            return f.read()

    try:
        ex = m2d.RcclExchange(0, rank, world, bootstrap)
        plan = m2d.shard_ranges_native(scene, R, world)
        if scenario == "empty":                             # rank 0 gets nothing, rank 1 gets rank 0's share as well
            plan = [(0, 0), (0, plan[0][1] + plan[1][1])] + plan[2:]
        c = Converter(0)
        c.set_resolution_hint(R)
        c.set_max_gaussians(0)
        c.set_triangle_range(*plan[rank])
        c.upload_scene(scene)
        mine = c.convert(R)
        out = {"transport": np.frombuffer(ex.transport.encode(), np.uint8), "mine": np.int64(mine)}
        if scenario.startswith("sort"):
            n, off = ex.sort_by_depth(c, view)
            out.update(n=np.int64(n), off=np.int64(off), sorted=c.download_sorted() if n else np.zeros((0, 24), np.float32))
        else:
            counts, offs = ex.all_gather_counts(mine)
            assert counts[rank] == mine
            for k in range(4):                              # pipelined counters keep their order
                ex.publish_count(mine + k)
            for k in range(4):
                assert ex.collect_counts()[0][rank] == mine + k
            total = offs[-1]
            sink, merged = merged_buffer(total)
            ex.gather_records(c.device_records or 0, counts, merged, -1, 0)
            ex.wait(0)
            out.update(counts=np.asarray(counts, np.int64), everybody=read_back(sink, merged, total, R))
            root = world - 1
            sink, rooted = merged_buffer(total) if rank == root else (None, 0)
            ex.gather_records(c.device_records or 0, counts, rooted, root, 0)
            ex.wait(0)
            if sink is not None:
                out["rooted"] = read_back(sink, rooted, total, R)
        np.savez(os.path.join(work, f"rank{rank}.npz"), **out)
        c.close()
        ex.close()
    except _lib.M2SError as e:
        print(f"[rank {rank}] {e}", file=sys.stderr, flush=True)
        sys.exit(1)


if __name__ == "__main__":
    main()
