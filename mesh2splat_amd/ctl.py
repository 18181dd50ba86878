"""Control plane of a multi-rank run on ONE node (one process per GPU), without torch.distributed.

Two stages, one object:
  * until the C-ABI communicator exists (m2s_dist_create: RCCL behind libm2s_hip.so) the ranks meet in a directory —
    FileRendezvous: every collective is "write my file, read everybody's", files are published by rename, names carry a
    sequence number, so a slow rank never reads a later round.  That is all a bring-up needs: the 128-byte id from rank 0,
    one agreement on "did every rank get a communicator", the error texts if not;
  * afterwards every barrier / reduction of the run goes through the communicator itself (m2s_dist_all_gather_counts: one
    8-byte word per rank over RCCL — opaque to the library), so the timed region is bracketed by the same transport it measures
    and the process holds ONE rendezvous mechanism and ONE user of RCCL.

The launcher contract is the usual environment: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT (only used as part of
the directory's name).  `python -m torch.distributed.run` sets them; so does bench.py's own launcher; nothing here imports torch.
"""
from __future__ import annotations

import json
import os
import struct
import tempfile
import time
from typing import List, Optional


class RendezvousTimeout(RuntimeError):
    pass


class FileRendezvous:
    def __init__(self, rank: int, world: int, key: Optional[str] = None, timeout: float = 300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        if key is None:
            # all ranks of one launch share their parent (the launcher); a relaunch on the same port gets a new directory
            key = "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        self.dir = os.environ.get("M2S_RDZV_DIR") or os.path.join(tempfile.gettempdir(), "m2s_rdzv_" + key)
        os.makedirs(self.dir, exist_ok=True)
        self._seq = 0

    def _path(self, seq: int, name: str, rank: int) -> str:
        return os.path.join(self.dir, "%06d_%s.%d" % (seq, name, rank))

    def _put(self, seq: int, name: str, data: bytes):
        p = self._path(seq, name, self.rank)
        tmp = p + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, p)

    def _get(self, seq: int, name: str, rank: int) -> bytes:
        p = self._path(seq, name, rank)
        t0 = time.perf_counter()
        spins = 0
        while True:
            try:
                with open(p, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                pass
            spins += 1
            if spins > 2000:
                time.sleep(0.0005)
            if time.perf_counter() - t0 > self.timeout:
                raise RendezvousTimeout("rank %d: no %r from rank %d after %.0f s (%s)" % (self.rank, name, rank, self.timeout, self.dir))

    def allgather(self, name: str, data: bytes) -> List[bytes]:
        seq = self._seq
        self._seq += 1
        self._put(seq, name, data)
        return [data if r == self.rank else self._get(seq, name, r) for r in range(self.world)]

    def broadcast(self, name: str, data: Optional[bytes], src: int = 0) -> bytes:
        seq = self._seq
        self._seq += 1
        if self.rank == src:
            self._put(seq, name, data or b"")
            return data or b""
        return self._get(seq, name, src)

    def barrier(self, name: str = "barrier"):
        self.allgather(name, b"1")

    def close(self):
        """Last collective of the run: after it rank 0 removes the directory."""
        try:
            self.barrier("close")
        except RendezvousTimeout:
            pass
        if self.rank == 0:
            try:
                for f in os.listdir(self.dir):
                    try:
                        os.unlink(os.path.join(self.dir, f))
                    except OSError:
                        pass
                os.rmdir(self.dir)
            except OSError:
                pass


class Ctl:
    """barrier / max / sum / gather for bench.py and the rank scripts; world == 1: everything is the identity."""

    def __init__(self, rank: int, world: int, timeout: float = 300.0):   # (a rank's first `import torch` on a fresh box can take two minutes)
        self.rank, self.world = int(rank), int(world)
        self.rdzv = FileRendezvous(rank, world, timeout=timeout) if world > 1 else None
        self.exchange = None

    def use(self, exchange):
        """From now on barriers and 64-bit gathers go through the C-ABI communicator (mesh2splat_amd.dist.RcclExchange)."""
        self.exchange = exchange

    def gather_u64(self, v: int) -> List[int]:
        v = int(v) & 0xFFFFFFFFFFFFFFFF
        if self.world == 1:
            return [v]
        if self.exchange is not None:
            counts, _ = self.exchange.all_gather_counts(v)
            return counts
        return [struct.unpack("<Q", b)[0] for b in self.rdzv.allgather("u64", struct.pack("<Q", v))]

    def barrier(self):
        if self.world > 1:
            self.gather_u64(0)

    def max_float(self, x: float) -> float:
        return max(self.gather_u64(int(round(float(x) * 1e9)))) / 1e9

    def sum_int(self, v: int) -> int:
        return sum(self.gather_u64(v))

    def min_int(self, v: int) -> int:
        return min(self.gather_u64(v))

    def gather_obj(self, name: str, obj) -> list:
        """Small JSON-serialisable objects (error texts, per-rank reports): always through the directory."""
        if self.world == 1:
            return [obj]
        return [json.loads(b.decode()) for b in self.rdzv.allgather(name, json.dumps(obj).encode())]

    def broadcast_bytes(self, name: str, data: Optional[bytes]) -> bytes:
        if self.world == 1:
            return data or b""
        return self.rdzv.broadcast(name, data)

    def close(self):
        if self.rdzv is not None:
            self.rdzv.close()
            self.rdzv = None
