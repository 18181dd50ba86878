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
the directory's name; M2S_RDZV_DIR, a private directory made by the launcher, takes precedence).  `python -m torch.distributed.run` sets them; so does bench.py's own launcher; nothing here imports torch.
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


def _proc_starttime(pid: int) -> Optional[int]:
    """Start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat), None if it is not visible / not alive."""
    try:
        with open("/proc/%d/stat" % pid, "rb") as f:
            rest = f.read().rsplit(b")", 1)[1].split()
        return int(rest[19])
    except (OSError, IndexError, ValueError):
        return None


def _my_start_wall() -> float:
    """Wall-clock time at which this process started (from /proc; falls back to 'now')."""
    try:
        st = _proc_starttime(os.getpid())
        with open("/proc/uptime") as f:
            up = float(f.read().split()[0])
        return time.time() - (up - st / os.sysconf("SC_CLK_TCK"))
    except Exception:  # noqa: BLE001
        return time.time()


def _secure_dir(path: str):
    """A directory only this user can write: created 0700; an existing one must be a real directory owned by us without group / other
    access (a predictable path under /tmp that somebody else pre-created is refused, not trusted)."""
    try:
        os.makedirs(path, mode=0o700, exist_ok=True)
    except FileExistsError:
        pass
    st = os.lstat(path)
    import stat as _stat
    if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
        raise RuntimeError("rendezvous directory %s is not a directory owned by uid %d: refusing to meet there" % (path, os.getuid()))
    if st.st_mode & 0o077:
        os.chmod(path, 0o700)


def default_key() -> str:
    """What names a launch.  Every rank of one launch sees the same MASTER_ADDR / MASTER_PORT (and TORCHELASTIC_RUN_ID under torchrun) —
    and nothing about who its PARENT is (the key used to contain getppid(), equal for the ranks of a plain torchrun but not
    under a launcher that wraps every rank in its own shell, numactl or exec)."""
    parts = [os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0")]
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    if run_id and run_id != "none":
        parts.append(run_id)
    return "".join(ch if ch.isalnum() or ch in "-." else "_" for ch in "_".join(parts))[:120]


class FileRendezvous:
    """Ranks of one launch meet under <base>/s_<nonce>/.  <base> = $M2S_RDZV_DIR (a launcher's mkdtemp) or
    $TMPDIR/m2s_rdzv_<uid>_<MASTER_ADDR>_<MASTER_PORT>[_<run id>].  Rank 0 opens the SESSION: it removes whatever an earlier launch on
    the same key left behind, creates s_<nonce> and publishes <base>/session = {nonce, its pid and start time}.  The other ranks wait
    for a session whose owner is ALIVE (same pid, same start time: a crashed launch's session file is never taken for this one's), so
    stale `000000_*` files of a dead run can neither be read nor hand anybody an old RCCL id."""

    def __init__(self, rank: int, world: int, key: Optional[str] = None, timeout: float = 300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.base = os.environ.get("M2S_RDZV_DIR") or os.path.join(tempfile.gettempdir(), "m2s_rdzv_%d_%s" % (os.getuid(), key or default_key()))
        _secure_dir(self.base)
        self._seq = 0
        self._t_deadline = None
        sess = os.path.join(self.base, "session")
        if self.rank == 0:
            self._purge(self.base)
            nonce = os.urandom(8).hex()
            self.dir = os.path.join(self.base, "s_" + nonce)
            os.mkdir(self.dir, 0o700)
            tmp = sess + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump({"nonce": nonce, "pid": os.getpid(), "start": _proc_starttime(os.getpid()), "wall": time.time()}, f)
            os.replace(tmp, sess)
        else:
            t0 = time.perf_counter()
            my_start = _my_start_wall()
            while True:
                d = self._live_session(sess, my_start)
                if d is not None:
                    self.dir = d
                    break
                time.sleep(0.002)
                if time.perf_counter() - t0 > self.timeout:
                    raise RendezvousTimeout("rank %d: rank 0 has not opened a session under %s after %.0f s" % (self.rank, self.base, self.timeout))

    @staticmethod
    def _purge(base: str):
        import shutil
        for name in os.listdir(base):
            q = os.path.join(base, name)
            try:
                if os.path.isdir(q) and not os.path.islink(q):
                    shutil.rmtree(q, ignore_errors=True)
                else:
                    os.unlink(q)
            except OSError:
                pass

    def _live_session(self, sess: str, my_start: float) -> Optional[str]:
        try:
            with open(sess) as f:
                info = json.load(f)
            d = os.path.join(self.base, "s_" + str(info["nonce"]))
            if not os.path.isdir(d):
                return None
            pid, start = int(info["pid"]), info.get("start")
            seen = _proc_starttime(pid)
            if seen is not None:
                return d if (start is None or seen == int(start)) else None      # alive, and the same process (not a reused pid)
            if os.path.exists("/proc/%d" % pid) or os.path.exists("/proc/self/stat"):
                # /proc works here and that pid is gone: a dead launch's leftover ... unless rank 0 lives in another pid namespace, where
                # all that can be said is whether the session is about as old as this process
                return d if (not os.path.exists("/proc/%d" % pid) and float(info.get("wall", 0.0)) >= my_start - 180.0 and os.environ.get("M2S_RDZV_FOREIGN_PIDS")) else None
            return d
        except (OSError, ValueError, KeyError):
            return None

    def set_deadline(self, seconds_from_now: Optional[float]):
        """Every wait from now on also ends at this deadline (bench.py: the whole multi-rank bring-up is time-boxed); None lifts it."""
        self._t_deadline = None if seconds_from_now is None else time.perf_counter() + float(seconds_from_now)

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
            now = time.perf_counter()
            if now - t0 > self.timeout or (self._t_deadline is not None and now > self._t_deadline):
                raise RendezvousTimeout("rank %d: no %r from rank %d after %.0f s (%s)" % (self.rank, name, rank, now - t0, self.dir))

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
                self._purge(self.base)
                os.rmdir(self.base)
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
