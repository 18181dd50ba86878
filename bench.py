#!/usr/bin/env python
"""bench.py — Gaussians/sec of the mesh -> 3DGS conversion pass on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: started by a launcher that sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* for one process per GPU (the
driver's torchrun does), or typed as is — bench.py then starts the N ranks itself.  The ranks meet in a directory for the 128-byte
RCCL id (mesh2splat_amd/ctl.py); every collective after that — data path, barriers, timing reductions — is the C-ABI communicator's
(m2s_dist_*: RCCL behind libm2s_hip.so).  No second process group next to it.
`--gpus N --dry-scale` runs that whole schedule on ONE GPU through the RCCL stand-in of tests/stub_rccl (CI; not a transport measurement).

A "step" is one execution of the conversion pass (== ConversionPass::execute) on geometry and textures already resident
in HBM.  Headline workload (BASELINE.json configs[2], the one the metric is quoted on): I-3 = cube-sphere n=289
(1 002 252 triangles), three procedural 2048^2 RGBA8 maps, density R = 1024.

N > 1, headline = WEAK scaling: the scene holds N such meshes (co-located, so every mesh has the same cumulative bbox and
the same fragment count), sharded by triangle range one mesh per rank; the only collective in the timed step is the
8-byte all-gather of the per-rank counters (RCCL through the C ABI, m2s_dist_*), which gives every rank its offset in the
merged splat buffer.  Reported next to it, outside `value`:
  gather           the all-pairs record exchange over xGMI that concatenates the per-rank blocks on every rank;
  strong_scaling   ONE scene (C3, and the C4 stand-in) cut into N fragment-balanced triangle ranges: convert + counter
                   exchange ("no_gather": what per-rank .ply slice writers need) and convert + record exchange ("gather").
N == 1 adds: cold_path (first call on a fresh context, densities never seen before, three rotating scene copies that
do not fit the Infinity Cache), extra_workloads (C2 stand-in, an 11.6- and a 26-fragments-per-triangle mesh, the C4 stand-in, the
heterogeneous synth.sponza_like scene, BASELINE config 5 at full size; each with kernel times, one blocking call and roofline fractions —
flat in roofline.workloads), overlapped (two asynchronous lanes),
viewer_passes, cpu_baseline.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# the host driver only supports dmabuf IPC: without this RCCL's multi-process bring-up fails with `hipIpcGetMemHandle: invalid argument`.
# Set before any HIP runtime is loaded, whoever launched this rank (the driver's torchrun exports it already; a bare launcher may not).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (cube-sphere n, texture size, R)
    "c3": (289, 2048, 1024),   # BASELINE configs[2] — default
    "c2": (76, 2048, 512),     # SciFiHelmet stand-in (I-2)
    "small": (24, 256, 256),   # CI-sized
    "c4": ("grid", 1024, 1024),  # Sponza stand-in (I-4): 64 meshes x cube-sphere n=18 (radius 0.12 s: 6.6 M Gaussians, under the 7 M cap), 64 materials
    "mid": (94, 2048, 1024),   # one mesh, 106 032 triangles, ~26 fragments per triangle (Sponza-like triangle sizes)
    "band": (140, 2048, 1024),  # one mesh, 235 200 triangles, 11.6 fragments per triangle: the band where AUTO runs the team kernel in batches of 40 triangles
    "hetero": ("sponza_like", 0, 1024),  # synth.sponza_like: 64 meshes, 266 840 triangles from 0.1 to 500 000 px each, maps of 256^2 ... 2048^2, some materials without
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the post-run record all-gather measurement")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget for the CPU baseline leg")
    ap.add_argument("--no-viewer-extra", action="store_true", help="skip the prepass / depth-sort measurement after the timed region")
    ap.add_argument("--no-overlap-extra", action="store_true",
                    help="skip the two-lane overlapped submission measured after the timed region (reported as 'overlapped'); use it "
                         "under rocprofv3 --kernel-trace so that the trace holds only isolated launches")
    ap.add_argument("--no-c5", action="store_true", help="skip BASELINE config 5 at full size (50 M triangles: ~15 s of host-side generation)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking m2s_convert per step instead of the two-deep pipeline")
    ap.add_argument("--no-extra-workloads", action="store_true", help="skip the C2 / mid-size / C4 lines (and the C4 strong-scaling run at N > 1)")
    ap.add_argument("--extras-timeout", type=float, default=420.0, help="N > 1: seconds after which the multi-GPU extras are abandoned and the headline line is printed")
    ap.add_argument("--no-strong-scaling", action="store_true", help="N > 1: skip the strong-scaling section (one scene cut into N ranges)")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-path measurements (first call, new R, rotating scene copies)")
    ap.add_argument("--dry-scale", action="store_true",
                    help="with --gpus N: the whole N-process schedule (bring-up, weak-scaling loop, record exchange, strong scaling) on ONE "
                         "GPU through the RCCL stand-in of tests/stub_rccl: what a CI box can run; the line has the shape of a real "
                         "SCALE record (scale_record) and says dry_scale: true.  Not a measurement of a transport")
    ap.add_argument("--one-device", action="store_true",
                    help="tests: every rank on device 0 (several processes share one GPU; needs an RCCL stand-in that allows it: M2S_RCCL_PATH)")
    ap.add_argument("--pipeline", default=None, choices=["auto", "multipass", "team", "sparse", "lean"],
                    help="A/B: force a pipeline setting for the headline workload (default: the library's AUTO)")
    ap.add_argument("--reps", type=int, default=0,
                    help="repetitions of the timed region (each: EXACTLY --steps conversions between barrier + synchronize); ms_per_step / value "
                         "are the MEDIAN repetition, all of them are in ms_per_step_reps.  Default: 5 when --steps < 100 (a 2.3 ms window "
                         "must not decide the headline), else 1")
    ap.add_argument("--bringup-timeout", type=float, default=100.0, help="N > 1: seconds the ranks get to meet and create the communicator; "
                    "then every rank leaves non-zero and rank 0 prints scale_record with the error text")
    ap.add_argument("--headline-timeout", type=float, default=240.0, help="N > 1: seconds for the weak-scaling headline after the bring-up")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end leg (the command line: load + upload + convert + export)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (RCCL init, convert_into, counter all-gather) even with 1 rank")
    return ap.parse_args()


def usable_cores() -> int:
    """Cores this process may actually use: affinity mask, further limited by a cgroup CPU quota if any."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except Exception:  # noqa: BLE001
        pass
    return "unknown"


def cpu_baseline(scene_one_mesh, R, budget_s, gpu_total=None):
    """Oracle (CPU port of the reference path) on the host cores: same workload, same timed region as the
    GPU (geometry + textures + mip chains resident, output buffer allocated -> records + count)."""
    from oracle import oracle
    oracle.build()
    cores = usable_cores()
    prep = oracle.PreparedScene(scene_one_mesh)
    total, out = prep.convert(R, n_threads=cores)             # warm-up, sizes the output buffer
    res = {}
    for name, threads, max_reps in (("all_cores", cores, 5), ("one_core", 1, 2)):
        best, reps, t_start = None, 0, time.perf_counter()
        while reps < max_reps and (time.perf_counter() - t_start) < budget_s / 2:
            t0 = time.perf_counter()
            prep.convert(R, n_threads=threads, out=out)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
            reps += 1
        res[name] = (best, reps)
    prep.close()
    best, reps = res["all_cores"]
    port = {"value": total / best, "unit": "Gaussians/s", "cores": cores, "kind": "port",
            "sample": f"full workload ({total} Gaussians), oracle with OpenMP over triangles (count pass + emit pass), "
                      f"mip chains and output buffer prepared outside the timed region, best of {reps}",
            "ms_per_mesh": best * 1e3,
            "single_thread": {"value": total / res["one_core"][0], "ms_per_mesh": res["one_core"][0] * 1e3}}
    port["counter_equals_gpu"] = None if gpu_total is None else bool(total == gpu_total)
    host = {"cpu_model": cpu_model(), "nproc": os.cpu_count(), "usable_cores": cores}
    ref = reference_baseline(scene_one_mesh, R, total if gpu_total is None else gpu_total)
    if ref is None:
        port["host"] = host
        return port
    ref["port"] = port          # the multi-core figure of our own CPU restatement, for scale
    ref["host"] = host
    return ref


def reference_baseline(scene_one_mesh, R, expect_total):
    """The reference ITSELF on the host, two ways (both built from /root/reference by oracle/Makefile; None if absent):
      * oracle/_ref/ref_gl_check — the reference's loader, ConversionPass::execute and its UNMODIFIED converter{VS,GS,FS}.glsl on
        a real OpenGL 4.6 implementation: Mesa llvmpipe, the GPU-less software GL of this box (north star: "the reference's
        own GL path timed on the same box's host GPU-less software-GL"), on as many threads as the process may use.  This is
        cpu_baseline.value when it runs;
      * oracle/_ref/ref_pipeline_check — the same host code with the shaders compiled as C++ through glm on a minimal
        software GL whose fixed-function stages are the oracle's: one thread (reported next to it)."""
    import subprocess
    import tempfile
    from mesh2splat_amd import gltf_io
    gl_exe = os.path.join(ROOT, "oracle", "_ref", "ref_gl_check")
    cpp_exe = os.path.join(ROOT, "oracle", "_ref", "ref_pipeline_check")
    have = [e for e in (gl_exe, cpp_exe) if os.path.isfile(e) and os.access(e, os.X_OK)]
    if not have:
        return None
    out, cpp = None, None
    cores = usable_cores()
    try:
        with tempfile.TemporaryDirectory() as d:
            glb = os.path.join(d, "workload.glb")
            gltf_io.write_glb(scene_one_mesh, glb, indexed=False)
            if gl_exe in have:
                env = dict(os.environ, LP_NUM_THREADS=str(min(cores, 16)))
                r = subprocess.run([gl_exe, glb, str(int(R)), "-"], capture_output=True, text=True, timeout=900, env=env)
                if r.returncode == 0:
                    info = json.loads(r.stdout.strip().splitlines()[-1])
                    sec = info["execute_ms"] * 1e-3
                    out = {"value": info["counter"] / sec, "unit": "Gaussians/s", "cores": min(cores, 16), "kind": "reference",
                           "sample": f"full workload ({info['counter']} Gaussians): the reference's ConversionPass::execute with its unmodified GLSL "
                                     f"shaders on {info['gl_renderer']} (OpenGL {info['gl_version']}), the box's software GL, LP_NUM_THREADS={min(cores, 16)}, one run",
                           "ms_per_mesh": info["execute_ms"], "counter": info["counter"],
                           "counter_equals_gpu": bool(info["counter"] == expect_total)}
            if cpp_exe in have:
                r = subprocess.run([cpp_exe, glb, str(int(R)), "-"], capture_output=True, text=True, timeout=900)
                if r.returncode == 0:
                    info = json.loads(r.stdout.strip().splitlines()[-1])
                    cpp = {"value": info["counter"] / (info["execute_ms"] * 1e-3), "unit": "Gaussians/s", "cores": 1, "ms_per_mesh": info["execute_ms"],
                           "counter": info["counter"], "counter_equals_gpu": bool(info["counter"] == expect_total),
                           "what": "the reference's execute() with its three shaders compiled as C++ through glm (oracle/ref_pipeline_check), one thread"}
    except Exception:  # noqa: BLE001
        pass
    if out is None and cpp is not None:
        out = dict(cpp, kind="reference", sample="full workload: " + cpp["what"])
    elif out is not None and cpp is not None:
        out["shaders_as_cpp_one_thread"] = cpp
    return out


def frame_passes(conv, p, n_records, reps=6):
    """One viewer frame on the context's records, two ways (wall clock of the blocking calls, ms; median of `reps` after a first one):
      two_calls  m2s_prepass (input order) + m2s_sort_prepass  == GaussiansPrepass::execute + RadixSortPass::execute (keys, sort, 96-byte gather)
      fused      m2s_prepass_sorted: the depth sort first, as a permutation; the prepass through it — same bytes out (tests/test_gpu_prepass.py)."""
    import numpy as np
    from dataclasses import replace
    p = replace(p, arrival_order=False)
    two, fused, st2, stf = [], [], [], []
    for k in range(reps + 1):
        t0 = time.perf_counter()
        vis = conv.prepass(p, download=False)
        t1 = time.perf_counter()
        conv.sort_prepass(download=False)
        t2 = time.perf_counter()
        two.append((t2 - t0) * 1e3)
        st2.append({"prepass_kernel": conv.last_prepass_ms, "sort_prepass": conv.last_sort_prepass_ms, "prepass_call": (t1 - t0) * 1e3})
        t0 = time.perf_counter()
        vis_f = conv.prepass_sorted(p, download=False)
        fused.append((time.perf_counter() - t0) * 1e3)
        s_ = conv.last_sort_stage_ms
        stf.append({"keys": s_["keys"], "radix_sort": s_["radix_sort"], "prepass_through_permutation": s_["gather"]})
        assert vis_f == vis
    med = lambda xs: float(np.median(xs[1:]))
    a, b = med(two), med(fused)
    return {"records": int(n_records), "visible": int(vis), "two_calls_ms": a, "fused_ms": b, "fused_over_two_calls": b / a,
            "two_calls_stages_ms": {k: med([x[k] for x in st2]) for k in st2[0]}, "fused_stages_ms": {k: med([x[k] for x in stf]) for k in stf[0]},
            "what": "m2s_prepass + m2s_sort_prepass against m2s_prepass_sorted (one pass over the records), blocking calls, wall clock"}


def viewer_camera():
    import math
    import numpy as np

    def look_at(eye, center):
        eye, center, up = np.asarray(eye, float), np.asarray(center, float), np.array([0.0, 1.0, 0.0])
        f = center - eye
        f /= np.linalg.norm(f)
        sv = np.cross(f, up)
        sv /= np.linalg.norm(sv)
        u = np.cross(sv, f)
        m = np.eye(4)
        m[0, 0], m[1, 0], m[2, 0] = sv
        m[0, 1], m[1, 1], m[2, 1] = u
        m[0, 2], m[1, 2], m[2, 2] = -f
        m[3, 0], m[3, 1], m[3, 2] = -sv.dot(eye), -u.dot(eye), f.dot(eye)
        return m.astype(np.float32)

    t = math.tan(math.radians(45.0) / 2)
    proj = np.zeros((4, 4), np.float32)
    proj[0, 0], proj[1, 1] = 1 / (16 / 9 * t), 1 / t
    proj[2, 2], proj[2, 3], proj[3, 2] = -(100 + 0.01) / (100 - 0.01), -1.0, -(2 * 100 * 0.01) / (100 - 0.01)
    return look_at, proj


def viewer_extra(conv, R, total):
    """GaussiansPrepass + RadixSortPass on the records of the last conversion: kernel ms (HIP events), algorithmic bytes
    96*n read + 100*visible written, fraction of the HBM peak."""
    import numpy as np
    from mesh2splat_amd.prepass import PrepassParams
    look_at, proj = viewer_camera()
    conv.convert(R)                                  # a blocking conversion: the context's records are the input
    out = {"camera": "perspective 45 deg 16:9, eye (1.6,1.1,2.3) -> (0.1,0,-0.1), 1920x1080", "records": int(total)}
    conv.set_profiling(True)
    for key, arrival in (("prepass_input_order", False), ("prepass_arrival_order", True)):
        p = PrepassParams(view_mat=look_at((1.6, 1.1, 2.3), (0.1, 0.0, -0.1)), proj_mat=proj, renderer_resolution=(1920, 1080),
                          resolution_target=R, arrival_order=arrival)
        ms = []
        for _ in range(12):
            vis = conv.prepass(p, download=False)
            ms.append(conv.last_prepass_ms)
        m = float(np.median(ms[2:]))
        b = 96 * total + 100 * vis
        out[key] = {"visible": int(vis), "kernel_ms": m, "algorithmic_bytes": int(b), "GBps": b / m / 1e6,
                    "frac_of_hbm_peak": b / (m * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None}
        try:   # HBM bytes per launch from the committed PMC passes (tools/pmc_prepass.sh), same input and camera
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f)
            if t.get("k_" + key + "_detail", {}).get("records") == int(total):
                out[key]["traffic"] = t.get("k_" + key + "_hbm_bytes_per_launch")
        except Exception:  # noqa: BLE001
            pass
    sms = []
    for _ in range(6):
        conv.sort_prepass(download=False)
        sms.append(conv.last_sort_prepass_ms)
    out["sort_prepass_ms"] = float(np.median(sms[1:]))
    try:
        out["frame"] = frame_passes(conv, p, total)
    except Exception as e:  # noqa: BLE001
        out["frame"] = {"error": str(e)}
    conv.set_profiling(False)
    return out


def self_launch(a) -> int:
    """`python bench.py --gpus N` typed by hand: start N ranks ourselves, one per GPU — plain processes with the launcher
    environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT); the ranks meet through mesh2splat_amd/ctl.py."""
    import shutil
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    rdzv = tempfile.mkdtemp(prefix="m2s_rdzv_")      # a private (0700) directory of this launch: where its ranks meet (ctl.py)
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), M2S_RDZV_DIR=rdzv)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p_ in procs:
        rc = p_.wait() or rc
    shutil.rmtree(rdzv, ignore_errors=True)
    return rc


def dry_scale_env(a):
    """--dry-scale: the RCCL stand-in of the tests (shared memory on one device) selected through M2S_RCCL_PATH, every rank on device 0."""
    import tempfile
    stub = os.path.join(ROOT, "tests", "stub_rccl", "_build", "librccl_stub.so")
    if not os.path.exists(stub):
        raise SystemExit("--dry-scale needs tests/stub_rccl/_build/librccl_stub.so: python -c 'import __graft_entry__ as g; g.build()'")
    tmp = tempfile.mkdtemp(prefix="m2s_dry_scale_")
    os.makedirs(os.path.join(tmp, "objs"), exist_ok=True)
    os.environ.update(M2S_RCCL_PATH=stub, M2S_STUB_RCCL_DIR=os.path.join(tmp, "objs"), M2S_STUB_RCCL_LOG=os.path.join(tmp, "rccl_log"))
    os.environ.setdefault("M2S_STUB_RCCL_TIMEOUT", "120")
    if "--one-device" not in sys.argv:
        sys.argv.append("--one-device")


class Rig:
    """One rank's converter on one scene + the step loops (pipelined or blocking), shared by the headline and the extras."""

    def __init__(self, torch, local_rank, scene, R, tri_range=None, cap=-1, out_rows=None, exchange=None, pipeline=None):
        from mesh2splat_amd.converter import Converter
        self.torch, self.R, self.exchange = torch, R, exchange
        self.conv = Converter(local_rank)
        if pipeline:
            self.conv.set_pipeline(pipeline)
        if tri_range is not None:
            self.conv.set_triangle_range(*tri_range)
        self.conv.upload_scene(scene)
        self.conv.set_max_gaussians(cap)
        # (caller-owned record buffers are converted into on a stream of the CALLER's: a non-blocking one of its own — torch's current
        #  stream is the legacy default stream, whose launches synchronise with every blocking stream of the process, the exchange's included)
        self._tstream = torch.cuda.Stream() if out_rows is not None else None
        self.stream = self._tstream.cuda_stream if self._tstream is not None else torch.cuda.current_stream().cuda_stream
        self.out = None
        if out_rows is not None:        # caller-owned record buffer (multi-GPU: the block this rank contributes)
            if out_rows == 0:
                probe = torch.empty((1, 24), dtype=torch.float32, device="cuda")
                out_rows = max(1, self.conv.convert_into(R, probe.data_ptr(), 1, self.stream))
            self.out = torch.empty((out_rows, 24), dtype=torch.float32, device="cuda")
        self.kms = {k: 0.0 for k in ("count", "scan", "offsets", "emit", "fused")}
        self.n_prof = 0
        self.in_flight_counts = 0
        self.last_counts = None

    def _publish(self, total):
        if self.exchange is None:
            return
        self.exchange.publish_count(total)
        self.in_flight_counts += 1
        while self.in_flight_counts > 2:       # keep two exchanges in flight: the host never waits for the newest one
            self.last_counts = self.exchange.collect_counts()
            self.in_flight_counts -= 1

    def drain_counts(self):
        while self.in_flight_counts:
            self.last_counts = self.exchange.collect_counts()
            self.in_flight_counts -= 1

    def submit(self):
        if self.out is not None:
            self.conv.submit(self.R, self.out.data_ptr(), self.out.shape[0], self.stream)
        else:
            self.conv.submit(self.R)

    def step_sync(self):
        if self.out is None:
            total = self.conv.convert(self.R)
        else:
            total = self.conv.convert_into(self.R, self.out.data_ptr(), self.out.shape[0], self.stream)
        self._publish(total)
        return total

    def run(self, k, timed=False, sync_steps=False, prof_every=4):
        """k conversions (pipelined two deep unless sync_steps); returns the last counter; accumulates sampled kernel times"""
        conv, total = self.conv, 0
        if k <= 0:
            return 0
        if sync_steps:
            for i in range(k):
                conv.set_profiling(timed and i % prof_every == 0)
                total = self.step_sync()
                if timed and i % prof_every == 0:
                    self._take_kms()
            conv.set_profiling(False)
            return total
        conv.set_profiling(timed)
        self.submit()
        for i in range(k):
            if i + 1 < k:
                conv.set_profiling(timed and (i + 1) % prof_every == 0)
                self.submit()
            total = conv.wait()
            self._publish(total)
            if timed and i % prof_every == 0:
                self._take_kms()
        conv.set_profiling(False)
        return total

    def _take_kms(self):
        self.n_prof += 1
        for n_, v in self.conv.last_kernel_ms().items():
            self.kms[n_] += v

    def kernel_ms(self):
        return {k: v / max(self.n_prof, 1) for k, v in self.kms.items()}

    def reset_kms(self):
        self.kms = {k: 0.0 for k in self.kms}
        self.n_prof = 0

    def close(self):
        self.conv.close()


def timed_loop(torch, ctl, multi, rig, steps, warmup, sync_steps=False, reset=True):
    """warmup, barrier + sync, `steps` conversions, barrier + sync; returns (seconds (max over ranks), last counter)"""
    def sync():
        torch.cuda.synchronize()
        if multi:
            ctl.barrier()
        torch.cuda.synchronize()
    if warmup:
        rig.run(warmup, sync_steps=sync_steps)
        rig.drain_counts()
    if reset:
        rig.reset_kms()
    sync()
    t0 = time.perf_counter()
    total = rig.run(steps, timed=True, sync_steps=sync_steps)
    rig.drain_counts()
    sync()
    dt = time.perf_counter() - t0
    if multi:
        dt = ctl.max_float(dt)
    return dt, total


def stored_of(rig):
    return rig.conv.num_stored


def whole_conversion_roofline(total_stored, tri, ms, traffic_key=None):
    """96 B per STORED Gaussian + 144 B per triangle over the whole conversion's time.  traffic_key: the entry of
    profiles/pmc_traffic.json that holds the HBM bytes per launch of this workload's dominant kernel (committed PMC passes)."""
    b = 96.0 * total_stored + 144.0 * tri
    gbs = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    out = {"algorithmic_bytes": b, "GBps": gbs, "frac_of_hbm_peak": gbs / HBM_PEAK_GBS}
    if traffic_key:
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(traffic_key, {})
            if t.get("algorithmic_bytes") == b:      # the same workload, to the byte
                k = t["kernel"]
                out["traffic"] = t[k + "_hbm_bytes_per_launch"]
                out["traffic_kernel"] = k
                out["traffic_source"] = "NOT measured in this run: committed rocprofv3 --pmc passes (profiles/pmc_traffic.json)"
        except Exception:  # noqa: BLE001
            pass
    return out


def cold_path(torch, local_rank, scene, R, steady_sync_ms):
    """What a conversion costs when nothing is warm (the reference converts on load and on every move of the density
    slider, guiRendererConcreteMediator.cpp:51-57 — always a NEW (scene, R)):
      first_call_ms  fresh context, resolution hinted, scene uploaded, first m2s_convert.  (Since round 4 the exact count that decides
                     the pipeline, the run table and the record pool are prepared by m2s_upload_scene at the hinted R — its "warm"
                     share is upload_ms["warm"]; first_call_plus_warm_ms adds it back for comparison with round 3);
      new_R_ms       blocking conversions at densities this context has never seen (R, R-8, R-16, ...): no cached decision,
                     no cached band bases, cap changes every time;
      cold_inputs    three independent copies of the scene (> 700 MB of inputs: more than the 256 MiB Infinity Cache) converted
                     round-robin, kernel time by HIP events."""
    import numpy as np
    from mesh2splat_amd.converter import Converter
    out = {}
    conv = Converter(local_rank)
    conv.set_resolution_hint(R)      # the reference knows its resolutionTarget when it loads a model (guiRendererConcreteMediator.cpp:11-29)
    conv.upload_scene(scene)
    out["upload_ms"] = conv.last_upload_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    total = conv.convert(R)
    out["first_call_ms"] = (time.perf_counter() - t0) * 1e3
    # since round 4 the exact count, the pipeline decision, the run table and the record pool are prepared inside m2s_upload_scene
    # (warm_scene: upload_ms["warm"]), at the hinted R: the cost of a first conversion as round 3 measured it is the sum
    out["first_call_plus_warm_ms"] = out["first_call_ms"] + out["upload_ms"]["warm"]
    t0 = time.perf_counter()
    conv.convert(R)
    out["second_call_ms"] = (time.perf_counter() - t0) * 1e3
    new_r = []
    for k in range(1, 9):
        r = R - 8 * k
        if r < 16:
            break
        t0 = time.perf_counter()
        conv.convert(r)
        new_r.append((time.perf_counter() - t0) * 1e3)
    same = []
    for _ in range(12):
        t0 = time.perf_counter()
        conv.convert(R)
        same.append((time.perf_counter() - t0) * 1e3)
    out["new_R_ms"] = {"median": float(np.median(new_r)), "max": float(np.max(new_r)), "densities": [R - 8 * k for k in range(1, len(new_r) + 1)]}
    out["same_R_sync_ms"] = float(np.median(same[2:]))
    out["new_R_over_steady"] = out["new_R_ms"]["median"] / out["same_R_sync_ms"]
    # rotating copies: distinct host arrays -> distinct device copies of geometry and textures
    import copy
    rigs = [conv]
    for _ in range(2):
        c2 = Converter(local_rank)
        c2.upload_scene(copy.deepcopy(scene))
        rigs.append(c2)
    T = conv.num_triangles
    for c_ in rigs:
        c_.convert(R)
        c_.convert(R)
        c_.set_profiling(True)
    ms, wall = [], []
    for i in range(30):
        c_ = rigs[i % 3]
        t0 = time.perf_counter()
        c_.convert(R)
        wall.append((time.perf_counter() - t0) * 1e3)
        k = c_.last_kernel_ms()
        ms.append(sum(k.values()))
    m = float(np.median(ms[3:]))
    inputs_mb = 3 * (144.0 * T + sum(int(t.nbytes) for me in scene.meshes for t in me.textures.values()) * 1.34) / 1e6
    out["cold_inputs"] = {"copies": 3, "input_MB_total": inputs_mb, "kernel_ms": m, "sync_ms": float(np.median(wall[3:])),
                          **whole_conversion_roofline(min(total, conv.num_stored), T, m)}
    for c_ in rigs:
        c_.close()
    return out


def extra_workload(torch, ctl, local_rank, name, steps=24, warmup=3):
    """One more BASELINE config on this GPU: whole-conversion ms, Gaussians/s, roofline fraction by algorithmic bytes."""
    from mesh2splat_amd import synth
    if name == "c5":
        return c5_workload(torch, local_rank)
    n, tex, R = WORKLOADS[name]
    scene = synth.sponza_like() if n == "sponza_like" else synth.sponza_standin(tex) if n == "grid" else synth.colocated_spheres(1, n, tex)
    rig = Rig(torch, local_rank, scene, R)
    dt, total = timed_loop(torch, ctl, False, rig, steps, warmup)
    ms = dt / steps * 1e3
    k = rig.kernel_ms()
    kern = sum(k.values())
    blocking = []
    for _ in range(12):       # the reference's call: one blocking conversion (ConversionPass.cpp:50-59)
        b0 = time.perf_counter()
        rig.step_sync()
        blocking.append((time.perf_counter() - b0) * 1e3)
    blocking.sort()
    res = {"workload": name, "R": R, "triangles": scene.n_triangles, "meshes": scene.n_meshes, "gaussians": int(total),
           "stored": int(stored_of(rig)), "ms_per_step": ms, "value": total / (ms * 1e-3), "value_stored": stored_of(rig) / (ms * 1e-3),
           "blocking_ms": blocking[len(blocking) // 2],
           "pipeline": rig.conv.last_pipeline,
           "kernel_ms": {a: b for a, b in k.items() if b > 0}, "kernels_total_ms": kern,
           "roofline_whole_conversion": whole_conversion_roofline(stored_of(rig), scene.n_triangles, kern, traffic_key=name)}
    res["roofline_blocking"] = whole_conversion_roofline(stored_of(rig), scene.n_triangles, res["blocking_ms"])["frac_of_hbm_peak"]
    try:   # throughput of back-to-back conversions on two lanes (not a kernel time: conversions overlap)
        ov = overlapped_run(torch, rig.conv, R, steps)
        ov["frac_of_hbm_peak"] = whole_conversion_roofline(stored_of(rig), scene.n_triangles, ov["ms_per_step"])["frac_of_hbm_peak"]
        res["overlapped"] = ov
    except Exception as e:  # noqa: BLE001
        res["overlapped"] = {"error": str(e)}
    rig.close()
    return res


def overlapped_run(torch, conv, R, n_ov):
    """m2s_set_async_lanes(2), three conversions in flight: consecutive conversions overlap on two streams (single-pass kernels:
    the tail of one with the head of the next; multi-pass: k_count_scan of one beside k_emit2 of the one before)."""
    conv.set_async_lanes(2)
    for _ in range(4):
        conv.submit(R)
    for _ in range(4):
        conv.wait()
    torch.cuda.synchronize()
    o0 = time.perf_counter()
    conv.submit(R); conv.submit(R)
    ov_total = 0
    for i in range(n_ov):
        if i + 2 < n_ov:
            conv.submit(R)
        ov_total = conv.wait()
    torch.cuda.synchronize()
    ov_ms = (time.perf_counter() - o0) / n_ov * 1e3
    conv.set_async_lanes(1)
    return {"ms_per_step": ov_ms, "value": ov_total / (ov_ms * 1e-3), "unit": "Gaussians/s",
            "what": "m2s_set_async_lanes(2), three conversions in flight: consecutive conversions overlap on two streams"}


def c5_workload(torch, local_rank, steps=8):
    """BASELINE config 5 at FULL size on this one GPU: 4 x cube-sphere n = 1021 = 50 037 168 triangles (7.2 GB of live vertex
    data), 4096^2 maps, R = 2048, cap lifted (the reference's 7 M envelope is exceeded).  Blocking conversions, kernel times by
    HIP events; whole-conversion roofline by algorithmic bytes (96 N + 144 T = 9.53 GB)."""
    import numpy as np
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    t0 = time.perf_counter()
    scene = synth.c5_scene(1021, 4096)
    gen_s = time.perf_counter() - t0
    T = scene.n_triangles
    conv = Converter(local_rank)
    conv.set_max_gaussians(0)          # (before the upload: it prepares the record pool for the conversion below)
    conv.set_resolution_hint(2048)
    conv.set_keep_positions(True)      # BASELINE config 5 sorts what it converts: the conversion also leaves the 16-byte position plane (+16 B written per record, counted in its time)
    t0 = time.perf_counter()
    conv.upload_scene(scene)
    up_s = time.perf_counter() - t0
    conv.set_profiling(True)
    total = conv.convert(2048)
    conv.convert(2048)
    wall, kern = [], []
    for _ in range(steps):
        t0 = time.perf_counter()
        conv.convert(2048)
        wall.append((time.perf_counter() - t0) * 1e3)
        kern.append(sum(conv.last_kernel_ms().values()))
    ms, kms = float(np.median(wall)), float(np.median(kern))
    res = {"workload": "c5 (BASELINE config 5 at full size, one GPU)", "R": 2048, "triangles": T, "meshes": scene.n_meshes, "gaussians": int(total),
           "stored": int(conv.num_stored), "ms_per_step": ms, "value": total / (ms * 1e-3), "value_stored": conv.num_stored / (ms * 1e-3),
           "pipeline": conv.last_pipeline, "kernels_total_ms": kms, "submission": "one blocking call per step",
           "host_generation_s": gen_s, "upload_s": up_s,
           "roofline_whole_conversion": whole_conversion_roofline(conv.num_stored, T, kms, traffic_key="c5")}
    # BASELINE config 5's "final radix sort of the merged splat buffer" (RadixSortPass semantics, m2s_sort_by_depth) on these records.
    # Algorithmic bytes per record: 16 (position read for the key) + 4 x (8 + 8) (four 8-bit passes over key + value, read and
    # write) + 2 x 96 (gather: read + write) = 272.  first: the sort right after the conversion (keys from the records themselves,
    # positions left behind as a plane); repeat: what every later frame pays (keys from the plane).
    try:
        view = np.eye(4, dtype=np.float32)
        view[2, 3] = -6.0
        n = conv.sort_by_depth(view, download=False)          # (allocations, first touch of the buffers)
        conv.convert(2048)                                    # new records at the same address: a plane left by an earlier SORT would be stale
        plane_ready = conv.positions_ready
        conv.sort_by_depth(view, download=False)
        first = {"ms": conv.last_sort_ms, **conv.last_sort_stage_ms, "keys_from": "the position plane the conversion left behind (m2s_set_keep_positions)" if plane_ready else "the records (96-byte stride)"}
        rep, stages = [], []
        for k in range(5):
            view[2, 3] = -6.0 - 0.25 * (k + 1)
            conv.sort_by_depth(view, download=False)
            rep.append(conv.last_sort_ms)
            stages.append(conv.last_sort_stage_ms)
        b = 272.0 * n
        med = float(np.median(rep))
        res["depth_sort"] = {"records": int(n), "first_after_a_conversion_ms": first, "repeat_ms": med,
                             "repeat_stages_ms": {k: float(np.median([s_[k] for s_ in stages])) for k in stages[0]},
                             "algorithmic_bytes": b, "GBps": b / (med * 1e-3) / 1e9, "frac_of_hbm_peak": b / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "sort_core": "rocPRIM radix_sort_pairs (library); key and gather kernels are this repository's"}
    except Exception as e:  # noqa: BLE001
        res["depth_sort"] = {"error": str(e)}
    # ... and one viewer frame over the same 24.3 M records (prepass + depth sort of its output), two calls against the fused pass
    try:
        from mesh2splat_amd.prepass import PrepassParams
        look_at, proj = viewer_camera()
        pv = PrepassParams(view_mat=look_at((3.75, 3.0, 12.0), (3.75, 0.0, 0.0)), proj_mat=proj, renderer_resolution=(1920, 1080), resolution_target=2048)
        res["frame"] = frame_passes(conv, pv, conv.num_stored, reps=4)
    except Exception as e:  # noqa: BLE001
        res["frame"] = {"error": str(e)}
    conv.close()
    return res


def end_to_end(scene_one_mesh, R):
    """SURVEY 8d's second metric, "end-to-end ms/mesh (load + convert + write)": the headless command line (tools/mesh2splat_cli.cpp ==
    the reference's load -> ConversionPass::execute -> exportPly, SceneManager.cpp:195-459, ConversionPass.cpp:9-68, parsers.cpp:431-514)
    on the headline scene written as a .glb with PNG maps; export formats 0 (the reference's 248-byte rows) and 1.  Best of two
    processes per format; every process pays the HIP runtime's start (on a second thread, overlapped with the .glb parse)."""
    import re
    import subprocess
    import tempfile
    from mesh2splat_amd import gltf_io
    cli = os.path.join(ROOT, "mesh2splat_amd", "_build", "mesh2splat")
    if not os.path.exists(cli):
        return {"error": "mesh2splat_amd/_build/mesh2splat is not built"}
    out = {"what": "mesh2splat <in.glb> <out.ply> --density R --format f --timing: wall clock of the stages inside the process, ms",
           "R": int(R)}
    pat = re.compile(r"load ([\d.]+) ms \(HIP runtime \+ context ([\d.]+) ms, on a second thread; waited ([\d.]+) ms for it\) \| upload ([\d.]+) ms "
                     r"\(geometry ([\d.]+), textures ([\d.]+), allocations ([\d.]+)\) \| convert ([\d.]+) ms.*\| export ([\d.]+) ms \| total ([\d.]+) ms")
    with tempfile.TemporaryDirectory() as tmp:
        glb = os.path.join(tmp, "scene.glb")
        gltf_io.write_glb(scene_one_mesh, glb)
        out["glb_bytes"] = os.path.getsize(glb)
        for fmt in (0, 1):
            ply = os.path.join(tmp, "out%d.ply" % fmt)
            best = None
            for _ in range(2):
                w0 = time.perf_counter()
                r = subprocess.run([cli, glb, ply, "--density", str(int(R)), "--format", str(fmt), "--timing"], capture_output=True, text=True, timeout=600)
                wall = (time.perf_counter() - w0) * 1e3
                m = pat.search(r.stdout) if r.returncode == 0 else None
                if m is None:
                    return {"error": (r.stderr or r.stdout)[-400:]}
                t = dict(zip(("load_ms", "hip_init_ms_overlapped", "waited_for_init_ms", "upload_ms", "upload_geometry_ms", "upload_textures_ms",
                              "upload_alloc_ms", "convert_first_call_ms", "export_ms", "total_ms"), map(float, m.groups())))
                t["wall_ms_whole_process"] = wall
                if best is None or t["total_ms"] < best["total_ms"]:
                    best = t
            best["ply_bytes"] = os.path.getsize(ply)
            out["format%d" % fmt] = best
    return out


class stdout_to_stderr:
    """gloo and librccl announce themselves on C stdout; the driver reads ONE JSON line from this process's stdout.  While the
    process group / the communicator come up, file descriptor 1 points at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        if a.dry_scale:
            dry_scale_env(a)
        raise SystemExit(self_launch(a))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        a.gpus = world
    if a.one_device:
        local_rank = 0

    import torch  # first, so that the HIP runtime torch ships is the one the process uses
    import numpy as np
    from mesh2splat_amd import synth
    from mesh2splat_amd import dist as m2d
    from mesh2splat_amd.ctl import Ctl

    torch.cuda.set_device(local_rank)
    multi = world > 1 or a.force_dist
    exchange = None
    phases = {}          # multi-GPU: how long each bring-up step took on this rank, and every error met on the way (-> the JSON line)
    dist_errors = []
    import threading
    from mesh2splat_amd.ctl import RendezvousTimeout

    def leave(phase, text, code=3):
        """A multi-rank phase did not complete: say so in the shape of the record the run was started for, and leave non-zero.
        Rank 0 prints it on stdout (the line a driver reads) and stderr; the other ranks on stderr."""
        rec = {"error": f"{phase}: {text}", "n_gpus": world, "value": None, "exchange_transport": getattr(exchange, "transport", None),
               "scale_record": {"rccl_ranks": getattr(exchange, "world", None), "error": f"{phase}: {text}", "phase": phase, "rank": rank,
                                "errors": dist_errors, "dry_scale": bool(a.dry_scale)},
               "multi_gpu_bringup": phases}
        line = json.dumps(rec)
        print(line, file=sys.stderr, flush=True)
        if rank == 0:
            print(line, flush=True)
        os._exit(code)

    class TimeBox:
        """every N > 1 phase runs under one: a hang in a collective becomes a record with the phase's name after `seconds`"""
        def __init__(self, phase, seconds):
            self.t = threading.Timer(seconds, lambda: leave(phase, f"no progress after {seconds:.0f} s (time box)", 4)) if world > 1 or a.force_dist else None
            if self.t:
                self.t.daemon = True
                self.t.start()

        def done(self):
            if self.t:
                self.t.cancel()

    try:
        ctl = Ctl(rank, world, timeout=a.bringup_timeout)
    except RendezvousTimeout as e:
        leave("rendezvous", str(e))
    if multi:
        box = TimeBox("bring-up (rendezvous + m2s_dist_create + first exchange)", a.bringup_timeout + 10.0)
        if ctl.rdzv is not None:
            ctl.rdzv.set_deadline(a.bringup_timeout)
        os.environ.setdefault("NCCL_DEBUG", "WARN")          # RCCL's own account of a failure goes to stderr
        # ONE user of RCCL per process — the C-ABI communicator (m2s_dist_*), which carries the data path — and ONE rendezvous: the
        # ranks meet in a directory (mesh2splat_amd/ctl.py) for the 128-byte id and the agreement that every rank has a communicator;
        # from then on the barriers and reductions of this run are 8-byte all-gathers of that communicator.
        t_pg = time.perf_counter()
        try:
            ctl.barrier()
        except RendezvousTimeout as e:
            leave("rendezvous", str(e))
        phases["control_plane"] = "directory rendezvous (mesh2splat_amd/ctl.py) until m2s_dist_create, then the communicator's own 8-byte all-gather"
        phases["control_plane_init_s"] = time.perf_counter() - t_pg

        # rank 0's RCCL id reaches the other ranks together with a flag: if rank 0 cannot get one (no librccl), EVERY rank learns it
        # here, before anybody enters a collective of the communicator
        payload = None
        if rank == 0:
            try:
                if os.environ.get("M2S_BENCH_FAIL_COMM"):      # (test hook: the exit path below)
                    raise RuntimeError("forced by M2S_BENCH_FAIL_COMM")
                payload = b"\x01" + m2d.RcclExchange.unique_id()
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] C-ABI RCCL exchange unavailable: {e!r}", file=sys.stderr, flush=True)
                dist_errors.append(f"rank 0: m2s_dist_unique_id: {e!r}")
                payload = b"\x00"
        try:
            payload = ctl.broadcast_bytes("rccl_id", payload)
        except RendezvousTimeout as e:
            leave("rendezvous (RCCL id)", str(e))
        exchange, ok = None, 0
        if payload[:1] == b"\x01":
            ident = bytes(payload[1:129])

            def bootstrap(_):
                return ident
            bootstrap.provides_id = True
            t_comm = time.perf_counter()
            try:
                with stdout_to_stderr():
                    exchange = m2d.RcclExchange(local_rank, rank, world, bootstrap)
                ok = 1
                phases["rccl_comm_init_s"] = time.perf_counter() - t_comm
            except Exception as e:  # noqa: BLE001
                print(f"[rank {rank}] C-ABI RCCL exchange unavailable: {e!r}", file=sys.stderr, flush=True)
                dist_errors.append(f"rank {rank}: m2s_dist_create: {e!r}")
                exchange, ok = None, 0
        try:
            reports = ctl.gather_obj("comm_ok", {"ok": ok, "errors": dist_errors})
        except RendezvousTimeout as e:
            leave("rendezvous (did every rank get a communicator?)", str(e))
        if min(r_["ok"] for r_ in reports) == 0:
            # NO fallback: the exchange under test is the product's (m2s_dist_* behind the C ABI: RCCL).  A run whose ranks cannot
            # create that communicator must not report a number measured on something else: every rank leaves, rank 0 says why.
            if exchange is not None:
                exchange.close()
            dist_errors[:] = [e for r_ in reports for e in r_["errors"]]
            exchange = None
            try:
                ctl.close()
            except Exception:  # noqa: BLE001
                pass
            leave("m2s_dist_create", "the C-ABI communicator could not be created on every rank; no measurement was taken")
        ctl.use(exchange)
        if exchange is not None:           # the first exchange of all: 8 bytes per rank, timed on its own
            t_x = time.perf_counter()
            try:
                exchange.all_gather_counts(rank)
                phases["first_count_exchange_ms"] = (time.perf_counter() - t_x) * 1e3
            except Exception as e:  # noqa: BLE001
                dist_errors.append(f"rank {rank}: first counter exchange: {e!r}")
                print(f"[rank {rank}] first counter exchange failed: {e!r}", file=sys.stderr, flush=True)
        # librccl announces itself through C stdio ("Librccl path : ..."); push that out now so that the JSON line
        # stays the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        if ctl.rdzv is not None:
            ctl.rdzv.set_deadline(None)
        box.done()

    n, tex, R = WORKLOADS[a.workload]
    if n in ("grid", "sponza_like"):
        if world > 1:
            raise SystemExit("workloads c4 / hetero as the headline are single-GPU (c4 is part of the strong-scaling section at N > 1)")
        scene = synth.sponza_standin(tex) if n == "grid" else synth.sponza_like()
        tri_per_mesh = scene.n_triangles
    else:
        scene = synth.colocated_spheres(world, n, tex)
        tri_per_mesh = scene.meshes[0].n_triangles

    # ---- headline: WEAK scaling, one I-3 mesh per rank -------------------------------------------------------------
    # N == 1: the reference's own cap formula.  N > 1: the merged scene exceeds the reference's 7 M envelope,
    # so the cap is lifted and each rank writes into its own buffer; the counters are exchanged every step (RCCL, C ABI).
    rig = Rig(torch, local_rank, scene, R, tri_range=(rank * tri_per_mesh, tri_per_mesh), cap=0 if multi else -1,
              out_rows=0 if multi else None, exchange=exchange, pipeline=a.pipeline)
    T_local = rig.conv.num_triangles
    box = TimeBox("weak-scaling headline (timed loop + counter exchange)", a.headline_timeout)
    # The timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize, max over ranks — repeated `reps` times
    # (warm-up only in front of the first); the line's ms_per_step / value are those of the MEDIAN repetition.
    reps = a.reps if a.reps > 0 else (5 if a.steps < 100 else 1)
    rep_dt = []
    for i in range(reps):
        dt_i, total = timed_loop(torch, ctl, multi, rig, a.steps, a.warmup if i == 0 else 0, sync_steps=a.sync_steps, reset=(i == 0))
        rep_dt.append(dt_i)
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    per_rank_total = ctl.gather_u64(total) if multi else [int(total)]
    box.done()
    n_all = int(sum(per_rank_total))
    ms_per_step = dt / a.steps * 1e3
    value = n_all / (dt / a.steps)
    rep_ms = [d / a.steps * 1e3 for d in rep_dt]
    kms = rig.kernel_ms()
    n_prof = rig.n_prof
    last_pipeline = rig.conv.last_pipeline
    stored_local = stored_of(rig)

    # for the record: the same loop with one blocking call per step (outside the timed region)
    per_step = []
    for _ in range(min(a.steps, 20)):
        p0 = time.perf_counter()
        rig.step_sync()
        per_step.append((time.perf_counter() - p0) * 1e3)
    rig.drain_counts()
    torch.cuda.synchronize()
    sync_ms = float(np.mean(per_step))
    per_step.sort()
    sync_stats = {"median": per_step[len(per_step) // 2], "p10": per_step[len(per_step) // 10], "p90": per_step[(len(per_step) * 9) // 10]}
    # a dedicated kernel-time sample: 64 blocking launches, HIP events on every one (not part of `value`)
    rig.reset_kms()
    rig.run(64, timed=True, sync_steps=True, prof_every=1)
    rig.drain_counts()
    dedicated = rig.kernel_ms()

    overlapped = None
    if not a.no_overlap_extra and not multi:
        overlapped = overlapped_run(torch, rig.conv, R, max(min(a.steps, 60), 6))
    viewer = None
    if not a.no_viewer_extra and not multi:
        try:
            viewer = viewer_extra(rig.conv, R, total)
        except Exception as e:  # noqa: BLE001 - an extra must not take the headline down
            viewer = {"error": str(e)}

    copy_gbs = None
    if rank == 0:
        nbytes = 1 << 30
        src = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        dstb = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        for _ in range(2):
            dstb.copy_(src)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(10):
            dstb.copy_(src)
        torch.cuda.synchronize()
        copy_gbs = 2.0 * nbytes * 10 / (time.perf_counter() - c0) / 1e9
        del src, dstb

    res = None
    if rank == 0:
        dom = "fused" if kms["fused"] > 0 else "emit"     # the dominant kernel of the pipeline that ran
        kname = {"team": "k_fused2", "lean": "k_fused3", "sparse": "k_sparse"}.get(last_pipeline, "k_emit2")
        emit_ms = kms[dom]
        # algorithmic bytes of one launch of the dominant kernel: 96 B per Gaussian STORED + 144 B per triangle read
        # (B_alg = 96 N + 144 T); textures, offsets and the entry list are not credited.
        b_alg = 96.0 * stored_local + 144.0 * T_local
        achieved = b_alg / (emit_ms * 1e-3) / 1e9 if emit_ms > 0 else 0.0
        res = {
            "metric": "Gaussians/sec emitted (mesh->3DGS conversion pass, density 1024^2)" if R == 1024 else
                      f"Gaussians/sec emitted (mesh->3DGS conversion pass, density {R}^2)",
            "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "ms_per_mesh": ms_per_step,
            "ms_per_step_reps": {"reps": reps, "all": rep_ms, "min": min(rep_ms), "max": max(rep_ms), "median": ms_per_step,
                                 "what": f"{reps} repetition(s) of the timed region, each exactly {a.steps} steps; ms_per_step and value are the median repetition's"},
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "exchange_transport": exchange.transport if multi else "none (single GPU: no exchange)",
            "config": {"workload": (f"synth.sponza_like: 64 meshes, {tri_per_mesh} triangles of 0.1 ... 500 000 px, maps of 256^2 ... 2048^2, R={R}") if n == "sponza_like" else
                                   (f"I-4 64 meshes x cube-sphere n=18 ({tri_per_mesh} triangles), 64 materials with 3 procedural "
                                    f"{tex}^2 RGBA8 maps each, R={R}") if n == "grid" else
                                   (f"I-3 cube-sphere n={n} ({tri_per_mesh} triangles/mesh) x {world} mesh(es), "
                                    f"3 procedural {tex}^2 RGBA8 maps, R={R}"),
                       "gaussians_per_step": n_all, "triangles_per_gpu": T_local, "parallelism": f"tri-range x{world}",
                       "pipeline": last_pipeline,
                       "rccl_ranks": (exchange.world if multi else 1),
                       "exchange": ("per step: 8-byte counter all-gather, two in flight; transport: " + exchange.transport) if multi else "none (single GPU)",
                       "cap": "unlimited (merged scene exceeds the 7M envelope)" if multi else "reference formula",
                       "submission": "one blocking call per step" if a.sync_steps else
                                     "pipelined 2 deep (m2s_convert_submit/wait): every conversion completes and its counter is read back in the timed region"},
            "sync_ms_per_step": sync_ms, "sync_ms_stats": sync_stats, "overlapped": overlapped, "viewer_passes": viewer,
            "kernel_ms": kms,
            "kernel_timing": f"HIP events on the launch stream around every 4th launch of the timed region(s) ({n_prof} launches)",
            "kernel_ms_dedicated": {"what": "64 blocking launches after the timed region, HIP events on every one", **dedicated},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None, "kernel": kname,
                         "algorithmic_bytes": b_alg, "measured_copy_peak": copy_gbs,
                         "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "frac_dedicated_sample": (b_alg / (dedicated[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if dedicated[dom] > 0 else None,
                         "write_only_frac": (96.0 * stored_local / (emit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if emit_ms > 0 else 0.0},
        }
        tr = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tr):
            try:
                with open(tr) as f:
                    t = json.load(f)
                if t.get("workload") == a.workload and t.get(kname + "_hbm_bytes_per_launch"):
                    res["roofline"]["traffic"] = t.get(kname + "_hbm_bytes_per_launch")
                    # the counters were taken on ONE build of the library: say which, and whether it is the one running now
                    from mesh2splat_amd import _lib as _m2slib
                    import hashlib
                    with open(_m2slib.LIB_PATH, "rb") as fb:
                        sha_now = hashlib.sha256(fb.read()).hexdigest()[:16]
                    res["roofline"]["traffic_binary_sha"] = {"when_measured": t.get("binary_sha", {}).get(kname), "now": sha_now,
                                                             "same_binary": t.get("binary_sha", {}).get(kname) == sha_now}
                    res["roofline"]["traffic_source"] = ("NOT measured in this run: (2*FETCH_SIZE + WRITE_SIZE)*1024 per launch from the committed "
                                                         "rocprofv3 --pmc passes of the same command (profiles/pmc_traffic.json)")
            except Exception:
                pass
    xch_ms = None
    if multi:       # one 8-byte all-gather (publish + collect, blocking): what a rank pays to learn its offset in the merged buffer
        ctl.barrier()
        x0 = time.perf_counter()
        for _ in range(20):
            exchange.all_gather_counts(total)
        xch_ms = ctl.max_float((time.perf_counter() - x0) / 20) * 1e3
    if rank == 0 and multi:
        # the record the driver's SCALE run is read for, in one flat place
        res["scale_record"] = {"rccl_ranks": exchange.world, "per_rank_gaussians": [int(x) for x in per_rank_total], "value": value,
                               "ms_per_step": ms_per_step, "exchange_ms": xch_ms,
                               "bringup_ms": (phases.get("control_plane_init_s", 0.0) + phases.get("rccl_comm_init_s", 0.0)) * 1e3,
                               "transport": exchange.transport, "dry_scale": bool(a.dry_scale),
                               # what the record exchange (`gather`, below) cannot beat on xGMI (MI355X_MICROARCH.md: 7 links x ~153 GB/s per
                               # GPU, point to point): every rank receives the other ranks' blocks, each over the one link to that peer
                               "gather_link_floor_ms": {
                                   "all_7_links": (world - 1) / max(world, 1) * 96.0 * n_all / (7 * 153e9) * 1e3,
                                   "links_in_use": (96.0 * max(per_rank_total) / 153e9 * 1e3) if world > 1 else 0.0,
                                   "what": "all_7_links: (N-1)/N x 96 B x N_total / (7 x 153 GB/s); links_in_use: the largest block over ONE link "
                                           "(a rank has N-1 peers, one link each: with N < 8 only N-1 links carry data)"},
                               "what": "exchange_ms: one blocking 8-byte counter all-gather; bringup_ms: rendezvous + m2s_dist_create (ncclCommInitRank)"}
        res["dry_scale"] = bool(a.dry_scale)
        res["multi_gpu_bringup"] = {**phases, "exchange": getattr(exchange, "transport", None), "errors": dist_errors,
                                    "what": "rank 0's timings of the bring-up steps; errors: every m2s_dist_* failure met so far (text of m2s_dist_last_error)"}
    # The multi-GPU extras below contain collectives that have never run on more than one GPU by the builder.  Should one of
    # them hang, every rank leaves after --extras-timeout seconds and rank 0 still prints the (complete) headline line.
    watchdog = None
    if multi:
        import threading

        def bail():
            if rank == 0:
                res["multi_gpu_extras"] = f"TIMED OUT after {a.extras_timeout:.0f} s: what had finished by then is in this line"
                print(json.dumps(res), flush=True)
            os._exit(0)
        watchdog = threading.Timer(a.extras_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    # ---- multi-GPU extras: the record exchange of the weak-scaling run, then STRONG scaling of ONE scene --------------
    gather = None
    strong = None
    if multi:
        counts, offs = exchange.all_gather_counts(total)
        reps = 5
        if not a.no_gather:
            merged = torch.empty((max(offs[-1], 1), 24), dtype=torch.float32, device="cuda")
            for _ in range(2):
                exchange.gather_records_t(rig.out, counts, merged, -1, rig.stream)
            torch.cuda.synchronize(); ctl.barrier()
            g0 = time.perf_counter()
            for _ in range(reps):
                rig.step_sync()
                exchange.gather_records_t(rig.out, counts, merged, -1, rig.stream)
            rig.drain_counts()
            torch.cuda.synchronize(); ctl.barrier()
            gdt = ctl.max_float((time.perf_counter() - g0) / reps)
            gather = {"ms_per_step": gdt * 1e3, "value": n_all / gdt, "unit": "Gaussians/s",
                      "bytes_received_per_rank": 96 * (offs[-1] - counts[rank]),
                      "what": "convert + exact-size all-pairs exchange of every rank's block to every rank (m2s_dist_gather_records: one "
                              "RCCL group of ncclSend/ncclRecv at the final offsets of the merged buffer)"}
            del merged
            if rank == 0:
                res["gather"] = gather
        rig.close()
        strong = {}
        if rank == 0:
            res["strong_scaling"] = strong
        strong_scenes = ([] if a.no_strong_scaling else ["c3"]) + ([] if (a.no_extra_workloads or a.no_strong_scaling) else ["c4"]) + (["c5p"] if ((world == 8 or os.environ.get("M2S_BENCH_FORCE_C5P")) and not a.no_extra_workloads) else [])
        for sname in strong_scenes:
            # every rank takes the same path through this block (collectives inside): an exception is recorded, not raised
            try:
                if sname == "c5p":
                    # BASELINE config 5 at 1/8 of its triangle count (the full 50 M-triangle scene is 7.2 GB of vertices per rank to
                    # generate on the host): 4 meshes x cube-sphere n=361 = 6.25 M triangles, 4096^2 maps, R = 2048, cap lifted,
                    # + the depth sort of the MERGED buffer (m2s_set_records + m2s_sort_by_depth on every rank)
                    one, sR = synth.sphere_row(4, int(os.environ.get("M2S_BENCH_C5P_N", "361")), int(os.environ.get("M2S_BENCH_C5P_TEX", "4096"))), 2048   # (env: test hooks)
                else:
                    sn, stex, sR = WORKLOADS[sname]
                    one = synth.sponza_standin(stex) if sn == "grid" else synth.colocated_spheres(1, sn, stex)
                plan = m2d.shard_ranges_native(one, sR, world)
                srig = Rig(torch, local_rank, one, sR, tri_range=plan[rank], cap=0, out_rows=0, exchange=exchange)
                sdt, stotal = timed_loop(torch, ctl, True, srig, a.steps, a.warmup)
                scounts, soffs = exchange.all_gather_counts(stotal)
                entry = {"scene": sname, "R": sR, "triangles": one.n_triangles, "gaussians": soffs[-1], "per_rank_gaussians": scounts,
                         "per_rank_triangles": [c_ for _, c_ in plan],
                         "no_gather": {"ms_per_step": sdt / a.steps * 1e3, "value": soffs[-1] / (sdt / a.steps),
                                       "what": "shards by estimated fragments (m2s_dist_shard_ranges), convert + counter exchange; every rank "
                                               "keeps its block and knows its offset (what per-rank .ply slice writers need)"}}
                if not a.no_gather:
                    merged = torch.empty((max(soffs[-1], 1), 24), dtype=torch.float32, device="cuda")
                    exchange.gather_records_t(srig.out, scounts, merged, -1, srig.stream)
                    torch.cuda.synchronize(); ctl.barrier()
                    g0 = time.perf_counter()
                    for _ in range(reps):
                        srig.step_sync()
                        exchange.gather_records_t(srig.out, scounts, merged, -1, srig.stream)
                    srig.drain_counts()
                    torch.cuda.synchronize(); ctl.barrier()
                    gdt = ctl.max_float((time.perf_counter() - g0) / reps)
                    entry["gather"] = {"ms_per_step": gdt * 1e3, "value": soffs[-1] / gdt,
                                       "what": "convert + all-pairs record exchange into the merged buffer on every rank"}
                    # the merged buffer must be the single-GPU output: checksum of checksums across ranks
                    chk = int(merged.view(torch.int32).to(torch.int64).sum().item())
                    allchk = ctl.gather_u64(chk)
                    entry["gather"]["merged_identical_on_all_ranks"] = bool(all(x == (chk & 0xFFFFFFFFFFFFFFFF) for x in allchk))
                    if sname == "c5p":
                        from mesh2splat_amd.converter import Converter
                        sink = Converter(local_rank)
                        sink.set_records(merged.data_ptr(), soffs[-1], sR)
                        view = np.eye(4, dtype=np.float32)
                        view[2, 3] = -6.0                          # camera on +z looking at the row of spheres
                        sink.set_profiling(True)
                        sms = []
                        for _ in range(3):
                            nsort = sink.sort_by_depth(view, download=False)
                            sms.append(sink.last_sort_ms)
                        entry["merged_depth_sort"] = {"records": int(nsort), "ms": float(np.median(sms)),
                                                      "what": "m2s_set_records(merged) + m2s_sort_by_depth: key build + radix sort + 96-byte gather, on every rank"}
                        sink.close()
                        # the same sort WITHOUT merging on one GPU: sample sort across the ranks behind the C ABI (one exact-size record
                        # exchange; every rank ends with its slice of the global order).  Checked against the merged buffer: same
                        # multiset (checksum), sorted inside every slice, slices ordered across ranks.
                        if True:
                            try:
                                srig.drain_counts()
                                torch.cuda.synchronize(); ctl.barrier()
                                d0 = time.perf_counter()
                                dn, doff = exchange.sort_by_depth(srig.conv, view)
                                ddt = ctl.max_float(time.perf_counter() - d0)
                                sl = torch.from_numpy(srig.conv.download_sorted()).cuda() if dn else torch.zeros((0, 24), dtype=torch.float32, device="cuda")
                                z = (sl[:, 2] + torch.tensor(-6.0, device="cuda")).contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF   # key = bits of view-space z
                                edge = [int(z[0].item()) if dn else -1, int(z[-1].item()) if dn else -1,
                                        int(sl.view(torch.int32).to(torch.int64).sum().item()), int(dn),
                                        int(bool((z[1:] >= z[:-1]).all().item())) if dn > 1 else 1]
                                E = ctl.gather_obj("sort_edges_" + sname, edge)
                                nonempty = [e for e in E if e[3] > 0]
                                entry["distributed_depth_sort"] = {
                                    "ms": ddt * 1e3, "per_rank_records": [e[3] for e in E],
                                    "same_multiset_as_merged": sum(e[2] for e in E) == chk,
                                    "sorted_inside_slices": all(e[4] == 1 for e in E),
                                    "slices_ordered": all(nonempty[i][1] <= nonempty[i + 1][0] for i in range(len(nonempty) - 1)),
                                    "what": "m2s_dist_sort_by_depth: local sort, 256 samples per rank, splitters, one all-pairs exchange of exact sizes, "
                                            "local sort of what arrived; wall time of one call, max over ranks"}
                            except Exception as e:  # noqa: BLE001
                                entry["distributed_depth_sort"] = {"error": repr(e)}
                    del merged
                strong[sname] = entry
                srig.close()
            except Exception as e:  # noqa: BLE001
                strong[sname] = {"error": repr(e)}

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        if not multi:
            one = scene if n in ("grid", "sponza_like") else synth.colocated_spheres(1, n, tex)
            if not a.no_cold:
                try:
                    res["cold_path"] = cold_path(torch, local_rank, one, R, sync_ms)
                except Exception as e:  # noqa: BLE001
                    res["cold_path"] = {"error": str(e)}
            # The fractions a reader should not have to dig for: every one is algorithmic bytes of ONE conversion over a
            # WALL-CLOCK time of that conversion as the caller sees it / 8 TB/s —
            #   frac            the dominant kernel alone, HIP events inside the timed region            (the kernel's roofline)
            #   frac_step       the driver-timed step: ms_per_step of this line                          (steady state, same scene and R)
            #   frac_first_call a NEW scene: first m2s_convert after m2s_upload_scene, blocking          (what the reference does on load)
            #   frac_new_R      the same scene at a density it has never been converted at, blocking     (a move of the density slider)
            #   frac_cold_inputs  inputs not resident in the 256 MiB Infinity Cache (kernel time, three rotating scene copies)
            rf, cp = res["roofline"], res.get("cold_path") or {}
            b = rf["algorithmic_bytes"]
            rf["frac_step"] = b / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS
            rf["frac_blocking"] = b / (sync_ms * 1e-3) / 1e9 / HBM_PEAK_GBS       # one blocking m2s_convert per step: the reference's semantics
            rf["frac_clock"] = ("frac: HIP events on the launch stream around every 4th launch of the timed region (two conversions in flight); "
                                "frac_dedicated_sample: HIP events around 64 blocking launches (agrees with rocprofv3's average); "
                                "frac_step: the driver-timed ms_per_step; frac_blocking: wall clock of one blocking call.  The claim "
                                "against the north star's 40 % is made on frac_step and frac_blocking (whole steps, host included).")
            if "first_call_ms" in cp:
                rf["frac_first_call"] = b / (cp["first_call_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                rf["frac_first_call_including_warm"] = b / (cp["first_call_plus_warm_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                # (the other densities move (R'/R)^2 of the records: scale the bytes of the record part accordingly)
                rs = cp["new_R_ms"]["densities"]
                scale = float(np.mean([(r / R) ** 2 for r in rs])) if rs else 1.0
                rf["frac_new_R"] = (96.0 * stored_local * scale + 144.0 * T_local) / (cp["new_R_ms"]["median"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                rf["frac_cold_inputs"] = cp["cold_inputs"].get("frac_of_hbm_peak")
            if rf.get("traffic") is not None:
                res["traffic_note"] = "roofline.traffic is a COMMITTED constant of the same command (see roofline.traffic_source), not a measurement of this run"
            if not a.no_extra_workloads and a.workload == "c3":
                res["extra_workloads"] = {}
                for w in ("c2", "band", "mid", "c4", "hetero") + (() if a.no_c5 else ("c5",)):
                    try:
                        res["extra_workloads"][w] = extra_workload(torch, ctl, local_rank, w)
                    except Exception as e:  # noqa: BLE001
                        res["extra_workloads"][w] = {"error": str(e)}
            # every fraction of this line in one flat place: algorithmic bytes of ONE conversion / its time / 8 TB/s;
            # for the other workloads the time is the sum of the conversion's kernels by HIP events (`*_blocking`: one blocking call)
            wl = {}
            for w, e in (res.get("extra_workloads") or {}).items():
                if isinstance(e, dict) and "roofline_whole_conversion" in e:
                    wl[w] = e["roofline_whole_conversion"]["frac_of_hbm_peak"]
                    if "roofline_blocking" in e:
                        wl[w + "_blocking"] = e["roofline_blocking"]
                    if isinstance(e.get("depth_sort"), dict) and "frac_of_hbm_peak" in e["depth_sort"]:
                        wl[w + "_depth_sort"] = e["depth_sort"]["frac_of_hbm_peak"]
            for key, name_ in (("frac_first_call", "first_call"), ("frac_new_R", "new_R"), ("frac_cold_inputs", "cold")):
                if rf.get(key) is not None:
                    wl[name_] = rf[key]
            rf["workloads"] = wl
            if not a.no_end_to_end:
                try:
                    res["end_to_end_ms"] = end_to_end(one, R)
                except Exception as e:  # noqa: BLE001
                    res["end_to_end_ms"] = {"error": str(e)}
            if not a.no_cpu_baseline:
                res["cpu_baseline"] = cpu_baseline(one, R, a.cpu_seconds, gpu_total=total)
        print(json.dumps(res), flush=True)

    if multi:
        ctl.barrier()
        ctl.use(None)
        exchange.close()
        ctl.close()


if __name__ == "__main__":
    main()
