#!/usr/bin/env python
"""bench.py — Gaussians/sec of the mesh -> 3DGS conversion pass on N MI355X GPUs.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one execution of the conversion pass (== ConversionPass::execute: count, scan,
offsets, emit, counter read-back) on geometry and textures already resident in HBM.

Workload (BASELINE.json configs[2], the one the metric is quoted on): I-3 = cube-sphere n=289
(1 002 252 triangles), three procedural 2048^2 RGBA8 maps, density R = 1024.
N > 1 is WEAK scaling: the scene holds N such meshes (co-located, so every mesh has the same
cumulative bbox and the same fragment count), sharded by triangle range one mesh per rank; the
only collective in the timed step is the all-gather of the per-rank counters that gives every
rank its offset in the virtual concatenated splat buffer.  The full record all-gather over xGMI
(north-star) is measured separately after the timed region and reported under "gather".

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (cube-sphere n, texture size, R)
    "c3": (289, 2048, 1024),   # BASELINE configs[2] — default
    "c2": (76, 2048, 512),     # SciFiHelmet stand-in (I-2)
    "small": (24, 256, 256),   # CI-sized
    "c4": ("grid", 1024, 1024),  # Sponza stand-in (I-4): 64 meshes x cube-sphere n=18, 64 materials (single GPU only)
    "mid": (94, 2048, 1024),   # one mesh, 106 032 triangles, ~26 fragments per triangle (Sponza-like triangle sizes)
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="skip the post-run record all-gather measurement")
    ap.add_argument("--gather-direct", action="store_true", help="also time the exact-size all-pairs record exchange (dist.all_gather_records)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="budget for the CPU baseline leg")
    ap.add_argument("--no-viewer-extra", action="store_true", help="skip the prepass / depth-sort measurement after the timed region")
    ap.add_argument("--overlap-extra", action="store_true",
                    help="after the timed region, also measure two-lane overlapped submission (reported as 'overlapped'; off by "
                         "default so that a rocprofv3 trace of the default command holds only isolated launches)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking m2s_convert per step instead of the two-deep pipeline")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-path measurements (first call, new R, rotating scene copies)")
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
    """The reference ITSELF on the host: oracle/_ref/ref_pipeline_check = the reference's SceneManager::loadModel,
    ConversionPass::execute and converter{VS,GS,FS}.glsl (C++ through glm), compiled from /root/reference by
    oracle/Makefile, on a minimal software GL (the GL driver is the one thing that cannot run here).  Timed: execute()
    on the whole workload, one thread (the reference's host code is single-threaded).  None if the binary is absent."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_pipeline_check")
    if not (os.path.isfile(exe) and os.access(exe, os.X_OK)):
        return None
    from mesh2splat_amd import gltf_io
    try:
        with tempfile.TemporaryDirectory() as d:
            glb = os.path.join(d, "workload.glb")
            gltf_io.write_glb(scene_one_mesh, glb, indexed=False)
            r = subprocess.run([exe, glb, str(int(R)), "-"], capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return None
        info = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        return None
    sec = info["execute_ms"] * 1e-3
    return {"value": info["counter"] / sec, "unit": "Gaussians/s", "cores": 1, "kind": "reference",
            "sample": f"full workload ({info['counter']} Gaussians): the reference's ConversionPass::execute + its three shaders "
                      "(as C++ through glm) on oracle/ref_pipeline_check's software GL, single thread, one run",
            "ms_per_mesh": info["execute_ms"], "counter": info["counter"],
            "counter_equals_gpu": bool(info["counter"] == expect_total)}


def viewer_extra(conv, R, total):
    """GaussiansPrepass + RadixSortPass on the records of the last conversion: kernel ms (HIP events), algorithmic bytes
    96*n read + 100*visible written, fraction of the HBM peak."""
    import math
    import numpy as np
    from mesh2splat_amd.prepass import PrepassParams

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
    conv.set_profiling(False)
    return out


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world

    import torch  # first, so that the HIP runtime torch ships is the one the process uses
    import torch.distributed as dist
    import numpy as np
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter

    torch.cuda.set_device(local_rank)
    multi = world > 1 or a.force_dist
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        dist.barrier()
        # librccl announces itself through C stdio ("Librccl path : ..."); push that out now so that the JSON line
        # stays the LAST line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()

    n, tex, R = WORKLOADS[a.workload]
    if n == "grid":
        if world > 1:
            raise SystemExit("workload c4 is a single-GPU diagnostic")
        scene = synth.sphere_grid(4, n=18, tex_size=tex)
        tri_per_mesh = scene.n_triangles
    else:
        scene = synth.colocated_spheres(world, n, tex)
        tri_per_mesh = scene.meshes[0].n_triangles

    conv = Converter(local_rank)
    conv.set_triangle_range(rank * tri_per_mesh, tri_per_mesh)
    conv.upload_scene(scene)
    # N == 1: the reference's own cap formula.  N > 1: the merged scene exceeds the reference's
    # 7 M envelope, so the cap is lifted and each rank writes into its own buffer.
    conv.set_max_gaussians(0 if multi else -1)
    T_local = conv.num_triangles

    stream = torch.cuda.current_stream().cuda_stream
    out = None
    RING = 4              # counters of up to RING steps in flight: each all-gather owns its send and receive slot
    counts_ring = torch.zeros((RING, world), dtype=torch.int64, device="cuda")
    mine_ring = torch.zeros((RING, 1), dtype=torch.int64, device="cuda")
    counts = counts_ring[0]
    n_published = [0]

    # Steps are pipelined two deep (m2s_convert_submit / m2s_convert_wait): while the GPU runs conversion k the host
    # has already enqueued k+1 and reads k's counter afterwards.  Every conversion runs to completion inside the
    # timed region and every counter is read back; what disappears is the launch + completion round trip between
    # consecutive kernels.  --sync-steps restores one blocking call per step.
    side = torch.cuda.Stream() if multi else None
    pending = []          # outstanding counter all-gathers (multi-GPU)

    def publish(total):
        """offsets of every rank in the merged buffer: an 8-byte all-gather per step, off the conversion stream"""
        nonlocal counts
        k = n_published[0] % RING
        n_published[0] += 1
        counts = counts_ring[k]
        with torch.cuda.stream(side):
            mine_ring[k].fill_(total)
            pending.append(dist.all_gather_into_tensor(counts_ring[k], mine_ring[k], async_op=True))
            while len(pending) > 2:
                pending.pop(0).wait()    # (orders the side stream behind an all-gather issued two steps ago; the host does not block)

    def submit():
        if multi:
            conv.submit(R, out.data_ptr(), out.shape[0], stream)
        else:
            conv.submit(R)

    def step_sync():
        if not multi:
            return conv.convert(R)
        total = conv.convert_into(R, out.data_ptr(), out.shape[0], stream)
        publish(total)
        return total

    PROF_EVERY = 8        # HIP events bracket every 8th launch: an event pair costs ~10 us of stream time per launch
    n_prof = [0]

    def run_steps(k, timed=False):
        """k conversions, pipelined two deep; returns the last counter; accumulates the sampled kernel times"""
        total = 0
        if a.sync_steps:
            for i in range(k):
                conv.set_profiling(timed and i % PROF_EVERY == 0)
                total = step_sync()
                if timed and i % PROF_EVERY == 0:
                    n_prof[0] += 1
                    for n_, v in conv.last_kernel_ms().items():
                        kms[n_] += v
            return total
        conv.set_profiling(timed)
        submit()
        for i in range(k):
            if i + 1 < k:
                conv.set_profiling(timed and (i + 1) % PROF_EVERY == 0)
                submit()
            total = conv.wait()
            if multi:
                publish(total)
            if timed and i % PROF_EVERY == 0:
                n_prof[0] += 1
                for n_, v in conv.last_kernel_ms().items():
                    kms[n_] += v
        return total

    if multi:
        # size the per-rank record buffer once (like the SSBO (re)allocation, outside the timed region)
        probe = torch.empty((1, 24), dtype=torch.float32, device="cuda")
        need = conv.convert_into(R, probe.data_ptr(), 1, stream)
        out = torch.empty((need, 24), dtype=torch.float32, device="cuda")

    def sync():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    kms = {k: 0.0 for k in ("count", "scan", "offsets", "emit", "fused")}
    if a.warmup:
        run_steps(a.warmup)
    kms = {k: 0.0 for k in kms}
    sync()
    t0 = time.perf_counter()
    total = run_steps(a.steps, timed=True)   # HIP events on the launch stream around every PROF_EVERY-th launch
    if pending:
        with torch.cuda.stream(side):
            for w in pending:
                w.wait()
        pending.clear()
    sync()
    dt = time.perf_counter() - t0
    conv.set_profiling(False)
    # for the record: the same loop with one blocking call per step (outside the timed region)
    sync()
    s0 = time.perf_counter()
    per_step = []
    for _ in range(min(a.steps, 20)):
        p0 = time.perf_counter()
        step_sync()
        per_step.append((time.perf_counter() - p0) * 1e3)
    if pending:
        with torch.cuda.stream(side):
            for w in pending:
                w.wait()
        pending.clear()
    sync()
    sync_ms = (time.perf_counter() - s0) / min(a.steps, 20) * 1e3
    # also for the record: two lanes (two streams, chains and record buffers), three conversions in flight, so that
    # consecutive conversions overlap.  Not the headline: overlapped launches have no meaningful individual duration,
    # and the roofline above is about the kernel.
    overlapped = None
    if a.overlap_extra and not multi and conv.last_pipeline in ("team", "wave"):
        conv.set_async_lanes(2)
        n_ov = max(min(a.steps, 60), 6)
        for _ in range(4):
            conv.submit(R)
        for _ in range(4):
            conv.wait()
        sync()
        o0 = time.perf_counter()
        conv.submit(R); conv.submit(R)
        for i in range(n_ov):
            if i + 2 < n_ov:
                conv.submit(R)
            ov_total = conv.wait()
        sync()
        ov_ms = (time.perf_counter() - o0) / n_ov * 1e3
        conv.set_async_lanes(1)
        overlapped = {"ms_per_step": ov_ms, "value": ov_total / (ov_ms * 1e-3), "unit": "Gaussians/s",
                      "what": "m2s_set_async_lanes(2), three conversions in flight: consecutive conversions overlap on two streams"}
    # the two viewer passes that consume the records in the reference's frame: for the record, after the
    # timed region; never part of `value`
    viewer = None
    if not a.no_viewer_extra and not multi:
        try:
            viewer = viewer_extra(conv, R, total)
        except Exception as e:  # noqa: BLE001 - an extra must not take the headline down
            viewer = {"error": str(e)}
    per_step.sort()
    sync_stats = {"median": per_step[len(per_step) // 2], "p10": per_step[len(per_step) // 10], "p90": per_step[(len(per_step) * 9) // 10]}
    # a second denominator for the roofline: what a plain device-to-device copy reaches on this box (read + write bytes)
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

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    ntot = torch.tensor([total], dtype=torch.int64, device="cuda")
    if multi:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(ntot, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    n_all = int(ntot.item())
    ms_per_step = dt / a.steps * 1e3
    value = n_all / (dt / a.steps)

    # optional: the north-star record all-gather (xGMI-bound), measured outside the timed region
    gather = None
    if multi and not a.no_gather:
        nmax = int(counts.max().item())
        send = torch.zeros((nmax, 24), dtype=torch.float32, device="cuda")
        send[: out.shape[0]] = out
        recv = torch.empty((world * nmax, 24), dtype=torch.float32, device="cuda")
        for _ in range(2):
            dist.all_gather_into_tensor(recv, send)
        sync()
        g0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            step_sync()
            send[: out.shape[0]] = out
            dist.all_gather_into_tensor(recv, send)
        sync()
        gdt = torch.tensor([(time.perf_counter() - g0) / reps], dtype=torch.float64, device="cuda")
        dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
        gather = {"ms_per_step": float(gdt.item()) * 1e3, "value": n_all / float(gdt.item()), "unit": "Gaussians/s",
                  "what": "convert + padded RCCL all-gather of all records to every rank"}
        if a.gather_direct:      # the exact-size all-pairs schedule of dist.all_gather_records (grouped isend / irecv)
            from mesh2splat_amd import dist as m2d
            cl = [int(x) for x in counts.tolist()]
            for _ in range(2):
                m2d.all_gather_records(out[: cl[rank]], cl)
            sync()
            g0 = time.perf_counter()
            for _ in range(reps):
                step_sync()
                m2d.all_gather_records(out[: cl[rank]], cl)
            sync()
            gdt = torch.tensor([(time.perf_counter() - g0) / reps], dtype=torch.float64, device="cuda")
            dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
            gather["direct"] = {"ms_per_step": float(gdt.item()) * 1e3, "value": n_all / float(gdt.item()),
                                "what": "convert + exact-size all-pairs exchange (isend/irecv group) into the merged buffer"}

    if rank == 0:
        dom = "fused" if kms["fused"] > 0 else "emit"     # the dominant kernel of the pipeline that ran
        kname = {"team": "k_fused2", "wave": "k_fused"}.get(conv.last_pipeline, "k_emit")
        emit_ms = kms[dom] / max(n_prof[0], 1)
        # algorithmic bytes of one emit launch: 96 B per Gaussian written + 144 B per triangle read
        # (B_alg = 96 N + 144 T); textures, offsets and the entry list are not credited.
        b_alg = 96.0 * total + 144.0 * T_local
        achieved = b_alg / (emit_ms * 1e-3) / 1e9 if emit_ms > 0 else 0.0
        res = {
            "metric": "Gaussians/sec emitted (mesh->3DGS conversion pass, density 1024^2)" if R == 1024 else
                      f"Gaussians/sec emitted (mesh->3DGS conversion pass, density {R}^2)",
            "value": value, "unit": "Gaussians/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "ms_per_mesh": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"I-4 64 meshes x cube-sphere n=18 ({tri_per_mesh} triangles), 64 materials with 3 procedural "
                                    f"{tex}^2 RGBA8 maps each, R={R}") if n == "grid" else
                                   (f"I-3 cube-sphere n={n} ({tri_per_mesh} triangles/mesh) x {world} mesh(es), "
                                    f"3 procedural {tex}^2 RGBA8 maps, R={R}"),
                       "gaussians_per_step": n_all, "triangles_per_gpu": T_local, "parallelism": f"tri-range x{world}",
                       "cap": "unlimited (merged scene exceeds the 7M envelope)" if multi else "reference formula",
                       "submission": "one blocking call per step" if a.sync_steps else
                                     "pipelined 2 deep (m2s_convert_submit/wait): every conversion completes and its counter is read back in the timed region"},
            "sync_ms_per_step": sync_ms, "sync_ms_stats": sync_stats, "overlapped": overlapped, "viewer_passes": viewer,
            "kernel_ms": {k: v / max(n_prof[0], 1) for k, v in kms.items()},
            "kernel_timing": f"HIP events on the launch stream around every {PROF_EVERY}th launch of the timed region ({n_prof[0]} launches)",
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": kname,
                         "algorithmic_bytes": b_alg, "measured_copy_peak": copy_gbs,
                         "frac_of_measured_copy": (achieved / copy_gbs) if copy_gbs else None,
                         "write_only_frac": (96.0 * total / (emit_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if emit_ms > 0 else 0.0},
        }
        tr = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tr):
            try:
                with open(tr) as f:
                    t = json.load(f)
                if t.get("workload") == a.workload:
                    res["roofline"]["traffic"] = t.get(kname + "_hbm_bytes_per_launch")
            except Exception:
                pass
        if gather:
            res["gather"] = gather
        if not a.no_cpu_baseline and world == 1:
            one = scene if n == "grid" else synth.colocated_spheres(1, n, tex)
            res["cpu_baseline"] = cpu_baseline(one, R, a.cpu_seconds, gpu_total=total)
        print(json.dumps(res), flush=True)

    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
