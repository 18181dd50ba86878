"""-m gpu: bench.py's own consistency on one GPU.
  * `--gpus 1 --force-dist` (the multi-GPU code path with ONE rank: real RCCL communicator behind the C ABI, caller-owned record buffer,
    per-step counter all-gather) must print the `value` of the plain N = 1 line within a few per cent — the N = 1 agreement a SCALE
    record is checked against its BENCH record with;
  * the line's repetition record and its end-to-end leg are present and consistent."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
QUIET = ["--no-extra-workloads", "--no-cold", "--no-cpu-baseline", "--no-viewer-extra", "--no-overlap-extra", "--no-c5"]


def run_bench(*args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "M2S_RCCL_PATH"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_force_dist_with_one_rank_agrees_with_the_plain_line(hiplib):
    common = ["--gpus", "1", "--steps", "400", "--warmup", "20", "--reps", "5", "--no-end-to-end", "--no-gather", "--no-strong-scaling"] + QUIET
    # the repetitions of one run spread by up to 20 % on a shared box (host jitter: two of five are often outliers), so the two code
    # paths are compared on their best repetition and on the line's value, the median repetition's (within 10 %).  Measured on four
    # boxes: the forced path is 0.4 ... 3.6 % slower — it really does more per step (an 8-byte RCCL all-gather on the exchange's own
    # stream beside every conversion, a caller-owned buffer under the unlimited cap): 6 % is the bar, and a pair of runs that misses
    # it is repeated once (best repetition of both pairs) before the test fails.  (The driver's N = 1 SCALE run is `--gpus 1`
    # WITHOUT --force-dist, i.e. the plain line itself.)
    pm = dm = float("inf")
    for attempt in range(2):
        plain = run_bench(*common)
        dist = run_bench(*common, "--force-dist")
        assert plain["config"]["gaussians_per_step"] == dist["config"]["gaussians_per_step"] == 2738368
        assert dist["exchange_transport"].startswith("rccl") and dist["scale_record"]["rccl_ranks"] == 1 and not dist["scale_record"]["dry_scale"]
        pm, dm = min(pm, plain["ms_per_step_reps"]["min"]), min(dm, dist["ms_per_step_reps"]["min"])
        if abs(dm / pm - 1.0) < 0.06 and abs(dist["value"] / plain["value"] - 1.0) < 0.10:
            break
    assert abs(dm / pm - 1.0) < 0.06, (plain["ms_per_step_reps"], dist["ms_per_step_reps"])
    assert abs(dist["value"] / plain["value"] - 1.0) < 0.10, (plain["value"], dist["value"], plain["ms_per_step_reps"], dist["ms_per_step_reps"])


def test_repetitions_and_end_to_end_are_in_the_line(hiplib):
    line = run_bench("--gpus", "1", "--steps", "20", "--warmup", "5", "--workload", "c2", *QUIET)
    assert line["steps"] == 20 and line["warmup"] == 5
    reps = line["ms_per_step_reps"]
    assert reps["reps"] == 5 and len(reps["all"]) == 5 and reps["min"] <= reps["median"] <= reps["max"]
    assert line["ms_per_step"] == reps["median"] and abs(line["value"] - line["config"]["gaussians_per_step"] / (line["ms_per_step"] * 1e-3)) < 1e-3 * line["value"]
    e2e = line["end_to_end_ms"]
    for fmt in ("format0", "format1"):
        t = e2e[fmt]
        assert t["total_ms"] > 0 and t["export_ms"] > 0 and t["upload_ms"] > 0 and t["ply_bytes"] > 0
        assert t["total_ms"] <= t["wall_ms_whole_process"]
    n = line["config"]["gaussians_per_step"]
    assert e2e["format0"]["ply_bytes"] > 248 * n and e2e["format1"]["ply_bytes"] > 76 * n       # header + rows of the stored Gaussians
