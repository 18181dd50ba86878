"""The heterogeneous scene (synth.sponza_like) under every pipeline setting: blocking ms, kernel ms by HIP events, what ran, and that
every setting writes the same bytes; AUTO's records against the oracle.  One JSON line per setting on stdout.
usage: python tools/hetero_probe.py [--R 1024] [--combo-only] [--no-oracle] [--reps 24]"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--R", type=int, default=1024)
    ap.add_argument("--combo-only", action="store_true")
    ap.add_argument("--no-oracle", action="store_true")
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--settings", default="auto,multipass,team,lean,sparse,wave")
    a = ap.parse_args()
    import torch  # noqa: F401  (the HIP runtime torch ships)
    from mesh2splat_amd import synth
    from mesh2splat_amd.converter import Converter
    scene = synth.sponza_like(combo_only=a.combo_only)
    T = scene.n_triangles
    ref_sha, first = None, None
    for name in a.settings.split(","):
        conv = Converter(0)
        try:
            conv.set_pipeline(name)
            conv.set_resolution_hint(a.R)
            conv.upload_scene(scene)
            total = conv.convert(a.R)
            conv.convert(a.R)
            conv.set_profiling(True)
            wall, kern, kd = [], [], {}
            for _ in range(a.reps):
                t0 = time.perf_counter()
                conv.convert(a.R)
                wall.append((time.perf_counter() - t0) * 1e3)
                kd = conv.last_kernel_ms()
                kern.append(sum(kd.values()))
            conv.set_profiling(False)
            rec = conv.download()
            sha = hashlib.sha256(rec.tobytes()).hexdigest()[:16]
            if first is None:
                first, ref_sha = rec, sha
            b = 96.0 * conv.num_stored + 144.0 * T
            ms, kms = float(np.median(wall)), float(np.median(kern))
            print(json.dumps({"setting": name, "ran": conv.last_pipeline, "R": a.R, "triangles": T, "gaussians": int(total), "stored": conv.num_stored,
                              "blocking_ms": ms, "kernels_ms": kms, "kernel_ms_last": {k: round(v, 4) for k, v in kd.items() if v},
                              "frac_of_hbm_peak_blocking": b / (ms * 1e-3) / 8e12, "frac_of_hbm_peak_kernels": b / (kms * 1e-3) / 8e12,
                              "records_sha": sha, "same_bytes_as_first_setting": sha == ref_sha}), flush=True)
        except Exception as e:  # noqa: BLE001
            print(json.dumps({"setting": name, "error": repr(e)}), flush=True)
        conv.close()
    if not a.no_oracle and first is not None:
        from oracle import oracle
        import parity
        t0 = time.perf_counter()
        n, ref, _ = oracle.convert(scene, a.R, n_threads=8)
        assert ref.shape[0] == first.shape[0], (ref.shape, first.shape)
        frac = parity.assert_records_match(first, ref, "sponza_like vs oracle")
        print(json.dumps({"oracle": "records match (1e-4 rule, tests/parity.py)", "oracle_total": int(n), "bit_identical_floats": frac,
                          "oracle_s": time.perf_counter() - t0}), flush=True)


if __name__ == "__main__":
    main()
