#!/usr/bin/env python
"""After tools/r4_final.sh (r3_final.sh): fold the PMC summaries of the closing run into profiles/pmc_traffic.json (HBM bytes per launch =
2 x FETCH_SIZE KB (gfx950 tallies 128-byte requests at 64) + WRITE_SIZE KB x the k_repack calibration) together with the sha of
the library they were measured on, and copy the evidence files to profiles/<round>/.   usage: python tools/pmc_update.py [tag] [round dir, default r04]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, "gpurun_out")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r4fin"
RND = sys.argv[2] if len(sys.argv) > 2 else "r04"
P = os.path.join(ROOT, "profiles", RND)
os.makedirs(P, exist_ok=True)
path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
t = json.load(open(path))
sha = open(os.path.join(O, TAG + "_binary_sha.txt")).read().strip()
WCAL = 144.3e6 / (140942 * 1024.0)      # WRITE_SIZE calibrated on k_repack (144.3 MB written, 140 942 KB reported)


def traffic(d):
    return 2 * d["FETCH_SIZE"] * 1024.0, d["WRITE_SIZE"] * 1024.0 * WCAL


c3 = json.load(open(os.path.join(O, TAG + "_pmc_c3_summary.json")))["m2s::k_fused2"]
c5 = json.load(open(os.path.join(O, TAG + "_pmc_c5_summary.json")))["m2s::k_sparse"]
c2 = json.load(open(os.path.join(O, TAG + "_pmc_c2_summary.json")))["m2s::k_fused2"]
r, w = traffic(c3)
t.update({"FETCH_SIZE_KB": c3["FETCH_SIZE"], "WRITE_SIZE_KB": c3["WRITE_SIZE"], "k_fused2_read_bytes": r, "k_fused2_write_bytes": w,
          "k_fused2_hbm_bytes_per_launch": r + w, "binary_sha": {"k_fused2": sha, "k_sparse": sha}})
t["source_round" + RND[-1]] = ("profiles/" + RND + "/final_pmc_c3_summary.json, final_pmc_c5_summary.json, final_pmc_c2_summary.json (tools/" + RND.replace("0", "") + "_final.sh: separate --pmc "
                      "passes FETCH_SIZE / WRITE_SIZE / two SQ sets, 23 blocking launches each for c3 and c2, 8 for c5; mean per launch); library sha256[:16] " + sha)
r5, w5 = traffic(c5)
t["c5"] = {"workload": "c5 at full size (50 037 168 triangles, 24 267 048 Gaussians)", "kernel": "k_sparse", "algorithmic_bytes": 9534988800.0,
           "k_sparse_read_bytes": r5, "k_sparse_write_bytes": w5, "k_sparse_hbm_bytes_per_launch": r5 + w5,
           "traffic_over_algorithmic": (r5 + w5) / 9534988800.0}
r2, w2 = traffic(c2)
t["c2"] = {"workload": "c2 stand-in (69 312 triangles, R = 512, 684 624 Gaussians)", "kernel": "k_fused2", "algorithmic_bytes": 75704832.0,
           "k_fused2_read_bytes": r2, "k_fused2_write_bytes": w2, "k_fused2_hbm_bytes_per_launch": r2 + w2,
           "traffic_over_algorithmic": (r2 + w2) / 75704832.0}
json.dump(t, open(path, "w"), indent=1)
for w_ in ("c3", "c2", "c5"):
    shutil.copy(os.path.join(O, f"{TAG}_pmc_{w_}_summary.json"), os.path.join(P, f"final_pmc_{w_}_summary.json"))
for src, dst in ((f"{TAG}_trace_bench/k_kernel_stats.csv", "final_bench_kernel_stats.csv"), (f"{TAG}_trace_c3/k_kernel_stats.csv", "final_c3_only_kernel_stats.csv"),
                 (f"{TAG}_bench.json", "final_bench.json"), (f"{TAG}_trace_bench.json", "final_bench_under_rocprofv3.json"), (f"{TAG}_tests.log", "final_gpu_tests.log")):
    if os.path.exists(os.path.join(O, src)):
        shutil.copy(os.path.join(O, src), os.path.join(P, dst))
print("sha", sha, "c3", r + w, (r + w) / 407207616.0, "c5", t["c5"]["traffic_over_algorithmic"], "c2", t["c2"]["traffic_over_algorithmic"])
