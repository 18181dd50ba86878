"""-m gpu: the multi-pass conversion in ONE launch (k_multipass: chunks of triangle blocks, chunk k + 1 counted while chunk k is emitted;
m2s_pass.cpp: plan_chunks, enqueue_multipass) writes the bytes of the two-kernel conversion — whatever the cuts, at the density the
upload counted at (emitters sized exactly), at densities it did not (sized from R^2-scaled counts), under a cap, submitted
asynchronously, and when the prediction is so wrong that a chunk's emitters fall short (reported; repeated with the two kernels).
One process per setting: the library's debug switches are read once per process."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

_SCRIPT = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
hint = int(sys.argv[2])
scene = synth.sponza_like(tex_scale=0.125)
out = []
c = Converter(0)
c.set_pipeline("multipass")
if hint:
    c.set_resolution_hint(hint)
c.upload_scene(scene)
for R, cap in ((1024, -1), (1024, -1), (640, -1), (1536, -1), (1024, 1_500_000), (1024, 0)):
    c.set_max_gaussians(cap)
    total = c.convert(R)
    rec = c.download()
    out.append({"R": R, "cap": cap, "total": int(total), "stored": int(rec.shape[0]), "chunks": c.last_chunks,
                "sha": hashlib.sha256(rec.tobytes()).hexdigest()[:20]})
# asynchronous submissions of the chunked conversion, three in flight, interleaved densities
c.set_max_gaussians(-1)
for R in (1024, 640, 1024):
    c.submit(R)
totals = [int(c.wait()) for _ in range(3)]
rec = c.download()                      # (the records of the newest submission: the earlier ones were overwritten in order)
out.append({"R": 1024, "cap": -1, "async": totals, "total": totals[2], "stored": int(rec.shape[0]), "chunks": c.last_chunks,
            "sha": hashlib.sha256(rec.tobytes()).hexdigest()[:20]})
c.close()
print(json.dumps(out))
"""


def _run(tmp_path, env_extra, hint=1024):
    script = tmp_path / "chunks_probe.py"
    script.write_text(_SCRIPT)
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, str(script), ROOT, str(hint)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_one_launch_multipass_writes_the_bytes_of_the_two_kernel_conversion(hiplib, tmp_path):
    one = _run(tmp_path, {"M2S_DEBUG": "1", "M2S_NO_MERGED": "1"})
    assert all(e["chunks"] == 0 for e in one)
    auto = _run(tmp_path, {})
    assert auto[0]["chunks"] >= 2, "the heterogeneous scene at R = 1024 is expected to run as one launch"
    runs = {"auto": auto}
    for marks in ("0.5,1", "0.02,0.04,0.06,0.08,0.1,1", "0.3,0.31,0.9,1", "0.9,1"):
        runs["marks " + marks] = _run(tmp_path, {"M2S_DEBUG": "1", "M2S_MP_MARKS": marks})
        assert runs["marks " + marks][0]["chunks"] >= 2
    for name, got in runs.items():
        assert len(got) == len(one)
        for a, b in zip(got, one):
            assert (a["R"], a["cap"], a["total"], a["stored"], a["sha"]) == (b["R"], b["cap"], b["total"], b["stored"], b["sha"]), (name, a, b)


def test_emitters_that_fall_short_are_reported_and_the_conversion_repeated(hiplib, tmp_path):
    """The upload counts at R = 48, where the foliage and most of the cloth cover no pixel centre: scaled by R^2 those counts
    say nothing about R = 1024, the chunks' emitters are far too few, the kernel reports it and the conversion is repeated with two kernels."""
    one = _run(tmp_path, {"M2S_DEBUG": "1", "M2S_NO_MERGED": "1"}, hint=48)
    forced = _run(tmp_path, {"M2S_DEBUG": "1", "M2S_MP_MARKS": "0.4,0.8,1"}, hint=48)
    for a, b in zip(forced, one):
        assert (a["R"], a["total"], a["stored"], a["sha"]) == (b["R"], b["total"], b["stored"], b["sha"]), (a, b)
    assert forced[0]["chunks"] == 0 or forced[1]["chunks"] == 0, "R = 1024: the short launch must have led to the two-kernel conversion"
