"""-m gpu: the persistent form of the single-pass kernel (k_fused2p: at most as many workgroups as the GPU holds, units of 256
triangles handed out by tickets, eight in-order queues, work stealing between them) — a measured NEGATIVE result that stays in the
library behind the debug switch M2S_PERSIST.  It must stay bit-identical to k_fused2: separate
processes, because the switch is read once per process."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from mesh2splat_amd import synth
from mesh2splat_amd.converter import Converter
kind, out = sys.argv[1], sys.argv[2]
if kind == "sphere":
    scene, R = synth.cube_sphere(160, tex_size=256), 700      # 307 200 triangles = 1 200 units: more than one generation
else:
    scene, R = synth.sphere_grid(2, n=60, tex_size=64), 300    # 8 meshes (345 600 triangles), cumulative bounding boxes, units that straddle meshes
c = Converter(0)
c.set_pipeline("team")
c.set_max_gaussians(0)
c.set_resolution_hint(R)
c.upload_scene(scene)
res = {}
for r in (R, R, R - 16, R - 16):        # banded from the count at upload, banded again, without bands at a new density, then with
    total = c.convert(r)
    assert c.last_pipeline == "team"
    res[f"n{len(res)}"] = c.download()
for _ in range(3):
    c.submit(R)
for _ in range(3):
    assert c.wait() == res["n0"].shape[0]
res["async"] = c.download()
np.savez(out, **res)
""" % ROOT


@pytest.mark.parametrize("kind", ["sphere", "grid"])
def test_persistent_form_is_bit_identical(hiplib, tmp_path, kind):
    outs = {}
    for name, env in (("plain", {}), ("persistent", {"M2S_DEBUG": "1", "M2S_PERSIST": "1"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, "-c", SCRIPT, kind, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for k in outs["plain"].files:
        a, b = outs["plain"][k], outs["persistent"][k]
        assert a.shape == b.shape and a.shape[0] > 50_000
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{kind} {k}: persistent and plain forms differ"
