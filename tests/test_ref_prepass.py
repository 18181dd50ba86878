"""CPU: the prepass oracle against the REFERENCE'S OWN pass and shader.

oracle/ref_prepass_check.cpp links the reference's GaussiansPrepass.cpp and executes gaussianSplattingPrepassCS.glsl
(+ common.glsl; rewritten only syntactically by oracle/glsl2cpp.py) through the vendored glm on a minimal software GL.
Compared: the atomic counter, every QuadNdcTransformation and every depth — BIT FOR BIT, NaN matching NaN (the oracle restates the shader
and glm's operators operation for operation in IEEE fp32; sin/exp come from the same libm on both sides).

  * golden — reference outputs committed under tests/golden/ref_host/prepass_* ; always runs.
  * live   — every case on more records through the binary; skipped where oracle/_ref was not built."""
import os

import numpy as np
import pytest

import prepass_cases
import refhost

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_host")
CASES = prepass_cases.cases()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits(a, b):
    """Bit-identical, except that a NaN matches any NaN (its sign and payload carry no meaning and depend on how the
    compiler orders a negation)."""
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def assert_same(ref, orc, what):
    rk, rq, rd = ref
    ok, oq, od = orc
    assert rk == ok, f"{what}: visible count {ok} != reference {rk}"
    assert same_bits(rq, oq).all(), f"{what}: quads differ in {np.argwhere(~same_bits(rq, oq))[:5].tolist()}"
    assert same_bits(rd, od).all(), f"{what}: depths differ"


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_oracle_matches_reference_prepass_golden(oracle, name):
    p = dict(CASES)[name]
    with open(os.path.join(GOLD, "prepass_records.bin"), "rb") as f:
        rec = np.frombuffer(f.read(), np.float32).reshape(-1, 24)
    with open(os.path.join(GOLD, f"prepass_{name}.out.bin"), "rb") as f:
        ref = refhost.parse_prepass_output(f.read())
    assert_same(ref, oracle.prepass(p, rec), name)
    if name == "colour":
        assert 0 < ref[0] < rec.shape[0]          # the case culls something and keeps something


@pytest.mark.skipif(not refhost.prepass_available(), reason="oracle/_ref/ref_prepass_check not built (no /root/reference)")
@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_oracle_matches_reference_prepass_live(oracle, tmp_path, name):
    p = dict(CASES)[name]
    rec = np.concatenate([prepass_cases.base_records(oracle, 14, 64), prepass_cases.hostile_records(2048)])
    k, q, d, info = refhost.run_prepass(p, rec, str(tmp_path))
    assert info["dispatches"] == 1 and info["n"] == rec.shape[0]
    assert_same((k, q, d), oracle.prepass(p, rec), name)


@pytest.mark.skipif(not refhost.prepass_available(), reason="oracle/_ref/ref_prepass_check not built (no /root/reference)")
def test_reference_dispatch_shape_and_empty_input(oracle, tmp_path):
    """GaussiansPrepass.cpp:44-49: ceil(sqrt(groups)) x ceil(groups / that) groups of 16x16; zero Gaussians dispatch (0,0)."""
    p = dict(CASES)["geometry_mode"]
    rec = np.tile(prepass_cases.base_records(oracle, 6, 24)[:1], (256 * 5 + 1, 1))   # 6 groups -> 3 x 2
    k, q, d, info = refhost.run_prepass(p, rec, str(tmp_path))
    assert info["groups"] == [3, 2]
    assert_same((k, q, d), oracle.prepass(p, rec), "dispatch shape")      # random2d depends on (x, y) of the invocation
    k, q, d, info = refhost.run_prepass(p, rec[:0], str(tmp_path))
    assert k == 0 and info["groups"] == [0, 0]
    assert oracle.prepass(p, rec[:0])[0] == 0
