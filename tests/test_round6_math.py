"""CPU-only: the arithmetic identities behind round 6's short IEEE sequences (mesh2splat_amd/csrc/m2s_exact.h), by enumeration.

The conversion kernels' DECISION arithmetic contains correctly rounded divisions, reciprocals and square roots (converterGS.glsl:326-399:
`rel / range`, `normalize`, `length`); round 6 replaced the compiler's 11-16-instruction expansions by 3-5-instruction sequences that
return the same bits.  The complete proof runs ON the GPU (tests/test_gpu_exact_math.py; all 7.0e13 significand pairs of a / b in
profiles/r06/exact_math_exhaustive.jsonl).  Here, without a GPU:

  * the DIVISION identity  RN(a y + (a - RN(a y) b) y) == RN(a / b), y = RN(1 / b)  involves no hardware approximation — only fp32
    multiply and FMA, which the CPU executes identically: a sample of divisor significands (edge cases + random) against ALL 2^23
    dividend significands;
  * the reciprocal / square-root sequences are exact from a correctly rounded seed for every significand, and NOT from a seed one
    ulp off — i.e. they rest on the actual v_rcp_f32 / v_rsq_f32 seeds, and their proof has to be (and is) the on-device exhaustion;
  * the guard ranges of m2s_exact.h keep every intermediate normal and the residual exact (the scale-invariance argument).
"""
import json
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "exact_math", "div_identity.c")
HDR = os.path.join(ROOT, "mesh2splat_amd", "csrc", "m2s_exact.h")


@pytest.fixture(scope="module")
def identity_line(tmp_path_factory):
    if "fma" not in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split():
        pytest.skip("this CPU has no fused multiply-add")
    exe = str(tmp_path_factory.mktemp("exact") / "div_identity")
    subprocess.run(["gcc", "-O2", "-mfma", "-fopenmp", "-ffp-contract=off", "-o", exe, SRC, "-lm"], check=True)
    out = subprocess.run([exe, "96", "20260930"], check=True, capture_output=True, text=True, timeout=600).stdout
    return json.loads(out)


def test_division_by_a_correctly_rounded_reciprocal_and_one_residual_step_is_the_ieee_quotient(identity_line):
    assert identity_line["divisors"] == 96 and identity_line["pairs"] == 96 << 23
    assert identity_line["div_mismatches"] == 0, identity_line


def test_reciprocal_and_root_sequences_are_exact_from_the_rounded_seed_but_depend_on_the_seed(identity_line):
    lo, mid, hi = identity_line["rcp_mismatches_by_seed_offset"]
    assert mid == 0 and lo + hi > 0, identity_line        # exact from RN(1/x); a seed one ulp off breaks some significands
    lo, mid, hi = identity_line["sqrt_mismatches_by_seed_offset"]
    assert mid == 0 and lo + hi > 0, identity_line


def consts():
    txt = open(HDR).read()
    out = {}
    for name in ("kSqrtLo", "kSqrtHi", "kRcpLo", "kRcpHi", "kDivLo", "kDivHi"):
        m = re.search(name + r" = (0x1p-?\d+)f", txt)
        assert m, name
        out[name] = float.fromhex(m.group(1))
    return out


def test_guard_ranges_keep_every_intermediate_normal_and_the_residual_exact():
    c = consts()
    tiny, huge = float(np.finfo(np.float32).tiny), float(np.finfo(np.float32).max)
    # sqrt_rn: y = rsq(x) in [2^-50, 2^48], s = x y in [2^-48, 2^50], residual x - s s ~ 2^-23 x >= 2^-119: all normal; its root lies inside rcp_rn's range
    assert tiny * 2 ** 23 <= c["kSqrtLo"] and c["kSqrtHi"] ** 0.5 <= c["kRcpHi"] and c["kSqrtLo"] ** 0.5 >= c["kRcpLo"]
    # rcp_rn: y = 1/x in [2^-64, 2^64]; e = 1 - x y ~ 2^-24: normal
    assert 1.0 / c["kRcpHi"] >= tiny and c["kRcpHi"] <= huge
    # div_rn: |a|, b in [lo, hi]: the divisor is inside rcp_rn's range, |q| = |a / b| in [lo/hi, hi/lo] is normal, and the residual
    # a - q b is a multiple of ulp(q) ulp(b) >= 2^(ea - 47): representable (>= 2^-149) for ea >= -102
    assert c["kRcpLo"] <= c["kDivLo"] and c["kDivHi"] <= c["kRcpHi"]
    assert c["kDivLo"] / c["kDivHi"] >= tiny * 2 and c["kDivHi"] / c["kDivLo"] <= huge / 2
    assert np.log2(c["kDivLo"]) - 47 >= -149


def test_the_on_device_check_includes_the_shipped_header_not_a_copy():
    chk = open(os.path.join(ROOT, "tests", "exact_math", "exact_math_check.hip")).read()
    assert '#include "../../mesh2splat_amd/csrc/m2s_exact.h"' in chk
    for fn in ("rcp_rn", "sqrt_rn", "div_rn"):
        assert len(re.findall(r"float " + fn + r"\(", chk)) == 0, fn       # used, never redefined
        assert fn + "(" in chk


def test_floor_division_by_an_approximate_reciprocal_plus_one_integer_correction_is_exact():
    """floordivmod_by (m2s_devfn.h): q0 = floor(fp64(num) * y), y ~ 1/den, then ONE correction step on the exact remainder.  Claim: exact
    for |num| < 2^52, 256 <= den < 2^31, |num / den| < 2^44, as long as y's relative error is at most 2^-50.  Enumerated here with
    reciprocals perturbed by up to 2^-46 (16x the bound the two Newton steps guarantee), on random operands and on the adversarial
    ones — numerators within a few units of a multiple of the divisor, where floor() of an approximate quotient lands on the wrong side."""
    rng = np.random.default_rng(0xF100D)
    n_cases = 400_000
    den = np.concatenate([rng.integers(256, 1 << 31, n_cases // 2), 256 * rng.integers(1, 1 << 23, n_cases // 2)]).astype(np.int64)
    k = rng.integers(-(1 << 43), 1 << 43, n_cases).astype(np.int64)
    k = np.clip(k, -((1 << 52) - 1) // den, ((1 << 52) - 1) // den)
    num = k * den + rng.integers(-3, 4, n_cases)                    # multiples of the divisor, a few units either side
    num[: n_cases // 4] = rng.integers(-(1 << 52) + 1, 1 << 52, n_cases // 4)
    num[n_cases // 4: n_cases // 2] = rng.integers(-(1 << 31), 1 << 31, n_cases // 4)   # the walkers' per-row steps (|256 b| < 2^31, x 64: < 2^37)
    num = np.clip(num, -(1 << 52) + 1, (1 << 52) - 1)
    want = np.array([int(a) // int(b) for a, b in zip(num, den)], dtype=np.int64)
    for rel in (0.0, 2.0 ** -50, -(2.0 ** -50), 2.0 ** -46, -(2.0 ** -46)):
        y = (1.0 / den.astype(np.float64)) * (1.0 + rel)
        q = np.floor(num.astype(np.float64) * y).astype(np.int64)
        assert np.abs(q - want).max() <= 1                                             # the approximate floor is never more than one off ...
        r = num - q * den
        q = np.where(r < 0, q - 1, np.where(r >= den, q + 1, q))                      # ... which one correction step repairs
        assert np.array_equal(q, want), rel
        r = num - q * den
        assert ((r >= 0) & (r < den)).all()
