"""-m gpu: the short correctly rounded sequences of mesh2splat_amd/csrc/m2s_exact.h against the compiler's IEEE expansions, by
exhaustion on the GPU the suite runs on (tests/exact_math/exact_math_check.hip includes the shipped header).

  rcp_rn   every |x| in [2^-64, 2^64], both signs      sqrt_rn   every x in [2^-96, 2^100] (+ rcp_rn(sqrt_rn(x)) vs 1.0f / sqrtf(x))
  div_rn   divisor significands: both ends of [1, 2) and random slices, each against ALL 2^23 dividend significands (1.2e11 pairs),
           + 2^24 random pairs over the whole guarded exponent range; M2S_TEST_DIVALL=1: all 2^23 divisors (7.0e13 pairs, ~45 s —
           the run recorded in profiles/r06/exact_math_exhaustive.jsonl)
The reciprocal and the square root start from hardware seeds (v_rcp_f32, v_rsq_f32), so this is the proof of those two, and it is
re-run wherever the suite runs; the division identity is seed-free (tests/test_round6_math.py has it on the CPU as well)."""
import json
import os
import random
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "exact_math", "_build", "exact_math_check")


def run(*args, timeout=600):
    if not os.path.exists(EXE):
        subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(EXE))], check=True, stdout=subprocess.DEVNULL)
    r = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    return [json.loads(ln) for ln in r.stdout.splitlines() if ln.startswith("{")]


def shipped(lines):
    return [ln for ln in lines if "candidate" in ln and "(shipped)" in ln["candidate"]]


def test_reciprocal_every_operand_of_its_range():
    lines = run("rcp")
    s = shipped(lines)
    assert len(s) == 1 and s[0]["mismatches"] == 0, s
    alone = [ln for ln in lines if ln["candidate"] == "v_rcp_f32 alone"][0]
    assert alone["mismatches"] > 0          # (the seed alone is NOT correctly rounded: the check can tell the difference)


def test_square_root_every_operand_of_its_range_and_the_reciprocal_length():
    lines = run("sqrt")
    s = shipped(lines)
    assert len(s) == 1 and s[0]["mismatches"] == 0, s
    comp = [ln for ln in lines if ln["candidate"].startswith("rcp_rn(sqrt_rn(x))")][0]
    assert comp["mismatches"] == 0, comp
    assert [ln for ln in lines if ln["candidate"] == "v_sqrt_f32 alone"][0]["mismatches"] > 0


def test_division_divisor_slices_against_all_dividends():
    if os.environ.get("M2S_TEST_DIVALL") == "1":
        lines = run("divall", 900, timeout=1200)
        assert lines[-1]["complete"] is True and lines[-1]["divisors_done"] == 1 << 23
        assert all(ln["mismatches"] == 0 for ln in shipped(lines)), lines
        return
    rng = random.Random(0xD1F)
    starts = [0, (1 << 23) - 4096] + [rng.randrange(0, (1 << 23) - 1024) for _ in range(6)]
    for k, m0 in enumerate(starts):
        n = 4096 if k < 2 else 1024
        lines = run("div", m0, n)
        assert lines[-1]["complete"] is True and lines[-1]["divisors_done"] == n
        s = shipped(lines)
        assert len(s) == 2 and all(ln["mismatches"] == 0 for ln in s), (m0, s)
