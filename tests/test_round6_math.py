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


# ---- the 32-bit row walker (m2s_devfn.h: RowWalker32, row_walker32_init / _next) ---------------------------------------------------
def _span_reference(a, b, c, bias, x0, x1, y):
    Py = 256 * y + 128
    lo, hi = x0, x1
    for i in range(3):
        alpha = 256 * a[i]
        beta = 128 * a[i] + b[i] * Py + c[i] + bias[i]
        if alpha > 0:
            lo = max(lo, (alpha - beta) // alpha)
        elif alpha < 0:
            hi = min(hi, (beta - 1) // -alpha)
        elif beta < 1:
            hi = lo - 1
    return max(hi - lo + 1, 0), lo


def _walker32(a, b, c, bias, x0, x1, y, stride, n):
    """Transcription of row_walker32_init + the caller's loop: list of (k, count, first x) of the rows it visits, and `safe`."""
    W = 1 << 29
    k0, k1 = 0, n - 1
    Py = 256 * y + 128
    for i in range(3):
        if a[i] == 0:
            beta = b[i] * Py + c[i] + bias[i]
            bs = 256 * b[i] * stride
            if bs > 0:
                k0 = max(k0, (bs - beta) // bs)
            elif bs < 0:
                k1 = min(k1, (beta - 1) // -bs)
            elif beta < 1:
                k1 = -1
    if k0 > k1:
        return [], True
    yy = y + k0 * stride
    steps = k1 - k0
    Py = 256 * yy + 128
    q, r, sq, sr, D, lower = [0] * 3, [0] * 3, [0] * 3, [0] * 3, [1] * 3, [True] * 3
    safe = True
    for i in range(3):
        alpha = 256 * a[i]
        beta = 128 * a[i] + b[i] * Py + c[i] + bias[i]
        bs = 256 * b[i] * stride
        if alpha == 0:
            q[i] = -2 * W
            continue
        d = abs(alpha)
        nmr = alpha - beta if alpha > 0 else beta - 1
        st = -bs if alpha > 0 else bs
        q[i], r[i] = nmr // d, nmr % d
        sq[i], sr[i] = st // d, st % d
        D[i] = d
        lower[i] = alpha > 0
        # (the kernel's test: the exact first quotient by its sign extension, the last one through an fp32 estimate against 2^28)
        qe = abs(np.float32(np.float32(max(min(q[i], 2 ** 31 - 1), -2 ** 31)) + np.float32(np.float32(steps) * np.float32(sq[i]))))
        safe = safe and -W <= q[i] < W and bool(qe < np.float32(268435456.0))
    rows = []
    if not safe:
        return rows, False
    for k in range(k0, k1 + 1):
        lo, hi = x0, x1
        for i in range(3):
            assert -(1 << 31) <= q[i] < (1 << 31), "a quotient left 32 bits although the init called the triangle safe"
            if lower[i]:
                lo = max(lo, q[i])
            else:
                hi = min(hi, q[i])
            t = r[i] + sr[i]
            assert t < (1 << 32)
            carry = t >= D[i]
            r[i] = t - D[i] if carry else t
            q[i] += sq[i] + (1 if carry else 0)
        assert -(1 << 31) <= hi - lo + 1 < (1 << 31)
        rows.append((k, max(hi - lo + 1, 0), lo))
    return rows, True


def _edges(X, Y):
    area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0])
    if area2 == 0:
        return None
    sgn = -1 if area2 < 0 else 1
    a, b, c, bias = [], [], [], []
    for i in range(3):
        ia, ib = (i + 1) % 3, (i + 2) % 3
        dy, dx = Y[ib] - Y[ia], X[ib] - X[ia]
        a.append(-dy * sgn); b.append(dx * sgn); c.append((dy * X[ia] - dx * Y[ia]) * sgn)
        bias.append(1 if (a[i] > 0 or (a[i] == 0 and b[i] > 0)) else 0)
    return a, b, c, bias


def test_walker32_visits_the_closed_form_spans_and_only_skips_empty_rows():
    """Every row the 32-bit walker visits has row_span's span; every row it skips (outside [k0, k1]: a horizontal edge fails it) is
    empty; a triangle it declines (a quotient beyond 2^29 somewhere over its rows) is declined at the init, never mid-walk.  Random
    triangles, triangles with an exactly horizontal edge (the rings of a cylinder, every axis-aligned quad), slivers whose long edge is
    one sub-pixel off horizontal (the declined class), both strides."""
    rng = np.random.default_rng(20260930)
    seen_h = seen_unsafe = seen_skip = 0
    for it in range(1500):
        R = int(rng.choice([256, 1024, 4096]))
        kind = it % 5
        X = [int(v) for v in rng.integers(-2000, R * 256 + 2000, 3)]
        Y = [int(v) for v in rng.integers(-2000, R * 256 + 2000, 3)]
        if kind == 1:      # one exactly horizontal edge, sometimes on a pixel-centre row
            Y[1] = Y[0] = int(rng.integers(0, R)) * 256 + int(rng.choice([128, 0, 77]))
        elif kind == 2:    # sliver: nearly horizontal long edge
            Y[1] = Y[0] + int(rng.choice([-2, -1, 1, 2])); X[1] = X[0] + int(rng.choice([-1, 1])) * int(rng.integers(R * 128, R * 256))
            Y[2] = int(rng.integers(-2000, R * 256 + 2000))
        elif kind == 3:    # small triangle
            X = [X[0] + int(v) for v in rng.integers(-2000, 2000, 3)]; Y = [Y[0] + int(v) for v in rng.integers(-2000, 2000, 3)]
        e = _edges(X, Y)
        if e is None:
            continue
        a, b, c, bias = e
        x0, x1 = max((min(X) - 128 + 255) >> 8, 0), min((max(X) - 128) >> 8, R - 1)
        y0, y1 = max((min(Y) - 128 + 255) >> 8, 0), min((max(Y) - 128) >> 8, R - 1)
        if x0 > x1 or y0 > y1:
            continue
        seen_h += 0 in a
        for stride, lanes in ((1, (0,)), (64, (0, 1, 17, 63))):
            for lane in lanes:
                y = y0 + lane
                if y > y1:
                    continue
                n = (y1 - y) // stride + 1
                rows, safe = _walker32(a, b, c, bias, x0, x1, y, stride, n)
                if not safe:
                    seen_unsafe += 1
                    continue
                visited = {k: (cnt, lo) for k, cnt, lo in rows}
                for k in range(n):
                    want, wlo = _span_reference(a, b, c, bias, x0, x1, y + k * stride)
                    if k in visited:
                        assert visited[k][0] == want and (want == 0 or visited[k][1] == wlo), (X, Y, stride, lane, k)
                    else:
                        seen_skip += 1
                        assert want == 0, (X, Y, stride, lane, k)
    assert seen_h > 100 and seen_unsafe > 20 and seen_skip > 20


def test_walker32_is_what_the_kernel_source_says():
    """The transcription above follows the shipped header: the constants and the carry rule it depends on are in m2s_devfn.h."""
    src = open(os.path.join(ROOT, "mesh2splat_amd", "csrc", "m2s_devfn.h")).read()
    assert "constexpr long long kWalk32 = 1ll << 29;" in src
    assert "q = -2 * kWalk32; w.D[i] = 1u;" in src
    assert "const bool c = t >= w.D[i];" in src and "w.q[i] += w.sq[i] + (c ? 1 : 0);" in src
    assert "safe = safe && (q >> 29) == (q >> 63) && qe < 268435456.0f;" in src
