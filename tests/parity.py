"""Shared comparison helpers for the parity tests (GPU records vs CPU oracle records).

Tolerance (north_star: "within 1e-4 relative float tolerance"):
  * independent scalars — colour rgba, scale xyz, metallic/roughness — are compared per component:
    |gpu - ref| <= 1e-4 * |ref| + 1e-6;
  * vector-valued fields — position, normal, rotation quaternion — are compared relative to the
    vector's magnitude: |gpu - ref| <= 1e-4 * max|ref_vector| + 1e-7 per component.  (A component of a
    unit normal that happens to be ~1e-3 cannot be reproduced to 1e-4 of ITSELF by two
    implementations that both carry ~1e-7 of rounding error relative to the vector; what is
    meaningful is the error relative to the vector.)
Counts and record order must match exactly; NaNs must match NaNs.
"""
import numpy as np

RTOL = 1e-4
ATOL_SCALAR = 1e-6
ATOL_VECTOR = 1e-7

VECTOR_FIELDS = (slice(0, 4), slice(12, 16), slice(16, 20))    # position, normal, rotation
SCALAR_FIELDS = (slice(4, 8), slice(8, 12), slice(20, 24))     # color, scale, pbr


def assert_records_match(gpu: np.ndarray, ref: np.ndarray, what: str = ""):
    assert gpu.shape == ref.shape, f"{what}: shape {gpu.shape} vs {ref.shape}"
    if gpu.size == 0:
        return 1.0
    both_nan = np.isnan(gpu) & np.isnan(ref)
    g = np.where(both_nan, 0, gpu).astype(np.float64)
    r = np.where(both_nan, 0, ref).astype(np.float64)
    with np.errstate(invalid="ignore"):
        same_inf = np.isinf(g) & (g == r)
    g = np.where(same_inf, 0, g)
    r = np.where(same_inf, 0, r)
    tol = np.empty_like(r)
    for f in SCALAR_FIELDS:
        tol[:, f] = RTOL * np.abs(r[:, f]) + ATOL_SCALAR
    for f in VECTOR_FIELDS:
        tol[:, f] = RTOL * np.abs(r[:, f]).max(axis=1, keepdims=True) + ATOL_VECTOR
    with np.errstate(invalid="ignore"):
        bad = ~(np.abs(g - r) <= tol)
    if bad.any():
        i, j = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} floats differ; first at record {i} float {j}: "
                             f"gpu {gpu[i, j]!r} oracle {ref[i, j]!r}\n gpu {gpu[i]}\n ref {ref[i]}")
    return float((gpu.view(np.uint32) == ref.view(np.uint32)).mean())


FIELD_NAMES = (("position", slice(0, 3)), ("color", slice(4, 8)), ("scale", slice(8, 10)), ("normal", slice(12, 15)),
               ("rotation", slice(16, 20)), ("pbr", slice(20, 22)))


def error_report(gpu: np.ndarray, ref: np.ndarray) -> dict:
    """What the parity assertion above does not show: per field, the ACHIEVED errors.
      max_rel_component       max |gpu - ref| / |ref| over components with |ref| >= 1e-3 (a relative error is meaningless on a
                              component that happens to be ~0, e.g. one coordinate of a unit normal)
      max_rel_to_vector       max |gpu - ref| / max|ref vector| (the rule used for position / normal / quaternion)
      max_abs                 max |gpu - ref|
      frac_within_1e-4_component   fraction of floats with |gpu - ref| <= 1e-4 |ref| + eps, strictly per component; eps = 1e-7, and
                              5e-7 (four fp32 ulps of 1) for the components of the UNIT vectors (normal, quaternion): a component of a unit
                              normal that is ~0 carries the rounding error of the whole vector, ~1e-7 absolute, not 1e-4 of itself
      frac_bit_identical
    and a histogram of the per-component relative error (decades)."""
    out = {}
    for name, sl in FIELD_NAMES:
        g, r = gpu[:, sl].astype(np.float64), ref[:, sl].astype(np.float64)
        fin = np.isfinite(g) & np.isfinite(r)
        d = np.where(fin, np.abs(g - r), 0.0)
        big = fin & (np.abs(r) >= 1e-3)
        rel = np.where(big, d / np.maximum(np.abs(r), 1e-30), 0.0)
        vec = np.abs(np.where(fin, r, 0.0)).max(axis=1, keepdims=True)
        relv = d / np.maximum(vec, 1e-30)
        edges = [0.0, 1e-8, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, np.inf]
        hist, _ = np.histogram(rel[big], bins=edges)
        out[name] = {"max_rel_component": float(rel.max()), "max_rel_to_vector": float(relv.max()), "max_abs": float(d.max()),
                     "frac_within_1e-4_component": float((d <= 1e-4 * np.abs(r) + (5e-7 if name in ("normal", "rotation") else 1e-7))[fin].mean()),
                     "frac_bit_identical": float((gpu[:, sl].view(np.uint32) == ref[:, sl].view(np.uint32)).mean()),
                     "rel_error_histogram": {f"<{e:g}": int(h) for e, h in zip(edges[1:], hist)}}
    return out


# What the HIP path ACHIEVES against the oracle on the full-size workloads (max |gpu - ref| per field, measured on the final
# libraries of rounds 2 and 3; gpurun_out/parity_*.json -> profiles/), with a factor 4 of room: the guard of the full-size tests.
# The 1e-4 rule above is what north_star allows; a regression that moved 0.1 % of the floats to 1e-3 would pass it on most
# fields — it does not pass this.  `frac` = required fraction of floats within 1e-4 of THEMSELVES (strictly per component):
# everything (unit-vector components with an absolute 5e-7, see error_report).
# field: (bound on max |gpu - ref|, required fraction within 1e-4 of the component itself)
ACHIEVED = {
    # config 3 (unit sphere at the origin): achieved 6.0e-8 / 3.6e-7 / 1.2e-6 / 4.8e-7 / 1.2e-7 / 3.6e-7 (profiles/r04/parity_c3.json)
    "c3": {"position": (2.4e-7, 1.0), "color": (1.5e-6, 1.0), "scale": (5e-6, 1.0), "normal": (2e-6, 1.0), "rotation": (5e-7, 1.0), "pbr": (1.5e-6, 1.0)},
    # config 4 stand-in (coordinates up to ~3): achieved 2.4e-7 / 3.0e-7 / 1.4e-6 / 4.8e-7 / 1.2e-7 / 3.0e-7 (parity_c4.json)
    "c4": {"position": (1e-6, 1.0), "color": (1.5e-6, 1.0), "scale": (6e-6, 1.0), "normal": (2e-6, 1.0), "rotation": (5e-7, 1.0), "pbr": (1.5e-6, 1.0)},
    # config 5 (coordinates up to ~8, Jacobians of sub-pixel triangles): achieved 9.5e-7 / 2.4e-7 / 3.8e-6 / 3.6e-7 / 1.2e-7 / 2.4e-7 (parity_c5.json)
    # synth.sponza_like (coordinates in [0, 1]; axis-aligned planes and cloth: one or two components of most normals are ~0):
    # achieved 1.2e-7 / 3.0e-7 / 4.8e-7 / 3.9e-7 / 1.2e-7 / 3.0e-7
    "hetero": {"position": (5e-7, 1.0), "color": (1.5e-6, 1.0), "scale": (2e-6, 1.0), "normal": (2e-6, 1.0), "rotation": (5e-7, 1.0), "pbr": (1.5e-6, 1.0)},
    "c5": {"position": (4e-6, 1.0), "color": (1.5e-6, 1.0), "scale": (1.6e-5, 1.0), "normal": (2e-6, 1.0), "rotation": (5e-7, 1.0), "pbr": (1.5e-6, 1.0)},
}


# The same guard for the randomised scenes of tests/test_gpu_fuzz.py (the vector-relative rule of
# assert_records_match is looser than a literal per-component 1e-4; the guard on the ACHIEVED absolute errors is the real one).
# Bounds on max |gpu - ref| per field over a whole fuzz run, 4x what the 200-case run achieves (gpurun_out/parity_fuzz.json:
# "max_abs_error"); scale is relative to the Gaussian's own scale (sub-pixel slivers of the soups have Jacobians of 1e-3 ... 1e3).
# achieved (200 cases, 39.9 M Gaussians, round 6): 2.4e-7 / 3.0e-7 / 4.3e-6 / 6.9e-6 (unnormalised normals of scenes without a normal map) / 1.2e-7 / 3.0e-7
FUZZ_ACHIEVED = {"position": 1e-6, "color": 1.2e-6, "scale_rel": 1.7e-5, "normal": 2.8e-5, "rotation": 5e-7, "pbr": 1.2e-6}


def fuzz_errors(gpu: np.ndarray, ref: np.ndarray) -> dict:
    """max |gpu - ref| per field of one fuzz case (scale: relative to the reference value, finite records only)."""
    out = {}
    for name, sl in FIELD_NAMES:
        g, r = gpu[:, sl].astype(np.float64), ref[:, sl].astype(np.float64)
        fin = np.isfinite(g) & np.isfinite(r)
        d = np.where(fin, np.abs(g - r), 0.0)
        if name == "scale":
            d = d / np.maximum(np.abs(np.where(fin, r, 1.0)), 1e-30)
            name = "scale_rel"
        out[name] = float(d.max()) if d.size else 0.0
    return out


def assert_achieved(gpu: np.ndarray, ref: np.ndarray, what: str, out_json: str = None, config: str = "c3") -> dict:
    """error_report + the guard above (ACHIEVED[config]); writes the report (with the sha of the library under test) to out_json."""
    rep = error_report(gpu, ref)
    rep["_what"] = f"HIP records vs oracle, {what} ({gpu.shape[0]} records); see tests/parity.py:error_report"
    rep["_frac_bit_identical_all_floats"] = float((gpu.view(np.uint32) == ref.view(np.uint32)).mean())
    try:
        import hashlib
        from mesh2splat_amd import _lib
        rep["_library_sha256"] = hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()[:16]
    except Exception:
        pass
    def dump():
        if out_json:
            import json, os
            os.makedirs(os.path.dirname(out_json), exist_ok=True)
            with open(out_json, "w") as fh:
                json.dump(rep, fh, indent=1)
    dump()
    for name, sl in FIELD_NAMES:
        bound, frac = ACHIEVED[config][name]
        mag = 1.0
        v = rep[name]
        v["guard_max_abs"] = bound
        assert v["max_abs"] <= bound * mag, (what, name, v["max_abs"], bound * mag)
        assert v["frac_within_1e-4_component"] >= frac, (what, name, v["frac_within_1e-4_component"])
    dump()
    return rep


def assert_ply_rows_match(mine: np.ndarray, ref: np.ndarray, what: str = ""):
    """Format-1 .ply rows (19 floats: xyz | nxyz | f_dc 3 | metallic roughness | opacity | log-scale 3 | rot 4) written from
    GPU records against the reference's file.  The rows are FUNCTIONS of the records, so the 1e-4 bar on the records becomes:
      x y z, normal, rotation : 1e-4 of the vector (as for the records);
      f_dc = (c - 0.5) / 0.2820948 : |d f_dc| = |dc| / 0.282 <= (1e-4 |c| + 1e-6) / 0.282;
      log-scale = log(s sigma / R) : |d log s| = |ds| / s <= 1e-4 (+ 1e-6 / s: scales are >= 1e-7 by construction);
      opacity = logit(a)          : judged in alpha space (see below)."""
    assert mine.shape == ref.shape, what
    m, r = mine.astype(np.float64), ref.astype(np.float64)
    same_inf = np.isinf(m) & (m == r)
    m = np.where(same_inf, 0, m); r = np.where(same_inf, 0, r)
    tol = np.empty_like(r)
    for sl in (slice(0, 3), slice(3, 6), slice(15, 19)):
        tol[:, sl] = RTOL * np.abs(r[:, sl]).max(axis=1, keepdims=True) + ATOL_VECTOR
    c = r[:, 6:9] * 0.28209479177387814 + 0.5
    tol[:, 6:9] = (RTOL * np.abs(c) + ATOL_SCALAR) / 0.28209479177387814 + 1e-7
    tol[:, 9:11] = RTOL * np.abs(r[:, 9:11]) + ATOL_SCALAR
    # opacity = logit(alpha): its derivative 1 / (a (1 - a)) is unbounded towards a = 1 (an opaque texel is +inf in the
    # reference's file, one ulp below it is 15.9), so a tolerance propagated into logit space would leave near-opaque rows —
    # the common case — unchecked.  The column is therefore judged where the 1e-4 bar is defined: back in alpha
    # space, |sigmoid(mine) - sigmoid(ref)| <= 1e-4 alpha + 1e-6, computed in fp64 (sigmoid(+inf) = 1).
    with np.errstate(over="ignore"):
        a_m = 1.0 / (1.0 + np.exp(-mine[:, 11].astype(np.float64)))
        a_r = 1.0 / (1.0 + np.exp(-ref[:, 11].astype(np.float64)))
    m[:, 11] = a_m
    r[:, 11] = a_r
    tol[:, 11] = RTOL * a_r + ATOL_SCALAR
    tol[:, 12:15] = RTOL + 2e-6
    bad = ~(np.abs(m - r) <= tol)
    if bad.any():
        i, j = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} row floats differ; first at row {i} column {j}: {mine[i, j]!r} vs {ref[i, j]!r}")
    return {"max_abs_log_scale": float(np.abs(m[:, 12:15] - r[:, 12:15]).max()), "max_abs_alpha": float(np.abs(m[:, 11] - r[:, 11]).max()),
            "max_abs_f_dc": float(np.abs(m[:, 6:9] - r[:, 6:9]).max())}
