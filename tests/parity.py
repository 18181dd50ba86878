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
