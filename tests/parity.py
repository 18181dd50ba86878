"""Shared comparison helpers for the parity tests (GPU records vs CPU oracle records)."""
import numpy as np

# north_star tolerance: 1e-4 relative on scale / rotation / position / colour / opacity.
RTOL = 1e-4
# absolute floor for components that are mathematically ~0 (e.g. z of a planar mesh, a quaternion
# component of an axis-aligned frame): 1e-6 of the field's natural magnitude (positions are O(1)).
ATOL = 1e-6


def assert_records_match(gpu: np.ndarray, ref: np.ndarray, what: str = ""):
    assert gpu.shape == ref.shape, f"{what}: shape {gpu.shape} vs {ref.shape}"
    if gpu.size == 0:
        return 1.0
    both_nan = np.isnan(gpu) & np.isnan(ref)
    g = np.where(both_nan, 0, gpu)
    r = np.where(both_nan, 0, ref)
    bad = ~np.isclose(g, r, rtol=RTOL, atol=ATOL)
    if bad.any():
        i, j = np.argwhere(bad)[0]
        raise AssertionError(f"{what}: {bad.sum()} of {bad.size} floats differ; first at record {i} float {j}: "
                             f"gpu {gpu[i, j]!r} oracle {ref[i, j]!r}\n gpu {gpu[i]}\n ref {ref[i]}")
    return float((gpu.view(np.uint32) == ref.view(np.uint32)).mean())
