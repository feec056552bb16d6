"""The 313-bin CIELab ab gamut table used by DISCO's colour classification.

Reference: utils/cielab.py:5-64 (ABGamut loads utils/gamut_pts.npy; CIELAB.q_to_ab
turns it into bin centres) and models/basic.py:150-153 (ColorLabel.q_to_ab).
[probe] q_to_ab is value- and order-identical to gamut_pts.npy.astype(float32).

The table is data, not code: the in-gamut ab grid points (multiples of 10) of
Zhang et al. 2016.  The points are sorted by a then b and every a-row is a
contiguous run of b values with step 10, so the whole table is the 20 runs
below (a, b_min, b_max).  tests/test_oracle_golden.py pins the expansion against
the golden copy captured from the reference (tests/golden/gamut.npz).
"""
import numpy as np

N_BINS = 313

# (a, b_min, b_max) — inclusive, step 10
_RUNS = (
    (-90, 50, 90), (-80, 20, 90), (-70, 0, 90), (-60, -20, 90), (-50, -30, 100),
    (-40, -40, 100), (-30, -50, 100), (-20, -50, 100), (-10, -60, 100), (0, -70, 100),
    (10, -80, 90), (20, -80, 90), (30, -90, 90), (40, -100, 90), (50, -100, 80),
    (60, -110, 80), (70, -110, 80), (80, -110, 70), (90, -110, 70), (100, -90, 0),
)


def gamut_points() -> np.ndarray:
    """(313, 2) float32 bin centres in ab units (a, b), reference bin order."""
    pts = [(a, b) for a, lo, hi in _RUNS for b in range(lo, hi + 1, 10)]
    out = np.asarray(pts, dtype=np.float32)
    assert out.shape == (N_BINS, 2)
    return out
