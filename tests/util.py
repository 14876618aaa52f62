"""Shared helpers for the parity tests.

Tolerance contract (BASELINE.json north_star): fp32 results match the reference CPU kernels within
1e-4 relative -- measured against the tensor's own scale, max|ref| (an element-wise relative error is
meaningless where cancellation leaves an output near 0) -- and shape / index ops are bit-exact.
"""
import numpy as np

REL_TOL = 1e-4


def rel_err(got: np.ndarray, ref: np.ndarray) -> float:
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    scale = max(float(np.abs(ref).max()), 1e-30)
    return float(np.abs(got - ref).max()) / scale


def assert_parity(got, ref, rel=REL_TOL, what=""):
    assert np.isfinite(np.asarray(got)).all(), "%s: non-finite values" % what
    e = rel_err(got, ref)
    assert e <= rel, "%s: max|diff|/max|ref| = %.3e > %.1e" % (what, e, rel)
    return e


def assert_exact(got, ref, what=""):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.array_equal(got, ref), "%s: %d of %d elements differ" % (what, int((got != ref).sum()), ref.size)


def rng_uniform(seed, shape, lo=0.0, hi=1.0):
    """Deterministic U[lo,hi) float32 (the reference tests use Eigen setRandom(): U[0,1), unseeded)."""
    r = np.random.Generator(np.random.Philox(seed))
    return (lo + (hi - lo) * r.random(shape, dtype=np.float32)).astype(np.float32)
