"""Shared helpers for the parity tests (test infrastructure)."""
from __future__ import annotations

import numpy as np

# Float tolerance of the contract (SURVEY.md App. D rule 4).  The kernels are built to reproduce the
# reference's float ops exactly, so the tests additionally require bit equality of t.
T_RTOL = 1e-4
N_ATOL = 1e-6


def chunks_of(world, n):
    return [world.chunk(i) for i in range(n)]


def assert_gbuffer_equal(got, want, what=""):
    """got/want: HIT_DTYPE arrays.  Integer fields bit-exact; t and normal within tolerance AND bit-exact."""
    got = np.asarray(got).reshape(-1)
    want = np.asarray(want).reshape(-1)
    assert got.shape == want.shape, what
    for f in ("flags", "material", "chunk", "node", "cell"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, f"{what}: field {f} differs at {bad[:8]} (of {bad.size}): got {got[f][bad[:8]]} want {want[f][bad[:8]]}"
    hit = (want["flags"] & 1) != 0
    tol = T_RTOL * np.maximum(1.0, np.abs(want["t"][hit]))
    assert np.all(np.abs(got["t"][hit] - want["t"][hit]) <= tol), f"{what}: t outside tolerance"
    assert np.array_equal(got["t"].view(np.uint32), want["t"].view(np.uint32)), f"{what}: t not bit-identical"
    gn, wn = got["normal"][hit], want["normal"][hit]
    both_nan = np.isnan(gn) & np.isnan(wn)
    assert np.all(both_nan | (np.abs(gn - wn) <= N_ATOL)), f"{what}: normal outside tolerance"


def random_rays(rng, n, lo, hi, inside_frac=0.5):
    """Origins in/around the box [lo,hi], random directions (normalised in float64 then cast)."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    ext = hi - lo
    o = lo + rng.random((n, 3)) * ext
    outside = rng.random(n) > inside_frac
    o[outside] = lo - 0.5 * ext + rng.random((int(outside.sum()), 3)) * 2.0 * ext
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)
