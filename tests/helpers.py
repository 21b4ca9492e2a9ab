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


def adversarial_rays(rng, n, lo, hi):
    """Ray lists built to land on the march's corner cases: origins on the voxel lattice / on chunk faces / outside the
    world, direction components of 1e-4 (creeping rays) and exactly 0, axis-parallel rays, directions within 1e-7 of the
    default shadow diagonal."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    ext = hi - lo
    o = lo + rng.random((n, 3)) * ext
    k = n // 8
    o[:k] = np.round(o[:k] * 4) / 4
    o[k:2 * k] = lo + np.round(rng.random((k, 3)) * (ext / 128)) * 128
    o[2 * k:3 * k] = lo - 0.3 * ext + rng.random((k, 3)) * 1.6 * ext
    d = rng.normal(size=(n, 3))
    d[3 * k:4 * k, rng.integers(0, 3)] *= 1e-4
    d[4 * k:5 * k, rng.integers(0, 3)] = 0.0
    z = 5 * k + np.arange(k)
    ax = rng.integers(0, 3, k)
    d[z] = 0.0
    d[z, ax] = rng.choice([-1.0, 1.0], k)                                # axis-parallel
    d[6 * k:7 * k] = np.sign(d[6 * k:7 * k]) * np.array([1.0, 1.0, 0.0]) + 1e-7 * rng.normal(size=(k, 3))
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 0, d / np.where(nrm > 0, nrm, 1), d)
    return o.astype(np.float32), d.astype(np.float32)


def guard_boundary_rays(rng, n, lo, hi, edge, eps=1.0 / 8192):
    """Rays around the conditions of the stack kernel's sure-miss brick test (step_asm_body.inc): one direction component at
    2^-4 .. 2^4 times the guard's threshold edge / (EPS 2^22), exactly 0, a float32 denormal, 1e-30; a third of the origins on or
    within an ulp-sized step of a brick lattice plane (multiples of `edge`, the planes through 0 of a world with negative
    coordinates among them), where p - bmin is inexact and a ray sits within EPS of the box it is about to leave."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    o = lo + rng.random((n, 3)) * (hi - lo)
    k = n // 3
    ax = rng.integers(0, 3, k)
    plane = np.round(o[np.arange(k), ax] / edge) * edge
    o[np.arange(k), ax] = plane + rng.choice([0.0, 1e-7, -1e-7, 3e-5, -3e-5, 1e-4, -1e-4], k) * rng.choice([1.0, edge], k)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c = edge / (eps * 2.0 ** 22)
    small = np.concatenate([c * 2.0 ** rng.integers(-4, 5, n - 3 * (n // 8)), np.zeros(n // 8), np.full(n // 8, 1e-39), np.full(n // 8, 1e-30)])
    rng.shuffle(small)
    d[np.arange(n), rng.integers(0, 3, n)] = small * rng.choice([-1.0, 1.0], n)
    return o.astype(np.float32), d.astype(np.float32)


FUZZ_CASES = [dict(w=2, h=1, d=2, depth=8, ccm=(0, 0, 0)), dict(w=2, h=2, d=2, depth=6, ccm=(-1, -1, -1)),
              dict(w=1, h=1, d=1, depth=11, ccm=(3, 0, -2)), dict(w=3, h=1, d=1, depth=7, ccm=(-2, 0, 5))]


def creeping_rays(rng, n, lo, hi, voxel, chunk_faces=False):
    """Rays that sit exactly on a lattice plane (voxel, brick or node face; chunk faces on request) and move towards its
    negative side by less than an ulp per step: the reference then advances by EPS alone, for hundreds or thousands of
    steps (src/Traverse.cpp:25-32 has no guard).  One, two or three pinned axes; the other components are ordinary.
    A chunk face is a lattice plane of every level at once: a ray pinned there creeps in chunkmarch, treemarch and
    twigmarch together (up to cap_chunk * cap_tree * cap_twig steps), so callers pass small caps with chunk_faces."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    o = lo + rng.random((n, 3)) * (hi - lo)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    kmax = max(1, int(np.log2(32.0 / voxel)) + 1)
    pitch = voxel * (2.0 ** rng.integers(0, kmax, n))                    # voxel, brick, coarser node faces (<= 32)
    for j in range(n):
        for a in rng.permutation(3)[: rng.integers(1, 4)]:
            o[j, a] = np.round(o[j, a] / pitch[j]) * pitch[j]
            if not chunk_faces and o[j, a] % 128.0 == 0.0:
                o[j, a] += pitch[j]
            d[j, a] = -abs(d[j, a]) * 10.0 ** -rng.integers(3, 7)        # -1e-3 .. -1e-6 of an ordinary component
    if chunk_faces:
        k = n // 8
        ax = rng.integers(0, 3, k)
        o[np.arange(k), ax] = np.round(o[np.arange(k), ax] / 128.0) * 128.0
        d[np.arange(k), ax] = -np.abs(d[np.arange(k), ax]) * 1e-4
    d /= np.linalg.norm(d, axis=1, keepdims=True)                        # unit directions: the ray does move
    # a quarter start off the plane and drift onto lattice-valued floats on the way (the off-lattice camera's case)
    q = n // 4
    o[-q:] += (rng.random((q, 3)) - 0.5) * 0.7
    return o.astype(np.float32), d.astype(np.float32)
