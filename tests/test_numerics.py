"""Float identities the HIP kernels rely on to stay bit-exact with src/Traverse.cpp (CPU only, numpy)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rng = np.random.default_rng(42)


def random_floats(n):
    bits = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
    f = bits.view(np.float32)
    return f[np.isfinite(f) & (f != 0)]


def test_double_reciprocal_rounded_to_float_equals_float_reciprocal():
    """cubeEscapeDistance computes vec3(1.0 / b.x, ...) in double (src/Traverse.cpp:27); the kernels hoist 1.0f / b."""
    b = np.concatenate([random_floats(2_000_000), rng.normal(size=1_000_000).astype(np.float32)])
    b = b[b != 0]
    with np.errstate(over="ignore", divide="ignore"):
        via_double = (np.float64(1.0) / b.astype(np.float64)).astype(np.float32)
        direct = np.float32(1.0) / b
    assert np.array_equal(via_double.view(np.uint32), direct.view(np.uint32))


def test_pow2_reciprocal_bit_trick():
    """recip_pow2(x) = bits(0x7F000000 - bits(x)) is exactly 1/x for normal powers of two."""
    for e in range(-100, 101):
        x = np.float32(2.0) ** np.float32(e)
        r = (np.uint32(0x7F000000) - x.view(np.uint32)).view(np.float32)
        assert r == np.float32(1.0) / x
    # and division by a power of two equals multiplication by that reciprocal
    a = random_floats(200_000)
    a = a[(np.abs(a) > 1e-20) & (np.abs(a) < 1e20)]
    for e in (-10, -5, -3, 0, 2, 7):
        x = np.float32(2.0) ** np.float32(e)
        assert np.array_equal((a / x).view(np.uint32), (a * (np.float32(1) / x)).view(np.uint32))


def test_inverse_sqrt_constants_in_kernel_source():
    src = open(os.path.join(ROOT, "octree-raymarcher_amd", "csrc", "kernel_stack.hip.h")).read()
    consts = re.findall(r"0x([0-9A-Fa-f]{8})u\);\s*// 1\.0f / sqrtf\((\d)\.0f\)", src)
    assert len(consts) == 2
    for hexbits, n in consts:
        want = (np.float32(1.0) / np.sqrt(np.float32(int(n)))).view(np.uint32)
        assert int(hexbits, 16) == int(want)


def test_integer_cell_coordinates_reproduce_float_descent():
    """The stack kernel replaces the reference's per-level `p >= mid` float tests (src/Traverse.cpp:39-45) by integer
    cell coordinates u = number of cell boundaries <= p, computed with one truncation and one compare-and-fix.
    Check against the literal float descent on adversarial points (on / next to lattice planes)."""
    for pos, size, levels in ((0.0, 128.0, 10), (-256.0, 128.0, 8), (384.0, 128.0, 6), (128.0, 128.0, 14)):
        cell = np.float32(size / 2**levels)
        n = 200_000
        k = rng.integers(0, 2**levels + 1, n)
        base = (np.float32(pos) + k.astype(np.float32) * cell).astype(np.float32)
        jitter = rng.integers(-3, 4, n)
        p = base.copy()
        for _ in range(3):
            up = jitter > 0
            dn = jitter < 0
            p[up] = np.nextafter(p[up], np.float32(np.inf)); jitter[up] -= 1
            p[dn] = np.nextafter(p[dn], np.float32(-np.inf)); jitter[dn] += 1
        p = np.concatenate([p, (np.float32(pos) + rng.random(n).astype(np.float32) * np.float32(size)).astype(np.float32)])
        p = p[(p >= np.float32(pos)) & (p <= np.float32(pos + size))]
        # literal descent: full depth, recording the path bits
        lo = np.full(p.shape, np.float32(pos)); s = np.float32(size); ref = np.zeros(p.shape, np.int64)
        for _ in range(levels):
            half = np.float32(s * np.float32(0.5))
            mid = (lo + half).astype(np.float32)
            ge = p >= mid
            lo = (lo + ge.astype(np.float32) * half).astype(np.float32)
            ref = ref * 2 + ge
            s = half
        # kernel formula
        inv = np.float32(1.0) / cell
        u = ((p - np.float32(pos)).astype(np.float32) * inv).astype(np.float32).astype(np.int64)
        u = np.minimum(u, 2**levels - 1)
        u -= ((np.float32(pos) + u.astype(np.float32) * cell).astype(np.float32) > p)
        assert np.array_equal(u, ref)
        # kernel v4 form: plain truncation, compare-and-fix only where the quotient is integral
        f = ((p - np.float32(pos)).astype(np.float32) * inv).astype(np.float32)
        u2 = f.astype(np.int64)
        integral = f == u2.astype(np.float32)
        uc = np.minimum(u2, 2**levels - 1)
        uc = uc - ((np.float32(pos) + uc.astype(np.float32) * cell).astype(np.float32) > p)
        u2 = np.where(integral, uc, u2)
        assert np.array_equal(u2, ref) and integral.sum() > 100
        # the node box rebuilt from the coordinates equals the incrementally accumulated bmin
        assert np.array_equal((np.float32(pos) + u.astype(np.float32) * cell).astype(np.float32), lo)
