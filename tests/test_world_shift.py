"""World::shift (src/World.cpp:334-378) as svo_world_shift: after sliding the grid, every chunk slot holds exactly what a
fresh World::init at the new chunkcoordmin puts there (toroidal indexing), and the march agrees with the oracle."""
import numpy as np
import pytest

from helpers import assert_gbuffer_equal, random_rays


def worlds_equal(a, b, n):
    ia, ib = a.info, b.info
    assert tuple(ia.chunkcoordmin) == tuple(ib.chunkcoordmin)
    for i in range(n):
        ca, cb = a.chunk(i), b.chunk(i)
        assert ca["position"] == cb["position"], i
        assert np.array_equal(ca["tree"], cb["tree"]) and np.array_equal(ca["twig"], cb["twig"]), i


@pytest.mark.parametrize("path", [[(1, 0, 0)], [(0, 0, -1)], [(0, 1, 0)], [(1, 0, 0), (1, 0, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0)]])
def test_shift_equals_fresh_world(svo, path):
    W = svo.World.generate(3, 2, 2, 128, 4, chunkcoordmin=(0, 0, 0))
    ccm = np.zeros(3, int)
    for off in path:
        W.shift(off)
        ccm += np.array(off)
    F = svo.World.generate(3, 2, 2, 128, 4, chunkcoordmin=tuple(int(v) for v in ccm))
    worlds_equal(W, F, 12)


def test_shift_rejects_bad_offsets_and_created_worlds(svo):
    W = svo.World.generate(2, 1, 2, 128, 3)
    for bad in [(0, 0, 0), (1, 1, 0), (2, 0, 0)]:
        with pytest.raises(svo.SvoError):
            W.shift(bad)
    C = svo.World.create([W.chunk(i) for i in range(4)], 2, 1, 2, 128)
    with pytest.raises(svo.SvoError) as e:
        C.shift((1, 0, 0))
    assert e.value.code == -6


@pytest.mark.gpu
def test_march_after_shift_matches_oracle(svo, oracle):
    W = svo.World.generate(3, 1, 3, 128, 6)
    W.upload(0)
    for off in [(1, 0, 0), (0, 0, -1), (1, 0, 0)]:
        W.shift(off)
    ccm = tuple(W.info.chunkcoordmin)
    assert ccm == (2, 0, -1)
    O = oracle.OracleWorld.generate(3, 1, 3, 128, 6, chunkcoordmin=ccm)
    lo = np.array(ccm, float) * 128
    hi = lo + np.array([3, 1, 3]) * 128
    o, d = random_rays(np.random.default_rng(8), 20000, lo, hi)
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"shifted/{k}")
    assert (want["flags"] & 1).sum() > 2000


@pytest.mark.gpu
def test_shift_of_a_device_resident_world(svo, oracle):
    """svo_world_shift on a world whose pools were built on the device and never copied to the host: the slid-in chunks
    replace their slots, the others keep their device-only bricks; march and (lazily fetched) pools match a fresh world."""
    W = svo.World.generate(3, 1, 3, 128, 6, build_device=0)
    for off in [(1, 0, 0), (0, 0, -1), (-1, 0, 0), (-1, 0, 0)]:
        W.shift(off)
    ccm = tuple(W.info.chunkcoordmin)
    assert ccm == (-1, 0, -1)
    O = oracle.OracleWorld.generate(3, 1, 3, 128, 6, chunkcoordmin=ccm)
    lo = np.array(ccm, float) * 128
    hi = lo + np.array([3, 1, 3]) * 128
    o, d = random_rays(np.random.default_rng(9), 20000, lo, hi)
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"shifted resident/{k}")
    F = svo.World.generate(3, 1, 3, 128, 6, chunkcoordmin=ccm)
    worlds_equal(W, F, 9)
