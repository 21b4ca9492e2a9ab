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


@pytest.mark.gpu
def test_many_shifts_of_a_resident_world_outgrow_slots(svo):
    """A long walk: entering chunks differ in size from the ones they replace, so slots are outgrown, chunks move to the pools'
    tails and, when those are full, the world is packed afresh (the device-only pools are fetched for that).  After every
    leg the world equals a fresh one at the new chunkcoordmin."""
    W = svo.World.generate(3, 1, 2, 128, 7, build_device=0, seed=3)
    ccm = np.zeros(3, int)
    cam_prm = dict(shadow=True)
    for leg, off in enumerate([(1, 0, 0)] * 7 + [(0, 0, 1)] * 6 + [(-1, 0, 0)] * 3 + [(0, 1, 0)]):
        W.shift(off)
        ccm += np.array(off)
        if leg % 4 == 3 or leg == 16:
            F = svo.World.generate(3, 1, 2, 128, 7, chunkcoordmin=tuple(int(v) for v in ccm), build_device=0, seed=3)
            lo = ccm.astype(float) * 128
            o, d = random_rays(np.random.default_rng(leg), 8000, lo, lo + np.array([3, 1, 2]) * 128)
            for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
                assert_gbuffer_equal(W.chunkmarch(o, d, kernel=k, **cam_prm), F.chunkmarch(o, d, kernel=k, **cam_prm), f"leg {leg}/kernel {k}")
            if leg == 16:
                worlds_equal(W, F, 6)
            F.destroy()
    W.destroy()


@pytest.mark.gpu
def test_shift_of_the_benchmark_world_is_fast(svo):
    """C3's world (4x1x4 chunks, depth 12) slides one chunk: four chunks of 6 M nodes + 2 M bricks each are generated where the
    pools live.  (The host path took seconds; the measured time is printed.)"""
    import time
    W = svo.World.generate(4, 1, 4, 128, 12, build_device=0)
    t0 = time.time(); W.shift((1, 0, 0)); t1 = time.time(); W.shift((0, 0, -1)); t2 = time.time()
    print(f"\nC3 world shift: {t1 - t0:.3f} s, {t2 - t1:.3f} s")
    F = svo.World.generate(4, 1, 4, 128, 12, chunkcoordmin=(1, 0, -1), build_device=0)
    assert W.info.total_trees == F.info.total_trees and W.info.total_twigs == F.info.total_twigs
    for i in (0, 5, 15):
        a, b = W.chunk(i, copy=False), F.chunk(i, copy=False)
        assert a["position"] == b["position"]
        assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["twig"], b["twig"])
    # (timings are printed, not asserted: see tests/test_gpu_edits.py)
    W.destroy(); F.destroy()
