"""BASELINE.json configs at full size on the GPU, checked against the oracle on the whole image and
through size-independent properties (kernel agreement, flag implications, band partition == whole frame)."""
import os

import numpy as np
import pytest

from helpers import assert_gbuffer_equal, chunks_of

pytestmark = pytest.mark.gpu
THREADS = os.cpu_count() or 8


def check_properties(svo, g, shadow):
    hit = (g["flags"] & svo.HIT_FLAG) != 0
    assert not np.any(g["flags"] & svo.ERR_FLAG)
    assert np.all(g["t"][~hit] == 0) and np.all(g["material"][~hit] == 0)
    assert np.all(g["material"][hit] > 0)
    traced = (g["flags"] & svo.SHADOW_TRACED) != 0
    shadowed = (g["flags"] & svo.SHADOWED) != 0
    assert np.array_equal(traced, hit if shadow else np.zeros_like(hit))
    assert not np.any(shadowed & ~traced)
    brick = hit & (g["cell"] != svo.CELL_NONE)
    assert np.all(g["cell"][brick] < 64) and np.all(g["cell"][hit & ~brick] == svo.CELL_NONE)
    n = g["normal"][hit]
    ok = ~np.isnan(n).any(axis=1)
    assert np.allclose(np.linalg.norm(n[ok], axis=1), 1.0, atol=1e-6)


def test_c2_1080p_depth10_single_chunk(svo, oracle):
    """configs[1]: 1920x1080 primary rays, depth-10 SVO."""
    W = svo.World.generate(1, 1, 1, 128, 10)
    O = oracle.OracleWorld.from_chunks([W.chunk(0, copy=False)], 1, 1, 1, 128)
    W.upload(0)
    cam = svo.default_camera(1, 1, 128, 1920, 1080)
    want = O.trace_image(cam, threads=THREADS)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        got = W.draw(cam, kernel=k)
        assert_gbuffer_equal(got, want, f"c2/{k}")
    check_properties(svo, got, False)
    assert (want["flags"] & 1).mean() > 0.1
    W.destroy()


@pytest.fixture(scope="module")
def c3(svo):
    W = svo.World.generate(4, 1, 4, 128, 12)
    W.upload(0)
    yield W
    W.destroy()


def test_c3_1080p_depth12_4x1x4_shadow(svo, oracle, c3):
    """configs[2] — the benchmark workload: primary + 1 shadow ray per hit, depth-12 multi-chunk world."""
    O = oracle.OracleWorld.from_chunks([c3.chunk(i, copy=False) for i in range(16)], 4, 1, 4, 128)
    cam = svo.default_camera(4, 4, 128, 1920, 1080)
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=THREADS)
    got = c3.draw(cam, shadow=True, kernel=svo.KERNEL_STACK)
    assert_gbuffer_equal(got, want, "c3/stack")
    assert c3.last_ray_count() == O.last_rays
    check_properties(svo, got, True)
    lit = c3.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL)
    assert_gbuffer_equal(lit, want, "c3/literal")
    assert 0.3 < (want["flags"] & 1).mean() < 0.9 and ((want["flags"] & 4) != 0).sum() > 10000


def test_c4_2160p_band_partition_of_8(svo, c3):
    """configs[3]: the 3840x2160 image as 8 ranks would trace it (8-row bands round-robin) equals the whole frame;
    every rank's bands are traced here on the one GPU and de-interleaved with the bench's helper."""
    cam = svo.default_camera(4, 4, 128, 3840, 2160)
    full = c3.draw(cam, shadow=True)
    part = svo.partition
    n = 8
    nb = part.bands_per_rank(2160, n)
    prm = svo.trace_params(shadow=True)
    gathered = []
    buf = svo.DeviceBuffer(nb * part.BAND * 3840 * 32)
    for r in range(n):
        c3.trace_rows(cam, prm, r, n, nb, part.BAND, buf.ptr)
        svo.lib.svo_stream_synchronize(None)
        gathered.append(buf.to_numpy(svo.HIT_DTYPE, nb * part.BAND * 3840).reshape(nb, part.BAND, 3840))
    frame = part.deinterleave(gathered, 2160)
    assert_gbuffer_equal(frame, full, "c4/bands")
    check_properties(svo, full, True)


def test_c5_depth16_sparse(svo, oracle):
    """configs[4]: depth-16 sparse SVO (full depth only inside a 1-unit band, bricks at depth 10 elsewhere): 14 branch
    levels exercise the deepest LDS-stack instantiation; whole 1080p image against the oracle."""
    W = svo.World.generate(1, 1, 1, 128, 16, pyramid_resolution=4096, water=False, coarse_depth=10,
                           refine_box=((63.5, -1e9, -1e9), (64.5, 1e9, 1e9)))
    assert W.info.max_chunk_depth == 16 and W.info.exact_geometry == 1
    O = oracle.OracleWorld.from_chunks([W.chunk(0, copy=False)], 1, 1, 1, 128)
    W.upload(0)
    cam = svo.make_camera((64.2, 150.0, -40.0), (0.0, -0.5, 0.866), (0.0, 1.0, 0.0), 60.0, 1920, 1080)
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=THREADS)
    got = W.draw(cam, shadow=True, kernel=svo.KERNEL_STACK)
    assert_gbuffer_equal(got, want, "c5/stack")
    check_properties(svo, got, True)
    # the refined band is actually hit: some hit voxels are depth-16 voxels (nodes at level 14 -> brick cells of 128/65536)
    hit = (want["flags"] & 1) != 0
    assert hit.mean() > 0.1
    W.destroy()
