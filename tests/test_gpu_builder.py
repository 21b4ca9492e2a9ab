"""Device-side world generation (csrc/builder.hip): noise, min/max mips, grow() and the water fill
(Ocroot::build) as kernels, pools left on the device (the world is uploaded when svo_world_generate returns).
The pools must be bit-identical to the oracle's restatement of grow / BoundsPyramid / Ocroot::build
(OracleWorld.generate) and to the host builder's; the march over a device-built world - whose node words and bricks
never visited the host - must equal the oracle's march over the oracle's world."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = [
    dict(w=1, h=1, d=1, depth=2),
    dict(w=1, h=1, d=1, depth=6),
    dict(w=2, h=2, d=2, depth=5, chunkcoordmin=(-1, -1, -1)),
    dict(w=3, h=1, d=2, depth=7, seed=77),
    dict(w=1, h=1, d=1, depth=8, pyramid_resolution=64),                    # bilinear path beyond the pyramid base
    dict(w=1, h=1, d=1, depth=9, water=False),
    dict(w=1, h=1, d=1, depth=10, amplitude=30.0, yshift=50.0, water_level=40.0),
    dict(w=1, h=1, d=1, depth=12, water=False, coarse_depth=8, refine_box=((60, -1e9, -1e9), (68, 1e9, 1e9))),
    # the water fill (Ocroot::build) on the device: planes on and off the node lattice, above and below all terrain,
    # in a chunk whose bricks sit at two levels (the fill splits the coarse EMPTY nodes it cuts down to depth-2)
    dict(w=1, h=1, d=1, depth=8, water_level=8.0),
    dict(w=1, h=1, d=1, depth=8, water_level=16.0, amplitude=20.0),
    dict(w=1, h=1, d=1, depth=7, water_level=33.37),
    dict(w=1, h=2, d=1, depth=6, water_level=128.0),
    dict(w=1, h=2, d=1, depth=6, water_level=300.0),
    dict(w=1, h=1, d=1, depth=6, water_level=-100.0),
    dict(w=2, h=1, d=1, depth=9, water_level=11.3, coarse_depth=6, refine_box=((100, -1e9, -1e9), (140, 1e9, 1e9))),
    dict(w=1, h=1, d=1, depth=11, water_level=6.0, chunkcoordmin=(5, 0, -3)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items() if k != "refine_box"))
def test_device_built_world_equals_host_built(svo, case):
    c = dict(case)
    w, h, d, depth = c.pop("w"), c.pop("h"), c.pop("d"), c.pop("depth")
    H = svo.World.generate(w, h, d, 128, depth, **c)
    D = svo.World.generate(w, h, d, 128, depth, build_device=0, **c)
    for i in range(w * h * d):
        a, b = H.chunk(i, copy=False), D.chunk(i, copy=False)
        assert a["position"] == b["position"] and a["depth"] == b["depth"]
        assert np.array_equal(a["tree"], b["tree"]), f"chunk {i}: node words differ"
        assert np.array_equal(a["twig"], b["twig"]), f"chunk {i}: bricks differ"
    # Ocroot's storage sizes (treestoragesize / twigstoragesize doubling) size the pool slots: the same either way
    H.upload(0)
    assert H.info.tree_pool_bytes == D.info.tree_pool_bytes and H.info.twig_pool_bytes == D.info.twig_pool_bytes
    H.destroy(); D.destroy()


ORACLE_CASES = [
    dict(w=1, h=1, d=1, depth=2),
    dict(w=2, h=1, d=2, depth=7),
    dict(w=2, h=2, d=2, depth=5, chunkcoordmin=(-1, -1, -1)),
    dict(w=1, h=1, d=1, depth=10),
    dict(w=1, h=1, d=1, depth=9, water=False),
]


@pytest.mark.parametrize("case", ORACLE_CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_device_built_world_equals_oracle_generate(svo, oracle, case):
    """Pools against OracleWorld.generate (the oracle's own World::init), then the march over the device-resident world
    BEFORE any host copy of the bricks exists against the oracle's march over its own world."""
    c = dict(case)
    w, h, d, depth = c.pop("w"), c.pop("h"), c.pop("d"), c.pop("depth")
    ccm = c.get("chunkcoordmin", (0, 0, 0))
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm, water=c.get("water", True))
    D = svo.World.generate(w, h, d, 128, depth, build_device=0, **c)
    assert D.info.uploaded_device == 0                              # resident: no svo_world_upload needed
    cam = svo.make_camera((ccm[0] * 128 + w * 64.0 + 0.3, ccm[1] * 128 + 150.0, ccm[2] * 128 - 40.0), (0.0, -0.5, 0.866), (0.0, 1.0, 0.0), 60.0, 320, 180)
    from helpers import assert_gbuffer_equal
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(D.draw(cam, shadow=True, kernel=k), want, f"march over the device-resident world / kernel {k}")
    D.upload(0)                                                     # a no-op on the device it was built on
    assert_gbuffer_equal(D.draw(cam, shadow=True), want, "after upload()")
    for i in range(w * h * d):
        a, b = O.chunk(i), D.chunk(i, copy=False)                   # first request: the bricks are fetched from HBM
        assert tuple(a["position"]) == tuple(b["position"]) and a["depth"] == b["depth"]
        assert np.array_equal(a["tree"], b["tree"]), f"chunk {i}: node words differ from the oracle's"
        assert np.array_equal(a["twig"], b["twig"]), f"chunk {i}: bricks differ from the oracle's"
    assert_gbuffer_equal(D.draw(cam, shadow=True), want, "after the host copy was made")
    D.destroy()


def test_device_builder_c3_world_and_speed(svo):
    """The benchmark world (4x1x4 chunks, depth 12): identical pools; report both generation times."""
    t0 = time.time(); H = svo.World.generate(4, 1, 4, 128, 12); th = time.time() - t0
    t0 = time.time(); D = svo.World.generate(4, 1, 4, 128, 12, build_device=0); td = time.time() - t0
    assert H.info.total_trees == D.info.total_trees and H.info.total_twigs == D.info.total_twigs
    for i in range(16):
        a, b = H.chunk(i, copy=False), D.chunk(i, copy=False)
        assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["twig"], b["twig"])
    print(f"\nC3 world generation: host threads {th:.2f} s (+ upload), device builder {td:.2f} s (resident: generate + upload)")
    H.destroy(); D.destroy()


@pytest.mark.parametrize("chunksize,depth,ccm", [(32, 6, (0, 0, 0)), (512, 7, (-1, 0, -2)), (2, 5, (3, -1, 0))])
def test_device_builder_other_chunk_edges(svo, oracle, chunksize, depth, ccm):
    """Chunk edges other than the reference's 128: the device builder, the host builder and the oracle's World::init
    produce the same pools."""
    H = svo.World.generate(2, 1, 2, chunksize, depth, chunkcoordmin=ccm)
    D = svo.World.generate(2, 1, 2, chunksize, depth, chunkcoordmin=ccm, build_device=0)
    O = oracle.OracleWorld.generate(2, 1, 2, chunksize, depth, chunkcoordmin=ccm)
    for i in range(4):
        a, b, c = H.chunk(i, copy=False), D.chunk(i, copy=False), O.chunk(i)
        assert a["position"] == b["position"] == tuple(c["position"]) and a["depth"] == b["depth"] == c["depth"]
        assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["tree"], c["tree"]), f"chunk {i}: node words differ"
        assert np.array_equal(a["twig"], b["twig"]) and np.array_equal(a["twig"], c["twig"]), f"chunk {i}: bricks differ"
    H.destroy(); D.destroy()


def test_update_of_a_device_resident_world(svo, oracle):
    """A world built on the device (bricks never copied to the host) takes svo_world_update like any other: the edited
    chunk's pools come from the caller, the other chunk's bricks stay where the builder left them; then a host copy of
    the untouched chunk is still what the oracle generated."""
    import ctypes as C
    from helpers import assert_gbuffer_equal, random_rays
    O = oracle.OracleWorld.generate(2, 1, 1, 128, 7)
    D = svo.World.generate(2, 1, 1, 128, 7, build_device=0)
    rng = np.random.default_rng(5)
    o, d = random_rays(rng, 20000, (0, 0, 0), (256, 128, 128))
    prm = oracle.make_params(shadow=True)
    assert_gbuffer_equal(D.chunkmarch(o, d, shadow=True), O.trace_rays(o, d, params=prm, threads=8), "before")
    for kind, lo, hi, mat in (("build", (20, 60, 20), (70, 110, 50), 5), ("destroy", (0, 0, 0), (128, 45, 30), 0)):
        dt, dw = oracle.Delta(), oracle.Delta()
        root = C.byref(O.w.chunk[0])
        if kind == "build":
            oracle.lib.orc_build(root, oracle.vec3(lo), oracle.vec3(hi), mat, C.byref(dt), C.byref(dw))
        else:
            oracle.lib.orc_destroy(root, oracle.vec3(lo), oracle.vec3(hi), C.byref(dt), C.byref(dw))
        c = O.chunk(0)
        D.update(0, c, tree_range=(min(dt.left, c["tree"].size), dt.right), twig_range=(min(dw.left, c["twig"].size // 64), dw.right),
                 realloc=bool(dt.realloc_ or dw.realloc_))
        want = O.trace_rays(o, d, params=prm, threads=8)
        for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
            assert_gbuffer_equal(D.chunkmarch(o, d, shadow=True, kernel=k), want, f"after {kind}/kernel {k}")
    a, b = O.chunk(1), D.chunk(1, copy=False)                       # fetched from HBM now
    assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["twig"], b["twig"])
    a, b = O.chunk(0), D.chunk(0, copy=False)
    assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["twig"], b["twig"])
    D.destroy()
