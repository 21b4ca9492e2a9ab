"""svo_trace_params.semantics = SVO_SEMANTICS_GLSL: the march of shaders/Chunkmarch.glsl - what the reference renders with
(EPS 1/4096, caps 256 / 512 / 64, `d < EPS ? BIGEPS : d`, tnear > 0 at the world entry, no chunk containment
re-check, LEAF hits at t) - by both HIP kernels against the oracle's restatement of the same twin (itself restated twice:
tests/test_oracle_cross_check.py).  Same bar as the CPU march: integer fields and t bit for bit."""
import numpy as np
import pytest

from helpers import adversarial_rays, assert_gbuffer_equal, creeping_rays, random_rays

pytestmark = pytest.mark.gpu
KERNELS = ["stack", "literal"]


def kid(svo, name):
    return {"stack": svo.KERNEL_STACK, "literal": svo.KERNEL_LITERAL}[name]


@pytest.fixture(scope="module")
def world(svo, oracle):
    ccm = (-1, 0, -1)
    W = svo.World.generate(2, 1, 2, 128, 8, chunkcoordmin=ccm)
    O = oracle.OracleWorld.from_chunks([W.chunk(i) for i in range(4)], 2, 1, 2, 128, ccm)
    W.upload(0)
    yield W, O, (-128.0, 0.0, -128.0), (128.0, 128.0, 128.0)
    W.destroy()


@pytest.mark.parametrize("kernel", KERNELS)
def test_ray_lists(svo, oracle, world, kernel):
    W, O, lo, hi = world
    rng = np.random.default_rng(61)
    lists = {"random": random_rays(rng, 60000, lo, hi), "adversarial": adversarial_rays(rng, 40000, lo, hi),
             "creeping": creeping_rays(rng, 6000, lo, hi, 0.5), "creeping on chunk faces": creeping_rays(rng, 4000, lo, hi, 1.0, chunk_faces=True)}
    for name, (o, d) in lists.items():
        want = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=1), threads=8)
        got = W.chunkmarch(o, d, shadow=True, kernel=kid(svo, kernel), semantics=svo.SEMANTICS_GLSL)
        assert_gbuffer_equal(got, want, f"GLSL {name}/{kernel}")
        assert W.last_ray_count() == O.last_rays
    cpu = O.trace_rays(*lists["random"], params=oracle.make_params(shadow=True), threads=8)
    # the twins do differ: LEAF hits without the back-off, the shader's EPS, rays with the world behind them
    o, d = lists["random"]
    g = W.chunkmarch(o, d, kernel=kid(svo, kernel), semantics=svo.SEMANTICS_GLSL)
    assert ((cpu["flags"] & 1) != (g["flags"] & 1)).sum() > 100 and (cpu["t"] != g["t"]).sum() > 1000


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("eps,caps,light", [(0.0, (0, 0, 0), (0.3, -0.8, 0.5)), (1.0 / 8192.0, (1000, 1000, 1000), (1.0, -1.0, 0.0)),
                                            (1.0 / 1024.0, (3, 9, 5), (0.0, -1.0, 0.0)), (0.003, (0, 0, 0), (1.0, -1.0, 0.0))])
def test_constants_of_either_twin_and_odd_ones(svo, oracle, world, kernel, eps, caps, light):
    """The shader's march with its own constants, with the CPU code's (EPS 1/8192, caps 1000: only the guard, the entry and the LEAF
    distance differ then), with tiny caps, and with an EPS that is no power of two."""
    W, O, lo, hi = world
    o, d = random_rays(np.random.default_rng(62), 30000, lo, hi)
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=1, eps=eps, caps=caps, light_dir=light), threads=8)
    got = W.chunkmarch(o, d, shadow=True, kernel=kid(svo, kernel), semantics=svo.SEMANTICS_GLSL, eps=eps, caps=caps, light_dir=light)
    assert_gbuffer_equal(got, want, f"GLSL eps {eps} caps {caps}/{kernel}")
    assert (want["flags"] & 1).sum() > 500


@pytest.mark.parametrize("kernel", KERNELS)
def test_camera_frames_and_face_normals(svo, oracle, world, kernel):
    W, O, lo, hi = world
    cam = svo.make_camera((0.3, 150.0, -170.2), (0.0, -0.5, 0.866), (0, 1, 0), 60.0, 480, 270)
    for nm in (svo.NORMAL_CUBE, svo.NORMAL_FACE):
        want = O.trace_image(cam, params=oracle.make_params(shadow=True, semantics=1, normal_mode=nm), threads=8)
        got = W.draw(cam, shadow=True, kernel=kid(svo, kernel), semantics=svo.SEMANTICS_GLSL, normal_mode=nm)
        assert_gbuffer_equal(got, want, f"GLSL frame, normal mode {nm}/{kernel}")
    assert (want["flags"] & 1).sum() > 20000
    # several frames per launch
    out = svo.DeviceBuffer(3 * 480 * 270 * 32)
    W.trace_frames([cam, cam, cam], svo.trace_params(shadow=True, kernel=kid(svo, kernel), semantics=svo.SEMANTICS_GLSL, normal_mode=svo.NORMAL_FACE), (0, 0, 480, 270), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    three = out.to_numpy(svo.HIT_DTYPE, 3 * 480 * 270).reshape(3, 270, 480)
    for f in range(3):
        assert_gbuffer_equal(three[f], want, f"GLSL frame {f} of 3/{kernel}")
    out.free()


@pytest.mark.parametrize("kernel", KERNELS)
def test_chunk_positions_that_do_not_contain_their_points(svo, oracle, kernel):
    """The CPU march ends a ray whose position lies outside the box of the chunk World::index finds for it
    (src/Traverse.cpp:154-155); the shader has no such check - its treemarch fails at once and rootmarch steps on out of THAT
    box (shaders/Chunkmarch.glsl:297-330).  A world whose two chunks sit at each other's positions shows the difference."""
    G = svo.World.generate(2, 1, 1, 128, 6)
    a, b = G.chunk(0), G.chunk(1)
    a["position"], b["position"] = (128.0, 0.0, 0.0), (0.0, 0.0, 0.0)
    W = svo.World.create([a, b], 2, 1, 1, 128)
    O = oracle.OracleWorld.from_chunks([a, b], 2, 1, 1, 128)
    W.upload(0)
    o, d = random_rays(np.random.default_rng(63), 20000, (0, 0, 0), (256, 128, 128))
    for sem in (0, 1):
        want = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=sem), threads=8)
        got = W.chunkmarch(o, d, shadow=True, kernel=kid(svo, kernel), semantics=sem)
        assert_gbuffer_equal(got, want, f"swapped chunks, semantics {sem}/{kernel}")
    W.destroy()


def test_unknown_semantics_is_refused(svo, world):
    W, O, lo, hi = world
    o, d = random_rays(np.random.default_rng(64), 64, lo, hi)
    with pytest.raises(svo.SvoError) as e:
        W.chunkmarch(o, d, semantics=2)
    assert e.value.code == -1


def test_deep_sparse_and_mixed_depth_worlds(svo, oracle):
    """Many wide levels per descent (depth 13 refined in a band) and chunks of different depths, stack kernel, GLSL twin."""
    W = svo.World.generate(1, 1, 1, 128, 13, coarse_depth=7, refine_box=((40.0, 0.0, 0.0), (56.0, 128.0, 128.0)))
    O = oracle.OracleWorld.from_chunks([W.chunk(0)], 1, 1, 1, 128)
    W.upload(0)
    o, d = random_rays(np.random.default_rng(65), 30000, (30, 0, 0), (70, 128, 128))
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=1), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k, semantics=1), want, f"GLSL deep sparse/{k}")
    W.destroy()
    gen = {dp: svo.World.generate(2, 1, 2, 128, dp) for dp in (3, 8, 5, 6)}
    chunks = [gen[dp].chunk(i) for i, dp in enumerate((3, 8, 5, 6))]
    W = svo.World.create(chunks, 2, 1, 2, 128)
    O = oracle.OracleWorld.from_chunks(chunks, 2, 1, 2, 128)
    W.upload(0)
    o, d = random_rays(np.random.default_rng(66), 40000, (0, 0, 0), (256, 128, 256))
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=1), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k, semantics=1), want, f"GLSL mixed depths/{k}")
    W.destroy()
