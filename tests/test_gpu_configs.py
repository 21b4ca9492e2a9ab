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


def test_c3_off_lattice_eye(svo, oracle, c3):
    """The benchmark camera moved 0.31 off the chunk seam x = 256: the pixel column with the smallest |dir.x| now creeps
    along lattice planes for up to ~5 300 steps per pixel (on the seam those rays have dir.x == 0 and NaN-miss).  Both
    kernels against the oracle over the whole 1080p frame."""
    O = oracle.OracleWorld.from_chunks([c3.chunk(i, copy=False) for i in range(16)], 4, 1, 4, 128)
    cam = svo.default_camera(4, 4, 128, 1920, 1080)
    cam.eye[0] += 0.31
    want, wc = O.trace_image(cam, params=oracle.make_params(shadow=True), counters=True, threads=THREADS)
    steps = wc[..., 1].astype(np.int64) + wc[..., 2] + wc[..., 3]
    assert steps.max() > 3000                   # the creeping column is in the picture
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        got = c3.draw(cam, shadow=True, kernel=k)
        assert_gbuffer_equal(got, want, f"c3-off-lattice/{k}")
    check_properties(svo, got, True)


def test_c4_2160p_band_partition_of_8(svo, oracle, c3):
    """configs[3]: the 3840x2160 image as 8 ranks would trace it (8-row bands round-robin) equals the whole frame;
    every rank's bands are traced here on the one GPU and de-interleaved with the bench's helper.  The whole frame is
    also compared with the oracle, record for record."""
    cam = svo.default_camera(4, 4, 128, 3840, 2160)
    full = c3.draw(cam, shadow=True)
    O = oracle.OracleWorld.from_chunks([c3.chunk(i, copy=False) for i in range(16)], 4, 1, 4, 128)
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=THREADS)
    assert_gbuffer_equal(full, want, "c4/whole frame vs oracle")
    del want
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


def test_reference_default_scene(svo, oracle):
    """The reference's OWN default workload (SURVEY.md §5 "default workload"): `world.init(4, 4, 4, 128)` with TREE_MAX_DEPTH 8 and a
    256-texel pyramid (src/Main.cpp:80, src/World.cpp:10-11,19-43) at 1920x1080 - 64 chunks of depth 8, every entry of the stack
    kernel's LDS chunk table in use, a world four chunks high.  Seen (a) from the bench's kind of view, whole frame, and (b) from
    the reference's start position: the world's corner, eye ON the lattice at (0, 0, 0), looking along +z with a 90 degree field
    of view (src/Main.cpp:133-139).  Primary + shadow rays, both kernels, the CPU march and its GLSL twin, against the oracle."""
    W = svo.World.generate(4, 4, 4, 128, 8)
    assert W.info.exact_geometry == 1
    O = oracle.OracleWorld.from_chunks([W.chunk(i, copy=False) for i in range(64)], 4, 4, 4, 128)
    W.upload(0)
    views = {"bench view": svo.default_camera(4, 4, 128, 1920, 1080),
             "reference start": svo.make_camera((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), (0.0, 1.0, 0.0), 90.0, 960, 540),
             "inside, over the terrain": svo.make_camera((200.3, 70.2, 100.1), (0.3, -0.25, 0.9), (0.0, 1.0, 0.0), 90.0, 960, 540)}
    hits = 0
    for name, cam in views.items():
        for sem in (svo.SEMANTICS_CPU, svo.SEMANTICS_GLSL):
            want = O.trace_image(cam, params=oracle.make_params(shadow=True, semantics=sem), threads=THREADS)
            for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
                got = W.draw(cam, shadow=True, kernel=k, semantics=sem)
                assert_gbuffer_equal(got, want, f"reference default scene/{name}/semantics {sem}/{k}")
            assert W.last_ray_count() == O.last_rays
            hits += int((want["flags"] & 1).sum())
            if sem == svo.SEMANTICS_CPU: check_properties(svo, got, True)
    assert hits > 500000                                            # the terrain is in the pictures
    assert W.draw(views["bench view"], shadow=True).tobytes() == W.draw(views["bench view"], shadow=True, kernel=svo.KERNEL_STACK).tobytes()   # AUTO = the stack kernel here
    W.destroy()


def node_levels(tree):
    """Level of every node of a BFS-ordered chunk tree (src/Octree.cpp:98-104,165-173): level l+1 holds the 8 children
    of each BRANCH of level l, in order.  Returns the index of the first node of each level."""
    starts, pos, count = [0], 0, 1
    while count > 0 and pos < tree.size:
        branches = int(((tree[pos:pos + count] >> 30) == 2).sum())
        pos += count
        count = 8 * branches
        starts.append(pos)
    return np.array(starts)


def test_c5_depth16_sparse(svo, oracle):
    """configs[4]: depth-16 sparse SVO (full depth only inside the band 62 <= x <= 66, bricks at depth 10 elsewhere):
    14 branch levels exercise the deepest LDS-stack instantiation.  The camera hovers over the band, so a large part of
    the picture is depth-16 voxels (bricks under nodes of level 14, edge 128/65536); whole 1080p image against the oracle."""
    scene = svo.c5_scene()
    W = svo.World.generate(1, 1, 1, 128, 16, **scene["generate"])
    assert W.info.max_chunk_depth == 16 and W.info.exact_geometry == 1
    chunk = W.chunk(0, copy=False)
    O = oracle.OracleWorld.from_chunks([chunk], 1, 1, 1, 128)
    W.upload(0)
    cam = scene["camera"](1920, 1080)
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=THREADS)
    got = W.draw(cam, shadow=True, kernel=svo.KERNEL_STACK)
    assert_gbuffer_equal(got, want, "c5/stack")
    check_properties(svo, got, True)
    lit = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL)
    assert_gbuffer_equal(lit, want, "c5/literal")
    # the deep part of the tree is what the picture shows: >= 25 % of the primary hits end at a node of level 14 (a brick
    # of depth-16 voxels or a LEAF of that level: 14 BRANCH levels above it), the rest at coarse depth-10 bricks
    starts = node_levels(chunk["tree"])
    assert len(starts) - 2 == 14                                    # deepest level that holds nodes
    hit = (want["flags"] & 1) != 0
    level = np.searchsorted(starts, want["node"], side="right") - 1
    deep = hit & (level == 14)
    assert hit.mean() > 0.5
    assert deep.sum() >= 0.25 * hit.sum(), f"only {deep.sum() / hit.sum():.3f} of the hits sit at level 14"
    W.destroy()


@pytest.mark.parametrize("depth,half,deep_hits", [(17, 0.5, 500), (18, 0.125, 500), (19, 0.125, 100)])
def test_deepest_chunks(svo, oracle, depth, half, deep_hits):
    """Chunks deeper than BASELINE configs[4]: the stack kernel's wide entries keep the reference node's level in 5 bits and
    it marches up to 22 branch levels (chunk depth 24; round 2 stopped at depth 17 and sent depth 18 to the literal
    kernel; depth 19 runs the <22> instantiation).  Same sparse scene as configs[4], the refined band narrowed so that the pools stay small."""
    scene = svo.c5_scene()
    gen = dict(scene["generate"], refine_box=((64.3 - half, -1e9, -1e9), (64.3 + half, 1e9, 1e9)))
    W = svo.World.generate(1, 1, 1, 128, depth, **gen)
    assert W.info.max_chunk_depth == depth and W.info.exact_geometry == 1
    chunk = W.chunk(0, copy=False)
    assert len(node_levels(chunk["tree"])) - 2 == depth - 2          # nodes down to the deepest branch level
    O = oracle.OracleWorld.from_chunks([chunk], 1, 1, 1, 128)
    W.upload(0)
    cam = scene["camera"](480, 270)
    want = O.trace_image(cam, params=oracle.make_params(shadow=True), threads=THREADS)
    hit = (want["flags"] & 1) != 0
    level = np.searchsorted(node_levels(chunk["tree"]), want["node"], side="right") - 1
    assert hit.mean() > 0.5 and (hit & (level == depth - 2)).sum() > deep_hits      # the deepest level is in the picture
    assert_gbuffer_equal(W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL), want, f"depth {depth}/literal")
    assert_gbuffer_equal(W.draw(cam, shadow=True, kernel=svo.KERNEL_AUTO), want, f"depth {depth}/auto")
    assert_gbuffer_equal(W.draw(cam, shadow=True, kernel=svo.KERNEL_STACK), want, f"depth {depth}/stack")
    W.destroy()
