"""Two independent restatements of the reference's CPU path - oracle/svo_oracle.c (C, the checker of every GPU test) and
oracle/svo_oracle_py.py (pure Python, numpy float32 scalars) - must agree bit for bit.  The reference holds no fixtures and
cannot be built here (parity unpinned), so this agreement is what stands in for a pin:

* world generation: glm::simplex, BoundsPyramid (base, every min / max level, the bilinear path), grow(), the water
  Ocroot::build - pools index for index, capacities included (src/BoundsPyramid.cpp, src/Octree.cpp:74-176, src/World.cpp:296-321);
* Ocroot::build / destroy / replace: a sequence of edits, pools and dirty ranges after every one (src/Octree.cpp:203-443);
* the march with EVERY field of the G-buffer record - hit flag, t, normal (cubeNormal, NaNs included), material, chunk, node,
  brick cell, shadow flags - and the reference work counters, on the C1 scene (one depth-8 chunk, 256x256 camera image cut +
  random + adversarial + creeping rays, shadow rays on) and on multi-chunk worlds with negative coordinates;
* the predicates and the camera ray.

CPU only; pure-Python loops, sized to finish in well under a minute."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

from helpers import adversarial_rays, creeping_rays, random_rays

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import svo_oracle_py as pyo  # noqa: E402


def py_world_of(O, n, w, h, d, ccm):
    """The Python restatement's view of the C oracle's world (same pools)."""
    chunks = []
    for i in range(n):
        c = O.chunk(i)
        chunks.append(pyo.Chunk(c["position"], c["size"], c["depth"], c["tree"], c["twig"]))
    return pyo.World(chunks, w, h, d, 128, ccm)


def assert_records_identical(got, want, what):
    """Every field of every record, bit for bit (t and normals as bit patterns: NaN normals must be the same NaNs)."""
    assert got.shape == want.shape, what
    for f in ("flags", "material", "chunk", "node", "cell"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, f"{what}: {f} differs at rays {bad[:6]}: python {got[f][bad[:6]]} C {want[f][bad[:6]]}"
    bad = np.nonzero(got["t"].view(np.uint32) != want["t"].view(np.uint32))[0]
    assert bad.size == 0, f"{what}: t differs at rays {bad[:6]}: python {got['t'][bad[:6]]} C {want['t'][bad[:6]]}"
    hit = (want["flags"] & 1) != 0
    gn, wn = got["normal"][hit], want["normal"][hit]
    same = (gn.view(np.uint32) == wn.view(np.uint32)) | (np.isnan(gn) & np.isnan(wn))
    assert np.all(same), f"{what}: normal differs at hits {np.nonzero(~same.all(axis=1))[0][:6]}"


def assert_pools_identical(P, O, n, what):
    for i in range(n):
        c, p = O.chunk(i), P.chunk[i]
        assert tuple(np.float32(v) for v in c["position"]) == p.position and np.float32(c["size"]) == p.size and c["depth"] == p.depth, (what, i)
        assert np.array_equal(c["tree"], p.tree_array()), f"{what}: tree[] of chunk {i}"
        assert np.array_equal(c["twig"], p.twig_array()), f"{what}: twig[] of chunk {i}"
        assert (c["treestoragesize"], c["twigstoragesize"]) == (p.treestoragesize, p.twigstoragesize), f"{what}: capacities of chunk {i}"


# ---------------------------------------------------------------------------------------------------------------------
def test_simplex_noise_agrees(oracle):
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-300, 300, 4000), rng.integers(-40, 40, 500).astype(np.float64), np.arange(256) / 256.0]).astype(np.float32)
    y = np.concatenate([rng.uniform(-300, 300, 4000), rng.integers(-40, 40, 500).astype(np.float64), np.arange(256)[::-1] / 256.0]).astype(np.float32)
    got = pyo.simplex2(x, y)
    want = np.array([oracle.lib.orc_simplex2(float(a), float(b)) for a, b in zip(x, y)], np.float32)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(want).max() <= 1.0 and np.abs(want).max() > 0.5


@pytest.mark.parametrize("size,xshift,zshift", [(64, 0.0, 0.0), (256, 512.0, -256.0)])
def test_bounds_pyramid_agrees_level_by_level(oracle, size, xshift, zshift):
    """BoundsPyramid::init (src/BoundsPyramid.cpp:47-135): the base and every min / max mip, and bound() on both paths."""
    L = oracle.lib
    L.orc_pyramid_init.argtypes = [C.POINTER(oracle.Pyramid), C.c_size_t, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float]
    L.orc_pyramid_deinit.argtypes = [C.POINTER(oracle.Pyramid)]
    L.orc_pyramid_min.argtypes = L.orc_pyramid_max.argtypes = [C.POINTER(oracle.Pyramid), C.c_float, C.c_float, C.c_size_t]
    L.orc_pyramid_min.restype = L.orc_pyramid_max.restype = C.c_float
    cp = oracle.Pyramid()
    L.orc_pyramid_init(C.byref(cp), size, 64.0, 1.0 / size, xshift, 16.0, zshift)
    pp = pyo.BoundsPyramid(size, 64.0, np.float32(1.0) / np.float32(size), xshift, 16.0, zshift)
    assert pp.levels == cp.levels
    for lv in range(cp.levels + 1):
        s = 1 << lv
        cmin = np.ctypeslib.as_array(cp.minquad[lv], shape=(s * s,))
        cmax = np.ctypeslib.as_array(cp.maxquad[lv], shape=(s * s,))
        assert np.array_equal(cmin.view(np.uint32), pp.minquad[lv].reshape(-1).view(np.uint32)), f"min level {lv}"
        assert np.array_equal(cmax.view(np.uint32), pp.maxquad[lv].reshape(-1).view(np.uint32)), f"max level {lv}"
    rng = np.random.default_rng(4)
    for _ in range(400):
        x, z = np.float32(rng.random()), np.float32(rng.random())
        lv = int(rng.integers(0, cp.levels + 3))           # levels + 1, + 2: the bilinear path beyond the base
        for cf, pf in ((L.orc_pyramid_min, pp.min), (L.orc_pyramid_max, pp.max)):
            a, b = np.float32(cf(C.byref(cp), float(x), float(z), lv)), np.float32(pf(x, z, lv))
            assert a.view(np.uint32) == b.view(np.uint32), (float(x), float(z), lv)
    L.orc_pyramid_deinit(C.byref(cp))


@pytest.mark.parametrize("w,h,d,depth,ccm,kw", [
    (2, 1, 2, 6, (0, 0, 0), {}),                                     # a depth-6 2x1x2 world with water
    (1, 1, 1, 8, (0, 0, 0), {}),                                     # the C1 scene
    (2, 2, 1, 5, (-1, -1, 0), {}),                                   # negative chunk coordinates, a chunk layer above the terrain
    (1, 1, 2, 7, (3, 0, -2), dict(pyramid_resolution=32)),           # pyramid coarser than the tree: bound()'s bilinear path in grow()
    (2, 1, 1, 6, (0, 0, 0), dict(water=False, seed=11, amplitude=40.0, yshift=30.0)),
    (1, 1, 1, 6, (0, 0, 0), dict(water_level=5.5, water_material=9)),  # a water plane off the voxel lattice
])
def test_generated_worlds_are_identical_index_for_index(oracle, w, h, d, depth, ccm, kw):
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm, **kw)
    P = pyo.World.generate(w, h, d, 128, depth, chunkcoordmin=ccm, **kw)
    assert_pools_identical(P, O, w * h * d, f"{w}x{h}x{d} depth {depth}")
    assert sum(c.twigs for c in P.chunk) > 0


def test_edit_sequence_pools_and_dirty_ranges(oracle):
    """Ocroot::build / destroy / replace (src/Octree.cpp:203-443): ten edits of all three kinds on a depth-6 chunk pair - boxes
    from a voxel to a third of a chunk, on and off the lattice, overlapping earlier edits - with the pools, their capacities and
    both Ocdelta dirty ranges compared after every edit."""
    O = oracle.OracleWorld.generate(2, 1, 1, 128, 6)
    P = pyo.World.generate(2, 1, 1, 128, 6)
    assert_pools_identical(P, O, 2, "before the edits")
    L = oracle.lib
    edits = [
        ("build", 0, (20, 60, 20), (50, 90, 50), 5),
        ("destroy", 0, (30, 0, 30), (60, 128, 44), 0),             # cuts through terrain, water and the box just built
        ("build", 0, (64.0, 64.0, 64.0), (66.0, 66.0, 66.0), 7),    # exactly one brick-aligned 2-unit cube
        ("replace", 0, (10.3, 2.1, 10.7), (47.9, 30.2, 33.3), 9),   # off the lattice
        ("destroy", 0, (33.0, 70.0, 33.0), (33.5, 70.5, 33.5), 0),  # a single voxel of the first box
        ("build", 1, (128, 0, 0), (256, 3, 128), 6),                # a slab across the whole second chunk
        ("destroy", 1, (150.25, 0.0, 20.25), (200.75, 128.0, 90.75), 0),
        ("replace", 1, (190, 10, 40), (230, 50, 80), 5),
        ("build", 0, (0, 100, 0), (128, 128, 128), 4),              # fills the top of chunk 0: LEAF nodes where whole nodes fit
        ("destroy", 0, (0, 0, 0), (128, 128, 128), 0),              # everything: the root becomes EMPTY
    ]
    for k, (kind, ci, lo, hi, mat) in enumerate(edits):
        dt, dw = oracle.Delta(), oracle.Delta()
        root = C.byref(O.w.chunk[ci])
        if kind in ("destroy", "replace"):
            L.orc_destroy(root, oracle.vec3(lo), oracle.vec3(hi), C.byref(dt), C.byref(dw))
        if kind == "replace":                                       # Ocroot::replace resets the deltas once, then destroys and builds
            dt2, dw2 = oracle.Delta(), oracle.Delta()
            L.orc_build(root, oracle.vec3(lo), oracle.vec3(hi), mat, C.byref(dt2), C.byref(dw2))
            dt.left, dt.right, dt.realloc_ = min(dt.left, dt2.left), max(dt.right, dt2.right), dt.realloc_ | dt2.realloc_
            dw.left, dw.right, dw.realloc_ = min(dw.left, dw2.left), max(dw.right, dw2.right), dw.realloc_ | dw2.realloc_
        if kind == "build":
            L.orc_build(root, oracle.vec3(lo), oracle.vec3(hi), mat, C.byref(dt), C.byref(dw))
        c = P.chunk[ci]
        ptree, ptwig = {"build": lambda: c.build(lo, hi, mat), "destroy": lambda: c.destroy(lo, hi), "replace": lambda: c.replace(lo, hi, mat)}[kind]()
        assert_pools_identical(P, O, 2, f"after edit {k} ({kind})")
        assert (dt.left, dt.right, bool(dt.realloc_)) == ptree.as_tuple(), f"edit {k}: tree dirty range"
        assert (dw.left, dw.right, bool(dw.realloc_)) == ptwig.as_tuple(), f"edit {k}: twig dirty range"
    assert pyo.node_type(P.chunk[0].tree[0]) == pyo.EMPTY
    # the march over the edited world agrees too (material 5 / 9 voxels, split LEAF nodes, emptied bricks)
    o, d = random_rays(np.random.default_rng(12), 1500, (0, 0, 0), (256, 128, 128))
    want = O.trace_rays(o, d, params=oracle.make_params(shadow=True), counters=True)
    got, cnt, rays = pyo.trace_rays(P, o, d, counters=True, shadow=True)
    assert_records_identical(got, want[0], "edited world")
    assert np.array_equal(cnt, want[1]) and rays == O.last_rays


def test_c1_scene_every_field_of_every_record(oracle, svo):
    """BASELINE configs[0]: one depth-8 chunk.  A 64x64 cut of the 256x256 camera image, 4 096 random rays, 600 adversarial
    rays and 300 creeping rays, primary + shadow: every field, the counters, and the ray totals."""
    O = oracle.OracleWorld.generate(1, 1, 1, 128, 8)
    P = pyo.World.generate(1, 1, 1, 128, 8)
    assert_pools_identical(P, O, 1, "C1")
    lo, hi = (0, 0, 0), (128, 128, 128)
    rng = np.random.default_rng(81)
    # camera rays of the image's centre cut, from the Python restatement of the build's camera
    cam = svo.default_camera(1, 1, 128, 256, 256)
    pcam = dict(eye=tuple(cam.eye), forward=tuple(cam.forward), right=tuple(cam.right), up=tuple(cam.up),
                tan_half_x=cam.tan_half_x, tan_half_y=cam.tan_half_y, width=cam.width, height=cam.height)
    co, cd = [], []
    ocam = oracle.camera_from(cam)
    for py in range(96, 160):
        for px in range(96, 160):
            eo, ed = pyo.camera_ray(pcam, px, py)
            vo, vd = oracle.Vec3(), oracle.Vec3()
            oracle.lib.orc_camera_ray(C.byref(ocam), px, py, C.byref(vo), C.byref(vd))
            assert tuple(np.float32(v) for v in (vd.x, vd.y, vd.z)) == ed and (vo.x, vo.y, vo.z) == tuple(float(v) for v in eo)
            co.append(eo); cd.append(ed)
    lists = {
        "camera": (np.array(co, np.float32), np.array(cd, np.float32), {}),
        "random": (*random_rays(rng, 4096, lo, hi), {}),
        "adversarial": (*adversarial_rays(rng, 600, lo, hi), {}),
        # creeping rays: pinned on voxel / brick / node faces (thousands of EPS steps each under the default caps: the caps are
        # lowered so that pure Python finishes, to values that still run out INSIDE creeping stretches)
        "creeping": (*creeping_rays(rng, 300, lo, hi, 0.5), dict(caps=(1000, 300, 60))),
        "creeping, GLSL EPS": (*creeping_rays(rng, 100, lo, hi, 0.5), dict(caps=(50, 200, 40), eps=1.0 / 4096.0)),
    }
    total_hits = 0
    for name, (o, d, kw) in lists.items():
        prm = oracle.make_params(shadow=True, caps=kw.get("caps", (0, 0, 0)), eps=kw.get("eps", 0.0))
        want, wcnt = O.trace_rays(o, d, params=prm, counters=True)
        got, cnt, rays = pyo.trace_rays(P, o, d, counters=True, shadow=True, **({"caps": kw["caps"]} if "caps" in kw else {}),
                                        **({"eps": np.float32(kw["eps"])} if "eps" in kw else {}))
        assert_records_identical(got, want, f"C1 {name}")
        assert np.array_equal(cnt, wcnt), f"C1 {name}: counters"
        assert rays == O.last_rays
        total_hits += int((want["flags"] & 1).sum())
    assert total_hits > 1500
    # face normals (the build's SVO_NORMAL_FACE) are restated twice as well
    o, d, _ = lists["random"]
    want = O.trace_rays(o[:800], d[:800], params=oracle.make_params(shadow=True, normal_mode=1))
    got, _ = pyo.trace_rays(P, o[:800], d[:800], shadow=True, normal_mode=1)
    assert_records_identical(got, want, "C1 face normals")


@pytest.mark.parametrize("w,h,d,depth,ccm", [(2, 1, 2, 6, (0, 0, 0)), (2, 2, 1, 4, (-1, -1, 0)), (3, 1, 1, 5, (-2, 0, 5))])
def test_multi_chunk_worlds_every_field(oracle, w, h, d, depth, ccm):
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm)
    P = py_world_of(O, w * h * d, w, h, d, ccm)            # the Python march over the C oracle's pools this time
    lo = np.array(ccm, np.float64) * 128
    hi = lo + np.array([w, h, d]) * 128
    rng = np.random.default_rng(17)
    o, dirs = random_rays(rng, 1200, lo, hi)
    ao, ad = adversarial_rays(rng, 400, lo, hi)
    # axis-parallel / on-lattice specials: inf and NaN reciprocals, origins on faces, origins outside the world
    sp_o = [[64, 100, 64], [lo[0], 30, lo[2]], [hi[0], 30, hi[2]], [32, 127.5, 32], [0, 0, 0], [64, 64, -20], [300, 50, 64]]
    sp_d = [[0, -1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [-1, 0, 0]]
    o = np.concatenate([o, ao, np.array(sp_o, np.float32)])
    dirs = np.concatenate([dirs, ad, np.array(sp_d, np.float32)])
    for light in ((1.0, -1.0, 0.0), (0.0, -1.0, 0.0), (0.3, -0.8, 0.5)):
        want, wcnt = O.trace_rays(o, dirs, params=oracle.make_params(shadow=True, light_dir=light), counters=True)
        got, cnt, rays = pyo.trace_rays(P, o, dirs, counters=True, shadow=True, light_dir=light)
        assert_records_identical(got, want, f"light {light}")
        assert np.array_equal(cnt, wcnt) and rays == O.last_rays
    assert (want["flags"] & 1).sum() > 200
    # a creeping list pinned on chunk faces: chunkmarch, treemarch and twigmarch creep together
    o, dirs = creeping_rays(rng, 200, lo, hi, 128.0 / 2 ** depth, chunk_faces=True)
    want = O.trace_rays(o, dirs, params=oracle.make_params(shadow=True, caps=(6, 40, 30)))
    got, _ = pyo.trace_rays(P, o, dirs, shadow=True, caps=(6, 40, 30))
    assert_records_identical(got, want, "chunk-face creep")


@pytest.mark.parametrize("w,h,d,depth,ccm", [(1, 1, 1, 8, (0, 0, 0)), (2, 1, 2, 6, (-1, 0, -1))])
def test_glsl_twin_every_field(oracle, w, h, d, depth, ccm):
    """svo_trace_params.semantics = SVO_SEMANTICS_GLSL: the march of shaders/Chunkmarch.glsl (EPS 1/4096, caps 256 / 512 / 64, the
    BIGEPS guard, tnear > 0 at the world entry, no containment re-check, LEAF hits at t) restated twice as well - same lists as
    the CPU march, every field and the counters; and the two twins do differ where SURVEY.md App. B says they do."""
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm)
    P = py_world_of(O, w * h * d, w, h, d, ccm)
    lo = np.array(ccm, np.float64) * 128
    hi = lo + np.array([w, h, d]) * 128
    rng = np.random.default_rng(23)
    lists = {"random": random_rays(rng, 2500, lo, hi), "adversarial": adversarial_rays(rng, 600, lo, hi),
             "creeping": creeping_rays(rng, 300, lo, hi, 128.0 / 2 ** depth), "creeping on chunk faces": creeping_rays(rng, 200, lo, hi, 1.0, chunk_faces=True)}
    for name, (o, dirs) in lists.items():
        want, wcnt = O.trace_rays(o, dirs, params=oracle.make_params(shadow=True, semantics=1), counters=True)
        rays_c = O.last_rays
        got, cnt, rays = pyo.trace_rays(P, o, dirs, counters=True, shadow=True, semantics=1)
        assert_records_identical(got, want, f"GLSL {name}")
        assert np.array_equal(cnt, wcnt) and rays == rays_c, name
        assert wcnt[:, 3].max() <= 2 * 256 * 512 and wcnt[:, 2].max() <= 2 * 256      # the shader's caps (primary + shadow ray)
        if name.startswith("creeping"):
            cpu, ccnt = O.trace_rays(o, dirs, params=oracle.make_params(shadow=True, caps=(1000, 200, 60)), counters=True)
            # the guard at work: a pinned ray advances by BIGEPS = 1/16 per step, not by EPS
            assert (ccnt[:, 3] >= 200).sum() > 5 and wcnt[:, 3].astype(np.int64).sum() < ccnt[:, 3].astype(np.int64).sum() // 2, (wcnt[:, 3].sum(), ccnt[:, 3].sum())
    # explicit constants override the twin's own; a world seen from outside with the box BEHIND the ray: the CPU march enters it
    # (tfar > tnear alone, src/Traverse.cpp:123), the shader does not (tnear > 0, shaders/Chunkmarch.glsl:124)
    o, dirs = lists["random"]
    want = O.trace_rays(o[:600], dirs[:600], params=oracle.make_params(shadow=True, semantics=1, eps=1.0 / 1024.0, caps=(5, 40, 9)))
    got, _ = pyo.trace_rays(P, o[:600], dirs[:600], shadow=True, semantics=1, eps=1.0 / 1024.0, caps=(5, 40, 9))
    assert_records_identical(got, want, "GLSL, explicit eps and caps")
    behind_o = np.array([[hi[0] + 50.0, 60.0, (lo[2] + hi[2]) / 2]], np.float32)
    behind_d = np.array([[1.0, 0.0, 0.0]], np.float32)
    assert oracle.OracleWorld.trace_rays(O, behind_o, behind_d, params=oracle.make_params(semantics=1))["flags"][0] == 0
    g, _ = pyo.trace_rays(P, behind_o, behind_d, semantics=1)
    assert g["flags"][0] == 0


def test_python_predicates_match_c(oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib
    for _ in range(400):
        a = rng.uniform(-2, 3, 3).astype(np.float32)
        b = rng.normal(size=3).astype(np.float32)
        if rng.random() < 0.3:
            b[rng.integers(0, 3)] = 0.0           # axis-parallel: infinite reciprocal
        if rng.random() < 0.3:
            a[rng.integers(0, 3)] = np.float32(rng.integers(0, 2))   # exactly on a face
        lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
        e_c = L.orc_cubeEscapeDistance(oracle.vec3(a), oracle.vec3(b), oracle.vec3(lo), oracle.vec3(hi))
        e_p = pyo.cube_escape_distance(tuple(a), tuple(b), tuple(lo), tuple(hi))
        assert (np.isnan(e_c) and np.isnan(e_p)) or np.float32(e_c).view(np.uint32) == np.float32(e_p).view(np.uint32)
        assert bool(L.orc_isInsideCube(oracle.vec3(a), oracle.vec3(lo), oracle.vec3(hi))) == pyo.is_inside_cube(tuple(a), tuple(lo), tuple(hi))
        flag = C.c_int()
        t_c = L.orc_intersectCube(oracle.vec3(a), oracle.vec3(b), oracle.vec3(lo), oracle.vec3(hi), C.byref(flag))
        t_p, hit_p = pyo.intersect_cube(tuple(a), tuple(b), tuple(lo), tuple(hi))
        assert bool(flag.value) == hit_p and ((np.isnan(t_c) and np.isnan(t_p)) or np.float32(t_c).view(np.uint32) == np.float32(t_p).view(np.uint32))
    # (heightMaterial is static in the C oracle: it is reached through grow(), test_generated_worlds_are_identical_index_for_index)
