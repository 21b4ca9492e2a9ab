"""Hand-derived known answers that pin the CPU oracle (oracle/svo_oracle.c).

The reference holds no tests or golden vectors for this path and cannot be compiled here (GLM is
absent), so the oracle is "parity unpinned" against the reference itself; these cases pin it against
answers worked out by hand from src/Traverse.cpp / src/Octree.cpp / src/World.cpp.
"""
import ctypes as C
import math

import numpy as np
import pytest

EPS = np.float32(1.0 / 8192.0)
BRANCH, TWIG, LEAF, EMPTY = 2 << 30, 3 << 30, 1 << 30, 0


def one_chunk(oracle, tree, twig=None, depth=2, pos=(0, 0, 0), size=128.0, ccm=(0, 0, 0), dims=(1, 1, 1)):
    c = dict(position=pos, size=size, depth=depth, tree=np.array(tree, np.uint32),
             twig=np.zeros(0, np.uint16) if twig is None else np.asarray(twig, np.uint16))
    return oracle.OracleWorld.from_chunks([c], *dims, int(size), ccm)


def test_predicates(oracle):
    v = oracle.vec3
    L = oracle.lib
    # isInsideCube: closed box, NaN -> false (src/Traverse.cpp:18-23)
    assert L.orc_isInsideCube(v((0, 0, 0)), v((0, 0, 0)), v((1, 1, 1))) == 1
    assert L.orc_isInsideCube(v((1, 1, 1)), v((0, 0, 0)), v((1, 1, 1))) == 1
    assert L.orc_isInsideCube(v((1.0000001, 1, 1)), v((0, 0, 0)), v((1, 1, 1))) == 0
    assert L.orc_isInsideCube(v((float("nan"), 0.5, 0.5)), v((0, 0, 0)), v((1, 1, 1))) == 0
    # cubeEscapeDistance (src/Traverse.cpp:25-32): from the centre of the unit cube along +x -> 0.5
    assert L.orc_cubeEscapeDistance(v((0.5, 0.5, 0.5)), v((1, 0, 0)), v((0, 0, 0)), v((1, 1, 1))) == 0.5
    # diagonal (1,1,1)/sqrt3 from the origin corner: sqrt3
    s = 1 / math.sqrt(3)
    e = L.orc_cubeEscapeDistance(v((0, 0, 0)), v((s, s, s)), v((0, 0, 0)), v((1, 1, 1)))
    assert abs(e - math.sqrt(3)) < 1e-6
    # axis-parallel ray exactly on a face: (cmin - a) * inf = NaN propagates
    e = L.orc_cubeEscapeDistance(v((0, 0.5, 0.5)), v((0, 1, 0)), v((0, 0, 0)), v((1, 1, 1)))
    assert math.isnan(e)
    # intersectCube (src/Traverse.cpp:115-125): box [0,1]^3 from (-1,.5,.5) along +x: tnear 1, hit
    hit = C.c_int(0)
    t = L.orc_intersectCube(v((-1, 0.5, 0.5)), v((1, 0, 0)), v((0, 0, 0)), v((1, 1, 1)), C.byref(hit))
    assert t == 1.0 and hit.value == 1
    t = L.orc_intersectCube(v((-1, 2.5, 0.5)), v((1, 0, 0)), v((0, 0, 0)), v((1, 1, 1)), C.byref(hit))
    assert hit.value == 0


def test_root_leaf_hit_distance(oracle):
    """Root is one LEAF (material 3): a ray from outside enters at t = d + EPS and the CPU march
    reports s = t_tree - EPS with t_tree = 0, i.e. sigma = (d + EPS) + (0 - EPS) in float (Traverse.cpp:93,139,160)."""
    W = one_chunk(oracle, [LEAF | 3])
    o = np.array([[64, 64, -10]], np.float32)
    d = np.array([[0, 0, 1]], np.float32)
    h = W.trace_rays(o, d)[0]
    t_enter = np.float32(10.0) + EPS
    expect = np.float32(t_enter + np.float32(np.float32(0.0) - EPS))
    assert h["flags"] == 1 and h["material"] == 3 and h["node"] == 0 and h["cell"] == 0xFF and h["chunk"] == 0
    assert h["t"] == expect
    # normal: the sample point alpha + beta*(t - EPS) lies just outside the -z face -> (0,0,-1)
    assert tuple(h["normal"]) == (0.0, 0.0, -1.0)
    hit, sigma = W.chunkmarch((64, 64, -10), (0, 0, 1))
    assert hit and sigma[2] == np.float32(np.float32(-10.0) + np.float32(1.0) * expect)


def test_empty_root_misses_and_inside_solid(oracle):
    W = one_chunk(oracle, [EMPTY])
    h = W.trace_rays(np.array([[64, 64, -10]], np.float32), np.array([[0, 0, 1]], np.float32))[0]
    assert h["flags"] == 0 and h["t"] == 0
    # origin inside a solid root: LEAF at the first step, s = 0 - EPS (negative distance, as the reference)
    W = one_chunk(oracle, [LEAF | 1])
    h = W.trace_rays(np.array([[64, 64, 64]], np.float32), np.array([[1, 0, 0]], np.float32))[0]
    assert h["flags"] == 1 and h["t"] == np.float32(-EPS)


def test_branch_child_selection_and_slots(oracle):
    """Root BRANCH with exactly one LEAF child per test: slot = x + 2y + 4z, `>= mid` goes to the upper child."""
    for slot in range(8):
        tree = [BRANCH | 1] + [EMPTY] * 8
        tree[1 + slot] = LEAF | (slot + 1)
        W = one_chunk(oracle, tree, depth=3)
        cx, cy, cz = [(96.0 if (slot >> k) & 1 else 32.0) for k in range(3)]
        o = np.array([[cx, 200.0, cz]], np.float32)
        d = np.array([[0, -1, 0]], np.float32)
        h = W.trace_rays(o, d)[0]
        if (slot >> 1) & 1:       # upper-y child: hit on entering the chunk
            assert h["flags"] == 1 and h["node"] == 1 + slot and h["material"] == slot + 1
            assert abs(h["t"] - 72.0) < 1e-3
        else:                      # lower-y child: first crosses the EMPTY upper child (64 units)
            assert h["flags"] == 1 and h["node"] == 1 + slot and h["material"] == slot + 1
            assert abs(h["t"] - 136.0) < 1e-3
    # a point exactly on the mid plane belongs to the upper child (greaterThanEqual, Traverse.cpp:42)
    tree = [BRANCH | 1] + [EMPTY] * 8
    tree[1 + 1] = LEAF | 9      # slot 1 = upper x, lower y, lower z
    W = one_chunk(oracle, tree, depth=3)
    h = W.trace_rays(np.array([[64.0, 32.0, 32.0]], np.float32), np.array([[0, 1, 0]], np.float32))[0]
    assert h["flags"] == 1 and h["node"] == 2 and h["t"] == np.float32(-EPS)


def test_brick_cell_hit(oracle):
    """depth 2: the root is a TWIG; one solid cell (x=2,y=1,z=3) of 32-unit voxels."""
    twig = np.zeros(64, np.uint16)
    twig[3 * 16 + 1 * 4 + 2] = 7
    W = one_chunk(oracle, [TWIG | 0], twig=twig, depth=2)
    # ray along +z through the cell column x in [64,96), y in [32,64): enters the chunk at z=0, cell starts at z=96
    o = np.array([[80.0, 48.0, -5.0]], np.float32)
    d = np.array([[0, 0, 1]], np.float32)
    h = W.trace_rays(o, d)[0]
    assert h["flags"] == 1 and h["material"] == 7 and h["cell"] == 3 * 16 + 1 * 4 + 2 and h["node"] == 0
    # entry at t = 5 + EPS puts p.z at EPS; every empty cell is left at "its far face + EPS" (the escape is
    # measured from the actual position, so the EPS does not accumulate): brick t = 96, sigma = 101 + EPS
    assert abs(float(h["t"]) - (101.0 + float(EPS))) < 2e-5
    assert tuple(h["normal"]) == (0.0, 0.0, -1.0)
    # a neighbouring column misses
    h = W.trace_rays(np.array([[16.0, 48.0, -5.0]], np.float32), d)[0]
    assert h["flags"] == 0


def test_world_index_and_index_float(oracle):
    """World::index is toroidal, index_float 'floors' negatives with the reference's off-by-one (World.cpp:276-293,323-332)."""
    chunks = [dict(position=(x * 128.0 - 128.0, 0.0, z * 128.0 - 128.0), size=128.0, depth=2, tree=np.array([EMPTY], np.uint32), twig=np.zeros(0, np.uint16))
              for z in range(2) for x in range(2)]
    W = oracle.OracleWorld.from_chunks(chunks, 2, 1, 2, 128, (-1, 0, -1))
    L = oracle.lib
    assert L.orc_world_index3(C.byref(W.w), 0, 0, 0) == 0
    assert L.orc_world_index3(C.byref(W.w), -1, 0, 0) == 1          # modulo(-1, 2) = 1
    assert L.orc_world_index3(C.byref(W.w), 3, 5, -3) == 1 + 2 * 1  # x=3->1, z=-3->1
    q = (C.c_int * 3)()
    L.orc_world_index_float(C.byref(W.w), oracle.vec3((-0.5, 10.0, 200.0)), q)
    assert tuple(q) == (-1, 0, 1)
    L.orc_world_index_float(C.byref(W.w), oracle.vec3((-128.0, 0.0, -256.0)), q)
    assert tuple(q) == (-2, 0, -3)        # exact negative integers land one chunk too low, as in the reference


def test_shadow_ray_definition(oracle):
    """Primary hits the top of a solid lower half; the shadow ray toward -light (default light (1,-1,0))
    leaves upward through empty space -> traced, not shadowed.  With the light pointing up it is blocked."""
    tree = [BRANCH | 1] + [LEAF | 2, LEAF | 2, EMPTY, EMPTY, LEAF | 2, LEAF | 2, EMPTY, EMPTY]   # lower-y children solid
    W = one_chunk(oracle, tree, depth=3)
    o = np.array([[40.0, 120.0, 40.0]], np.float32)
    d = np.array([[0, -1, 0]], np.float32)
    h = W.trace_rays(o, d, params=oracle.make_params(shadow=True))[0]
    assert h["flags"] == 1 | 2 and abs(h["t"] - 56.0) < 1e-3 and W.last_rays == 2
    h = W.trace_rays(o, d, params=oracle.make_params(shadow=True, light_dir=(0.0, 1.0, 0.0)))[0]
    assert h["flags"] == 1 | 2 | 4        # shadow ray goes straight down into the solid it stands on


def test_step_caps_are_honoured(oracle):
    """max_tree_steps = 1: the single allowed tree step crosses one EMPTY child, then treemarch gives up;
    chunkmarch then escapes the chunk and misses (Traverse.cpp:79,164-168)."""
    tree = [BRANCH | 1] + [LEAF | 2, LEAF | 2, EMPTY, EMPTY, LEAF | 2, LEAF | 2, EMPTY, EMPTY]
    W = one_chunk(oracle, tree, depth=3)
    o = np.array([[40.0, 120.0, 40.0]], np.float32)
    d = np.array([[0, -1, 0]], np.float32)
    assert W.trace_rays(o, d, params=oracle.make_params(caps=(0, 2, 0)))[0]["flags"] == 1
    assert W.trace_rays(o, d, params=oracle.make_params(caps=(0, 1, 0)))[0]["flags"] == 0


def test_grow_matches_hand_built_tiny_chunk(oracle):
    """depth-2 chunk: grow() emits a single node.  With amplitude 0 the height field is flat at yshift:
    below the chunk -> EMPTY, above it -> LEAF(material by normalised y = 0 -> 1), inside -> TWIG whose
    cells are solid where h >= cell floor (src/Octree.cpp:105-154)."""
    for yshift, expect in ((-5.0, "empty"), (200.0, "leaf"), (40.0, "twig")):
        O = oracle.OracleWorld.generate(1, 1, 1, 128, 2, amplitude=0.0, yshift=yshift, water=False)
        c = O.chunk(0)
        assert c["tree"].size == 1
        word = int(c["tree"][0])
        if expect == "empty":
            assert word >> 30 == 0
        elif expect == "leaf":
            assert word >> 30 == 1 and word & 0x3FFFFFFF == 1
        else:
            assert word >> 30 == 3 and c["twig"].size == 64
            cells = c["twig"].reshape(4, 4, 4)        # [z][y][x]
            # h = 40: cell floors 0 and 32 are <= 40 -> layers y=0,1 solid, y=2,3 empty, for every column
            assert np.all(cells[:, 0, :] == 1) and np.all(cells[:, 1, :] == 1)
            assert np.all(cells[:, 2, :] == 0) and np.all(cells[:, 3, :] == 0)


def test_water_build_fills_below_level(oracle):
    """Ocroot::build(y <= 6, material 6) on an all-EMPTY depth-4 chunk: closed-box semantics fill every
    voxel whose box touches [0,6] in y, i.e. voxel layers y=0 (8 units each: [0,8]) only (Octree.cpp:320-436)."""
    O = oracle.OracleWorld.generate(1, 1, 1, 128, 4, amplitude=0.0, yshift=-50.0, water=True, water_level=6.0, water_material=6)
    o = np.array([[x + 0.5, 100.0, z + 0.5] for x in (3, 64, 120) for z in (5, 77)], np.float32)
    d = np.tile(np.array([[0, -1, 0]], np.float32), (len(o), 1))
    h = O.trace_rays(o, d)
    assert np.all(h["flags"] == 1) and np.all(h["material"] == 6)
    assert np.all(np.abs(h["t"] - 92.0) < 1e-2)       # water surface at y = 8 (top of the first voxel layer)
