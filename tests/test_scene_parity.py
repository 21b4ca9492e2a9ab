"""Host-side world generation of libsvo_amd (terrain.cpp) against the oracle's restatement of
World::init / grow() / BoundsPyramid / Ocroot::build: pools must be bit-identical (CPU only)."""
import numpy as np
import pytest

CASES = [
    dict(w=1, h=1, d=1, depth=2),
    dict(w=1, h=1, d=1, depth=5),
    dict(w=1, h=1, d=1, depth=8),                                   # the reference's TREE_MAX_DEPTH
    dict(w=2, h=1, d=2, depth=6),
    dict(w=2, h=2, d=2, depth=5, ccm=(-1, -1, -1)),                 # negative chunk coordinates
    dict(w=3, h=1, d=2, depth=6, seed=1234),
    dict(w=1, h=1, d=1, depth=7, water=False),
    dict(w=1, h=1, d=1, depth=8, pyramid_resolution=64),            # pyramid coarser than the tree: bilinear path
    dict(w=1, h=1, d=1, depth=6, amplitude=20.0, yshift=64.0, water_level=70.0, water_material=9),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join(f"{k}{v}" for k, v in c.items()))
def test_generated_pools_match_oracle(svo, oracle, case):
    c = dict(case)
    w, h, d, depth = c.pop("w"), c.pop("h"), c.pop("d"), c.pop("depth")
    ccm = c.pop("ccm", (0, 0, 0))
    W = svo.World.generate(w, h, d, 128, depth, chunkcoordmin=ccm, threads=4, **c)
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm, **c)
    for i in range(w * h * d):
        a, b = W.chunk(i), O.chunk(i)
        assert a["position"] == b["position"] and a["size"] == b["size"] and a["depth"] == b["depth"]
        assert np.array_equal(a["tree"], b["tree"]), f"chunk {i}: node words differ"
        assert np.array_equal(a["twig"], b["twig"]), f"chunk {i}: bricks differ"
    info = W.info
    assert info.total_trees == sum(O.chunk(i)["tree"].size for i in range(w * h * d))
    assert info.exact_geometry == 1
    W.destroy()


def test_generated_tree_is_well_formed(svo):
    """BFS layout facts the kernels rely on: 1+8k nodes, children after parents, TWIGs only at level depth-2."""
    W = svo.World.generate(1, 1, 1, 128, 7)
    c = W.chunk(0)
    tree = c["tree"]
    assert (tree.size - 1) % 8 == 0
    types, offs = tree >> 30, tree & 0x3FFFFFFF
    level = np.full(tree.size, -1)
    level[0] = 0
    for i in range(tree.size):
        if level[i] < 0:
            continue
        if types[i] == 2:
            assert offs[i] > i and (offs[i] - 1) % 8 == 0 and offs[i] + 8 <= tree.size
            level[offs[i]:offs[i] + 8] = level[i] + 1
        elif types[i] == 3:
            assert level[i] == 7 - 2 and offs[i] < c["twig"].size // 64
    assert level.max() == 5
    # materials: heightMaterial clamps to 1..4, water is 6 (src/Octree.cpp:69-72, src/World.cpp:320)
    mats = set(np.unique(c["twig"])) | set(np.unique(offs[types == 1]))
    assert mats <= {0, 1, 2, 3, 4, 6}
    W.destroy()


def test_thread_count_does_not_change_the_world(svo):
    a = svo.World.generate(2, 1, 3, 128, 6, threads=1)
    b = svo.World.generate(2, 1, 3, 128, 6, threads=6)
    for i in range(6):
        assert np.array_equal(a.chunk(i)["tree"], b.chunk(i)["tree"]) and np.array_equal(a.chunk(i)["twig"], b.chunk(i)["twig"])
