"""Two independent restatements of src/Traverse.cpp — oracle/svo_oracle.c (C) and oracle/svo_oracle_py.py (pure Python,
numpy float32 scalars) — must agree bit for bit: hit flag, chunk, node, brick cell and the float t.  CPU only, small cases."""
import os
import sys

import numpy as np
import pytest

from helpers import random_rays

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import svo_oracle_py as pyo  # noqa: E402


def build_py_world(O, n, w, h, d, ccm):
    chunks = []
    for i in range(n):
        c = O.chunk(i)
        chunks.append(pyo.Chunk(c["position"], c["size"], c["depth"], c["tree"], c["twig"]))
    return pyo.World(chunks, w, h, d, 128, ccm)


@pytest.mark.parametrize("w,h,d,depth,ccm", [(1, 1, 1, 5, (0, 0, 0)), (2, 1, 2, 4, (0, 0, 0)), (2, 2, 1, 4, (-1, -1, 0))])
def test_c_and_python_restatements_agree(oracle, w, h, d, depth, ccm):
    O = oracle.OracleWorld.generate(w, h, d, 128, depth, chunkcoordmin=ccm)
    P = build_py_world(O, w * h * d, w, h, d, ccm)
    lo = np.array(ccm, np.float64) * 128
    hi = lo + np.array([w, h, d]) * 128
    rng = np.random.default_rng(17)
    o, dirs = random_rays(rng, 250, lo, hi)
    # axis-parallel / on-lattice specials: inf and NaN reciprocals, origins on faces
    sp_o = [[64, 100, 64], [lo[0], 30, lo[2]], [hi[0], 30, hi[2]], [32, 127.5, 32], [0, 0, 0], [64, 64, -20], [300, 50, 64]]
    sp_d = [[0, -1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [1, 0, 0], [0, 0, 1], [-1, 0, 0]]
    o = np.concatenate([o, np.array(sp_o, np.float32)])
    dirs = np.concatenate([dirs, np.array(sp_d, np.float32)])
    want = O.trace_rays(o, dirs)
    hits = 0
    for k in range(len(o)):
        hit, t, chunk, node, cell = pyo.chunkmarch(o[k], dirs[k], P)
        assert hit == bool(want["flags"][k] & 1), f"ray {k}: hit flag"
        if hit:
            hits += 1
            assert (chunk, node, cell) == (int(want["chunk"][k]), int(want["node"][k]), int(want["cell"][k])), f"ray {k}: voxel id"
            assert np.float32(t).view(np.uint32) == want["t"][k].view(np.uint32), f"ray {k}: t {t} vs {want['t'][k]}"
    assert hits > 40


def test_python_predicates_match_c(oracle):
    rng = np.random.default_rng(3)
    L = oracle.lib
    for _ in range(300):
        a = rng.uniform(-2, 3, 3).astype(np.float32)
        b = rng.normal(size=3).astype(np.float32)
        if rng.random() < 0.3:
            b[rng.integers(0, 3)] = 0.0           # axis-parallel: infinite reciprocal
        if rng.random() < 0.3:
            a[rng.integers(0, 3)] = np.float32(rng.integers(0, 2))   # exactly on a face
        lo, hi = np.zeros(3, np.float32), np.ones(3, np.float32)
        e_c = L.orc_cubeEscapeDistance(oracle.vec3(a), oracle.vec3(b), oracle.vec3(lo), oracle.vec3(hi))
        e_p = pyo.cube_escape_distance(a, b, lo, hi)
        assert (np.isnan(e_c) and np.isnan(e_p)) or np.float32(e_c) == np.float32(e_p)
        assert bool(L.orc_isInsideCube(oracle.vec3(a), oracle.vec3(lo), oracle.vec3(hi))) == pyo.is_inside_cube(a, lo, hi)
