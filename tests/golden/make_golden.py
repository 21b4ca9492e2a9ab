#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the PYTHON restatement of the reference (oracle/svo_oracle_py.py): world generation
(BoundsPyramid, grow(), the water Ocroot::build), the march with its per-hit extras, shadow rays and work counters.

The direction of the check matters: these vectors come from the second, independently written restatement; the C
oracle (oracle/svo_oracle.c) - what every GPU parity test is judged against - has to REPRODUCE them (tests/test_golden.py, CPU
suite), and so have both HIP kernels (GPU suite).  They remain SELF-GENERATED: the reference holds no tests or fixtures for this
path and cannot be compiled in this image (GLM missing), so nothing here comes from the reference itself - parity unpinned.

    python tests/golden/make_golden.py          (about a minute of pure Python)
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import svo_oracle_py as pyo     # noqa: E402
from helpers import random_rays  # noqa: E402

CASES = {
    # name: world parameters of World::init
    "c1_depth8_single": dict(w=1, h=1, d=1, depth=8, ccm=(0, 0, 0)),          # BASELINE configs[0] scene
    "grid_2x1x2_depth6": dict(w=2, h=1, d=2, depth=6, ccm=(0, 0, 0)),
    "grid_neg_2x2x2_depth5": dict(w=2, h=2, d=2, depth=5, ccm=(-1, -1, -1)),
}


def main():
    for name, s in CASES.items():
        P = pyo.World.generate(s["w"], s["h"], s["d"], 128, s["depth"], chunkcoordmin=s["ccm"])
        lo = np.array(s["ccm"], np.float64) * 128
        hi = lo + np.array([s["w"], s["h"], s["d"]]) * 128
        rng = np.random.default_rng(20261004)
        o, d = random_rays(rng, 4096, lo, hi)
        # a few axis-parallel and on-lattice rays (inf / NaN reciprocals)
        extra_o = np.array([[64, 100, 64], [0, 0, 0], [64, 64, 64], [lo[0], 50, lo[2]], [hi[0], 50, hi[2]], [32.5, 127, 32.5]], np.float32)
        extra_d = np.array([[0, -1, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float32)
        o = np.concatenate([o, extra_o]); d = np.concatenate([d, extra_d])
        hits, cnt, rays = pyo.trace_rays(P, o, d, counters=True, shadow=True)
        tree_sizes = np.array([c.trees for c in P.chunk], np.int64)
        twig_sizes = np.array([c.twigs * 64 for c in P.chunk], np.int64)
        # the pools themselves travel as CRC-32s per chunk (index-for-index equality with the C oracle's pools is also
        # tests/test_oracle_cross_check.py's business, on worlds generated there)
        tree_crc = np.array([zlib.crc32(c.tree_array().tobytes()) for c in P.chunk], np.uint32)
        twig_crc = np.array([zlib.crc32(c.twig_array().tobytes()) for c in P.chunk], np.uint32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), origins=o, dirs=d, hits=hits, counters=cnt,
                            params=np.array([s["w"], s["h"], s["d"], s["depth"], *s["ccm"]], np.int64),
                            tree_sizes=tree_sizes, twig_sizes=twig_sizes, tree_crc=tree_crc, twig_crc=twig_crc, rays=np.int64(rays),
                            generator=np.array("oracle/svo_oracle_py.py"))
        print(name, "rays", rays, "hits", int((hits["flags"] & 1).sum()), "shadowed", int(((hits["flags"] & 4) != 0).sum()))


if __name__ == "__main__":
    main()
