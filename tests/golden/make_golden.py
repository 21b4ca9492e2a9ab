#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/svo_oracle.c).

These fixtures are SELF-GENERATED regression vectors: the reference has no tests/fixtures for this path and
cannot be compiled in this image (GLM missing), so nothing here comes from the reference itself.  They pin the
oracle (and through it the HIP kernels) against accidental change, and travel to the GPU box.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_binding as ob   # noqa: E402
from helpers import random_rays  # noqa: E402

CASES = {
    # name: world parameters of World::init restated by the oracle
    "c1_depth8_single": dict(w=1, h=1, d=1, depth=8, ccm=(0, 0, 0)),          # BASELINE configs[0] scene
    "grid_2x1x2_depth6": dict(w=2, h=1, d=2, depth=6, ccm=(0, 0, 0)),
    "grid_neg_2x2x2_depth5": dict(w=2, h=2, d=2, depth=5, ccm=(-1, -1, -1)),
}


def main():
    for name, s in CASES.items():
        O = ob.OracleWorld.generate(s["w"], s["h"], s["d"], 128, s["depth"], chunkcoordmin=s["ccm"])
        lo = np.array(s["ccm"], np.float64) * 128
        hi = lo + np.array([s["w"], s["h"], s["d"]]) * 128
        rng = np.random.default_rng(20261004)
        o, d = random_rays(rng, 4096, lo, hi)
        # a few axis-parallel and on-lattice rays (inf / NaN reciprocals)
        extra_o = np.array([[64, 100, 64], [0, 0, 0], [64, 64, 64], [lo[0], 50, lo[2]], [hi[0], 50, hi[2]], [32.5, 127, 32.5]], np.float32)
        extra_d = np.array([[0, -1, 0], [0, 0, 1], [1, 0, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0]], np.float32)
        o = np.concatenate([o, extra_o]); d = np.concatenate([d, extra_d])
        hits, cnt = O.trace_rays(o, d, params=ob.make_params(shadow=True), counters=True)
        tree_sizes = np.array([O.chunk(i)["tree"].size for i in range(O.volume)], np.int64)
        twig_sizes = np.array([O.chunk(i)["twig"].size for i in range(O.volume)], np.int64)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), origins=o, dirs=d, hits=hits, counters=cnt,
                            params=np.array([s["w"], s["h"], s["d"], s["depth"], *s["ccm"]], np.int64),
                            tree_sizes=tree_sizes, twig_sizes=twig_sizes, rays=np.int64(O.last_rays))
        print(name, "rays", O.last_rays, "hits", int((hits["flags"] & 1).sum()), "shadowed", int(((hits["flags"] & 4) != 0).sum()))


if __name__ == "__main__":
    main()
