"""Device-side world generation (csrc/builder.hip) must produce pools bit-identical to the host
builder (which tests/test_scene_parity.py ties to the oracle's restatement of grow / BoundsPyramid / build)."""
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
    H.destroy(); D.destroy()


def test_device_builder_c3_world_and_speed(svo):
    """The benchmark world (4x1x4 chunks, depth 12): identical pools; report both generation times."""
    t0 = time.time(); H = svo.World.generate(4, 1, 4, 128, 12); th = time.time() - t0
    t0 = time.time(); D = svo.World.generate(4, 1, 4, 128, 12, build_device=0); td = time.time() - t0
    assert H.info.total_trees == D.info.total_trees and H.info.total_twigs == D.info.total_twigs
    for i in range(16):
        a, b = H.chunk(i, copy=False), D.chunk(i, copy=False)
        assert np.array_equal(a["tree"], b["tree"]) and np.array_equal(a["twig"], b["twig"])
    print(f"\nC3 world generation: host threads {th:.2f} s, device builder {td:.2f} s")
    H.destroy(); D.destroy()
