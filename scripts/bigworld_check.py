"""One-off scale check (GPU box): a world whose brick pool has more than 2^32 cells (8x1x8 chunks, depth 12: ~20 GB of
pools in HBM), generated, uploaded, marched by both kernels and compared with the CPU oracle over the same arrays."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
import oracle_binding as ob
from helpers import assert_gbuffer_equal
gw, gd, depth = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), (int(sys.argv[2]) if len(sys.argv) > 2 else 8), 12
on_device = len(sys.argv) > 3 and sys.argv[3] == "device"          # the device builder (noise, mips, grow, water fill as kernels) instead of host threads
t = time.time(); W = svo.World.generate(gw, 1, gd, 128, depth, build_device=0 if on_device else None); tg = time.time() - t
info = W.info
print(f"generated {gw}x1x{gd} depth {depth} in {tg:.1f} s: {info.total_trees/1e6:.0f} M nodes, {info.total_twigs/1e6:.0f} M bricks "
      f"({info.total_twigs*64/2**32:.2f} x 2^32 brick cells)", flush=True)
t = time.time(); W.upload(0); tu = time.time() - t
info = W.info
print(f"upload {tu:.1f} s; HBM pools {(info.tree_pool_bytes+info.twig_pool_bytes+info.mask_pool_bytes)/2**30:.1f} GiB", flush=True)
n = gw * gd
O = ob.OracleWorld.from_chunks([W.chunk(i, copy=False) for i in range(n)], gw, 1, gd, 128)
w, h = 960, 540
for name, cam in (("far corner", svo.make_camera((gw * 128 - 40.0, 150.0, gd * 128 + 40.0), (-0.4, -0.45, -0.8), (0, 1, 0), 60.0, w, h)),
                  ("default", svo.default_camera(gw, gd, 128, w, h))):
    want = O.trace_image(cam, rect=(0, 0, w, h), params=ob.make_params(shadow=True), threads=os.cpu_count())
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        got = W.draw(cam, shadow=True, kernel=k)
        assert_gbuffer_equal(got.reshape(-1), want.reshape(-1), f"{name} kernel {k}")
    chunks = np.unique(want["chunk"][(want["flags"] & 1) != 0])
    print(f"{name}: {int((want['flags'] & 1).sum())} hits in {len(chunks)} chunks (highest index {chunks.max()}), both kernels bit-identical to the oracle", flush=True)
print("BIGWORLD OK")
