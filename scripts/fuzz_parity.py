"""One-off adversarial fuzz: stack + literal kernels vs the CPU oracle on millions of rays (GPU box)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
import oracle_binding as ob
from helpers import assert_gbuffer_equal
T = os.cpu_count() or 8
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400000


def rays(lo, hi, n):
    ext = hi - lo
    o = lo + rng.random((n, 3)) * ext
    k = n // 8
    # origins snapped to the voxel lattice / chunk faces, outside points, tiny / zero direction components
    o[:k] = np.round(o[:k] * 4) / 4
    o[k:2 * k] = lo + np.round(rng.random((k, 3)) * (ext / 128)) * 128
    o[2 * k:3 * k] = lo - 0.3 * ext + rng.random((k, 3)) * 1.6 * ext
    d = rng.normal(size=(n, 3))
    d[3 * k:4 * k, rng.integers(0, 3)] *= 1e-4
    d[4 * k:5 * k, rng.integers(0, 3)] = 0.0
    z = 5 * k + np.arange(k); ax = rng.integers(0, 3, k)
    d[z] = 0.0; d[z, ax] = rng.choice([-1.0, 1.0], k)                   # axis-parallel
    d[6 * k:7 * k] = np.sign(d[6 * k:7 * k]) * np.array([1.0, 1.0, 0.0]) + 1e-7 * rng.normal(size=(k, 3))   # near the 45 deg shadow direction
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 0, d / np.where(nrm > 0, nrm, 1), d)
    return o.astype(np.float32), d.astype(np.float32)


cases = [dict(w=2, h=1, d=2, depth=8, ccm=(0, 0, 0)), dict(w=2, h=2, d=2, depth=6, ccm=(-1, -1, -1)), dict(w=1, h=1, d=1, depth=11, ccm=(3, 0, -2)),
         dict(w=3, h=1, d=1, depth=7, ccm=(-2, 0, 5))]
for c in cases:
    W = svo.World.generate(c["w"], c["h"], c["d"], 128, c["depth"], chunkcoordmin=c["ccm"])
    n = c["w"] * c["h"] * c["d"]
    O = ob.OracleWorld.from_chunks([W.chunk(i, copy=False) for i in range(n)], c["w"], c["h"], c["d"], 128, c["ccm"])
    W.upload(0)
    lo = np.array(c["ccm"], float) * 128; hi = lo + np.array([c["w"], c["h"], c["d"]]) * 128
    o, d = rays(lo, hi, N)
    for light in ((1.0, -1.0, 0.0), (0.2, -0.9, 0.4)):
        t = time.time()
        want = O.trace_rays(o, d, params=ob.make_params(shadow=True, light_dir=light), threads=T)
        for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
            got = W.chunkmarch(o, d, shadow=True, kernel=k, light_dir=light)
            assert_gbuffer_equal(got, want, f"{c} kernel {k} light {light}")
        print(c, light, "ok: rays", O.last_rays, "hits", int((want["flags"] & 1).sum()), "err flags", int((want["flags"] & 0x8000).sum()), "%.1fs" % (time.time() - t), flush=True)
    W.destroy()
print("FUZZ OK")
