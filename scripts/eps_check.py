"""Soak: random and adversarial rays at extreme EPS values, both kernels against the CPU oracle (GPU box)."""
import importlib, os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
import oracle_binding as ob
from helpers import assert_gbuffer_equal, adversarial_rays, random_rays, creeping_rays
W = svo.World.generate(2, 1, 2, 128, 7)
O = ob.OracleWorld.from_chunks([W.chunk(i) for i in range(4)], 2, 1, 2, 128)
W.upload(0)
lo = np.zeros(3); hi = np.array([256.0, 128.0, 256.0])
rng = np.random.default_rng(77)
sets = {"random": random_rays(rng, 20000, lo, hi), "adversarial": adversarial_rays(rng, 40000, lo, hi)}
for eps in (2.0, 0.3, 2.0 ** -20, 1e-6, 2.0 ** -10):
    for name, (o, d) in sets.items():
        prm = ob.make_params(shadow=True, eps=eps, caps=(64, 4000, 64))
        want = O.trace_rays(o, d, params=prm, threads=16)
        for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
            got = W.chunkmarch(o, d, shadow=True, kernel=k, eps=eps, caps=(64, 4000, 64))
            assert_gbuffer_equal(got, want, f"eps {eps} {name} kernel {k}")
        print("eps", eps, name, "ok: hits", int((want["flags"] & 1).sum()), "err", int(((want["flags"] >> 15) & 1).sum()), flush=True)
print("EPS OK")
