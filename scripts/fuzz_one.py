"""Debug aid for scripts/fuzz_parity.py: replay its ray generator up to one case and march single rays of it.
    python scripts/fuzz_one.py seed rays_per_case semantics case_index ray_index [ray_index ...]"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
import oracle_binding as ob
from helpers import adversarial_rays, FUZZ_CASES
seed, N, SEM, ci = (int(x) for x in sys.argv[1:5])
idx = [int(x) for x in sys.argv[5:]]
rng = np.random.default_rng(seed)
for k, c in enumerate(FUZZ_CASES):
    lo = np.array(c["ccm"], float) * 128; hi = lo + np.array([c["w"], c["h"], c["d"]]) * 128
    o, d = adversarial_rays(rng, N, lo, hi)
    if k == ci: break
print(c)
W = svo.World.generate(c["w"], c["h"], c["d"], 128, c["depth"], chunkcoordmin=c["ccm"])
n = c["w"] * c["h"] * c["d"]
O = ob.OracleWorld.from_chunks([W.chunk(i, copy=False) for i in range(n)], c["w"], c["h"], c["d"], 128, c["ccm"])
W.upload(0)
for i in idx:
    oo, dd = o[i:i + 1], d[i:i + 1]
    print("ray", i, "o", oo[0].tolist(), [x.hex() for x in oo[0].astype(np.float32).view(np.uint32)[:0]] , "d", dd[0].tolist())
    print("   o bits", [hex(int(x)) for x in oo[0].astype(np.float32).view(np.uint32)], "d bits", [hex(int(x)) for x in dd[0].astype(np.float32).view(np.uint32)])
    want, cnt = O.trace_rays(oo, dd, params=ob.make_params(shadow=True, semantics=SEM), counters=True)
    print("   oracle", want[0], "counters (node words, brick cells, chunk descs, tree steps)", cnt[0].tolist())
    for kern in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        got = W.chunkmarch(oo, dd, shadow=True, kernel=kern, semantics=SEM)
        print("   kernel", kern, got[0])
