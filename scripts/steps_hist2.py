import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
depth = int(sys.argv[1])
W = svo.World.generate(4, 1, 4, 128, depth); W.upload(0)
cam = svo.default_camera(4, 4, 128, 1920, 1080)
g0, c0 = W.draw(cam, shadow=False, kernel=svo.KERNEL_LITERAL, counters=True)
g1, c1 = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL, counters=True)
c0 = c0.astype(np.int64); c1 = c1.astype(np.int64)
s1 = c1[..., 3] + c1[..., 1] + c1[..., 2]
idx = np.argsort(s1.ravel())[-12:]
for i in idx:
    y, x = divmod(int(i), 1920)
    print(f"pixel ({y},{x}) primary: tree {c0[y,x,3]} brick {c0[y,x,1]} chunk {c0[y,x,2]} | +shadow: tree {c1[y,x,3]-c0[y,x,3]} brick {c1[y,x,1]-c0[y,x,1]} chunk {c1[y,x,2]-c0[y,x,2]} | t={g1['t'][y,x]:.3f} flags={g1['flags'][y,x]} cell={g1['cell'][y,x]}")
