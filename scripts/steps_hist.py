"""Per-pixel work distribution of a bench frame, from the literal kernel's reference counters (GPU)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dx = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
W = svo.World.generate(4, 1, 4, 128, depth); W.upload(0)
cam = svo.default_camera(4, 4, 128, 1920, 1080); cam.eye[0] += dx
t = time.time(); g, c = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL, counters=True); print("literal+copy %.3fs" % (time.time() - t))
c = c.astype(np.int64)
steps = c[..., 3] + c[..., 1] + c[..., 2]
for name, a in (("tree_steps", c[..., 3]), ("brick", c[..., 1]), ("chunk", c[..., 2]), ("nodes", c[..., 0]), ("iters", steps)):
    a = a.ravel()
    print(f"{name:10s} sum {a.sum():12d} mean {a.mean():7.1f} p50 {int(np.percentile(a,50)):5d} p99 {int(np.percentile(a,99)):5d} p99.9 {int(np.percentile(a,99.9)):6d} p99.99 {int(np.percentile(a,99.99)):6d} max {a.max()}")
idx = np.argsort(steps.ravel())[-8:]
print([(int(i // 1920), int(i % 1920), int(steps.ravel()[i])) for i in idx])
for thr in (500, 1000, 2000, 5000, 20000):
    print("pixels with >%d iters: %d" % (thr, (steps > thr).sum()))
