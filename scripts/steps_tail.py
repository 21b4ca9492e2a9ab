"""Diagnostic: tail of the per-pixel step distribution (primary + shadow, literal counters) and where it sits in the image."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
W = svo.World.generate(4, 1, 4, 128, 12); W.upload(0)
cam = svo.default_camera(4, 4, 128, 1920, 1080)
if os.environ.get("SVO_BENCH_EYE_DX"):
    cam.eye[0] += float(os.environ["SVO_BENCH_EYE_DX"])
if os.environ.get("SVO_PATH_CAM"):                      # camera k of bench.py's path
    sys.path.insert(0, ROOT)
    import bench
    cam = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)[int(os.environ["SVO_PATH_CAM"])]
g, c = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL, counters=True)
c = c.astype(np.int64)
steps = (c[..., 3] + c[..., 1] + c[..., 2]).reshape(1080, 1920)
n = steps.size
for thr in (64, 100, 150, 200, 300, 400, 600, 800, 1000, 1200):
    m = steps > thr
    tiles = m.reshape(135, 8, 240, 8).any(axis=(1, 3))
    print(f">{thr:5d} steps: {m.sum():8d} pixels ({100.0*m.sum()/n:.3f} %), in {tiles.sum():6d} of 32400 tiles, rows {np.where(m.any(axis=1))[0].min() if m.any() else -1}..{np.where(m.any(axis=1))[0].max() if m.any() else -1}")
rows = steps.reshape(135, 8, 1920).sum(axis=(1, 2))
print("work per 8-row band (first 135):", " ".join(str(int(r // 1000)) for r in rows))
top = np.argsort(steps.ravel())[-12:][::-1]
for k in top:
    y, x = divmod(int(k), 1920)
    print(f"pixel ({x},{y}): tree steps {c[k // 1920 * 1920 + k % 1920, 3] if c.ndim == 2 else c[y, x, 3]} brick cells {c.reshape(-1, 4)[k, 1]} chunk descs {c.reshape(-1, 4)[k, 2]} node words {c.reshape(-1, 4)[k, 0]} flags {int(g.reshape(-1)[k]['flags'])} t {float(g.reshape(-1)[k]['t']):.3f}")
cols = steps.max(axis=0)
print("columns with a pixel > 1000 steps:", np.where(cols > 1000)[0].tolist()[:40])
