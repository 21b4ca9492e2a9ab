import importlib, os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
svo = importlib.import_module("octree-raymarcher_amd")
for depth in (8, 9, 10, 11, 12):
    W = svo.World.generate(4, 1, 4, 128, depth, build_device=0); W.upload(0)
    cam = svo.default_camera(4, 4, 128, 1920, 1080)
    g = W.draw(cam, shadow=True)
    hit = (g["flags"] & 1) != 0
    nan = np.isnan(g["normal"]).any(axis=-1) & hit
    brick = hit & (g["cell"] != 0xFF)
    print("depth %d: hits %.1f%%  NaN normals %.1f%% of hits (brick hits %.1f%% of hits)  shadowed %.1f%% of hits" % (
        depth, 100 * hit.mean(), 100 * nan.sum() / hit.sum(), 100 * brick.sum() / hit.sum(), 100 * ((g["flags"] & 4) != 0).sum() / hit.sum()))
    W.destroy()
