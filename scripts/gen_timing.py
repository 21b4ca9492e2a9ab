import importlib, time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
svo = importlib.import_module("octree-raymarcher_amd")
for rep in range(2):
    t = time.time(); W = svo.World.generate(4, 1, 4, 128, 12, build_device=0); print("generate+upload", round(time.time() - t, 3)); W.destroy()
