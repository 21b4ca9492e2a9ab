import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
svo = importlib.import_module("octree-raymarcher_amd")
W = svo.World.generate(1, 1, 1, 128, 6, build_device=0)      # warm up HIP
t = time.time(); W = svo.World.generate(1, 1, 1, 128, 12, build_device=0, water=False); print("depth-12 chunk, device builder: %.3f s" % (time.time() - t))
i = W.info; print("nodes", i.total_trees, "bricks", i.total_twigs)
