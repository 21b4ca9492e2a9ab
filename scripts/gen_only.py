"""Diagnostic: time svo_world_generate (device builder) for the C3 world, cold and warm (SVO_BUILD_TIMING=1 prints the phases)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for r in range(reps):
    t0 = time.time()
    W = svo.World.generate(4, 1, 4, 128, depth, build_device=0)
    t1 = time.time()
    print("generate %d: %.3f s   trees %d twigs %d" % (r, t1 - t0, W.info.total_trees, W.info.total_twigs), flush=True)
    del W
