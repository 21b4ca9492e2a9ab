"""Diagnostic: does hipMalloc stall after a large hipFree on this box?  (svo_device_alloc / svo_device_free are plain hipMalloc / hipFree.)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
GB = 1 << 30
worst = 0.0
for rep in range(12):
    t0 = time.time()
    ptrs = [svo.lib.svo_device_alloc(GB // 2) for _ in range(24)]          # 12 GB in 512 MB pieces, as the builder's brick buffers
    t1 = time.time()
    for p in ptrs: svo.lib.svo_device_free(p)
    t2 = time.time()
    worst = max(worst, t1 - t0, t2 - t1)
    print("rep %2d: alloc 12 GB %.3f s, free %.3f s" % (rep, t1 - t0, t2 - t1), flush=True)
print("worst %.3f s" % worst)
