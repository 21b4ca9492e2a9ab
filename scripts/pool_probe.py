"""Probe: which call shape makes the pool kernel slow (diagnostic)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
import bench
W = svo.World.generate(4, 1, 4, 128, 12, build_device=0)
path = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)
w, h = 1920, 1080
out = svo.DeviceBuffer(4 * w * h * 32)
def timed(label, fn):
    fn(); svo.lib.svo_stream_synchronize(None)
    t = time.time(); fn(); svo.lib.svo_stream_synchronize(None)
    print(f"{label}: {(time.time()-t)*1e3:.2f} ms", flush=True)
for kname, k in (("stack", svo.KERNEL_STACK), ("pool", svo.KERNEL_POOL)):
    timed(f"{kname} 1 frame, tiles_per_wave 0", lambda: W.trace(path[5], svo.trace_params(shadow=True, kernel=k), (0, 0, w, h), out.ptr))
    timed(f"{kname} 1 frame, tiles_per_wave 4", lambda: W.trace(path[5], svo.trace_params(shadow=True, kernel=k, tiles_per_wave=4), (0, 0, w, h), out.ptr))
    timed(f"{kname} 4 frames in one call", lambda: W.trace_frames(path[:4], svo.trace_params(shadow=True, kernel=k, tiles_per_wave=4), (0, 0, w, h), out.ptr))
    timed(f"{kname} camera 0 (on the seam)", lambda: W.trace(path[0], svo.trace_params(shadow=True, kernel=k), (0, 0, w, h), out.ptr))
    timed(f"{kname} camera 31 (grazing)", lambda: W.trace(path[31], svo.trace_params(shadow=True, kernel=k), (0, 0, w, h), out.ptr))
