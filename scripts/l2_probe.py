"""Diagnostic driver for PMC passes: the C3 world, every camera of bench.py's path marched once by the stack kernel, one frame per launch
(no parity checks: experiment libraries may write wrong records)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
import bench
W = svo.World.generate(4, 1, 4, 128, 12, build_device=0)
path = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)
out = svo.DeviceBuffer(1920 * 1080 * 32)
prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK, tiles_per_wave=4)
for rep in range(2):
    for cam in path:
        W.trace(cam, prm, (0, 0, 1920, 1080), out.ptr)
        svo.lib.svo_stream_synchronize(None)
print("done")
