"""Run the auxiliary HBM-bound kernels (shade, pack, unpack, brick masks via upload) a few times for rocprofv3 --kernel-trace."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
svo = importlib.import_module("octree-raymarcher_amd")
W = svo.World.generate(4, 1, 4, 128, 12, build_device=0); W.upload(0)
w, h = 1920, 1080
cam = svo.default_camera(4, 4, 128, w, h)
g = svo.DeviceBuffer(w * h * 32); p = svo.DeviceBuffer(w * h * 8); g2 = svo.DeviceBuffer(w * h * 32); rgba = svo.DeviceBuffer(w * h * 16)
W.trace(cam, svo.trace_params(shadow=True), (0, 0, w, h), g.ptr)
P = svo.shade_defaults()
for _ in range(20):
    svo.shade(cam, P, (0, 0, w, h), g.ptr, rgba.ptr)
    svo.gbuffer_pack(g.ptr, p.ptr, w * h)
    svo.shade_packed(cam, P, (0, 0, w, h), p.ptr, rgba.ptr)
    svo.gbuffer_unpack(p.ptr, g2.ptr, w * h)
svo.lib.svo_stream_synchronize(None)
print("bricks", W.info.total_twigs)
