"""Probe (GPU box): per-camera frame time of the stack kernel on an 8x1x8 world of depth `depth`, next to the largest per-pixel reference work
(the literal kernel's counters) - which frames are slow, and is it a handful of rays?"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 9
gw = gd = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1920, 1080
W = svo.World.generate(gw, 1, gd, 128, depth, build_device=0)
out = svo.DeviceBuffer(w * h * 32)
for f in range(0, 16, 2):
    a = 2.0 * np.pi * f / 16
    eye = (gw * 64.0 + np.cos(a) * gw * 70.0 + 0.37, 150.0 + 10.0 * np.sin(3 * a), gd * 64.0 + np.sin(a) * gd * 70.0 + 0.41)
    fwd = (gw * 64.0 - eye[0], -110.0, gd * 64.0 - eye[2])
    cam = svo.make_camera(eye, fwd, (0, 1, 0), 60.0, w, h)
    g, c = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL, counters=True)
    work = c[..., 3].astype(np.int64) + c[..., 1].astype(np.int64)
    prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK)
    W.trace(cam, prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)
    t = time.time()
    for _ in range(3):
        W.trace(cam, prm, (0, 0, w, h), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    ms = (time.time() - t) / 3 * 1e3
    srt = np.sort(work.reshape(-1))
    print(f"camera {f:2d}: stack kernel {ms:7.2f} ms per frame; reference steps per pixel: mean {work.mean():.1f}, p99.9 {srt[int(0.999 * srt.size)]}, the five largest {srt[-5:].tolist()}", flush=True)
