"""Per-wave diagnostics of the experimental pool kernel (counters_dev): iterations, services, idle spins, in-place fallbacks, statements, ticks."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
import bench
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W = svo.World.generate(4, 1, 4, 128, depth, build_device=0)
path = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
F = int(sys.argv[4]) if len(sys.argv) > 4 else 1
cam = path[5]; cam.width, cam.height = w, h
out = svo.DeviceBuffer(F * w * h * 32)
nw = 256 * 24
cnt = svo.DeviceBuffer.from_numpy(np.zeros(nw * 8, np.uint32))
prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_POOL, counters_dev=cnt.ptr)
W.trace_frames([cam] * F, prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)
t = time.time(); W.trace_frames([cam] * F, prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None); dt = time.time() - t
prs = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK)
W.trace_frames([cam] * F, prs, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)
t = time.time(); W.trace_frames([cam] * F, prs, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None); dts = time.time() - t
print(f"stack kernel, same launch: {dts*1e3:.2f} ms")
c = cnt.to_numpy(np.uint32, nw * 8).reshape(nw, 8).astype(np.int64)
c = c[c[:, 0] > 0]
lanes = c[:, 5] >> 12; c[:, 5] &= 0xFFF
print(f"marching lanes at a statement's start: {lanes.sum() / max(1, (c[:, 0] - c[:, 4]).sum()):.1f} of 64 (per iteration that marched)")
print(f"{w}x{h} x {F} frames: {dt*1e3:.2f} ms, rays {W.last_ray_count()}, waves {len(c)}")
names = ["iters", "serve_world", "serve_hit", "serve_tile", "idle_spins", "inplace", "statements", "ticks(10ns)"]
for i, n in enumerate(names):
    print(f"  {n:12s} sum {c[:, i].sum():12d}  p50 {int(np.percentile(c[:, i], 50)):9d}  p99 {int(np.percentile(c[:, i], 99)):9d}  max {c[:, i].max():9d}")
