"""Scale check (GPU box): a world beyond the 32-bit offsets of the default stack kernel - 10x1x10 chunks of depth 12 by default
(~70 GB of pools in HBM, a wide pool above 4 GiB, more than 2^32 brick cells) - generated (on the device with `device`),
marched by the stack kernel's large-world instantiation (64-bit addresses; kernel_stack.hip.h BIG) and by the literal kernel,
compared with each other on a 1080p frame and with the CPU oracle over the same arrays on a 960x540 frame; then the stack
kernel's throughput on that world next to the C3 world's (4x1x4, the default instantiation), same cameras, same launch shape.

    python scripts/bigworld_check.py [grid_w grid_d [device]]          (10 10 device)
"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
import oracle_binding as ob
from helpers import assert_gbuffer_equal
gw, gd, depth = (int(sys.argv[1]) if len(sys.argv) > 1 else 10), (int(sys.argv[2]) if len(sys.argv) > 2 else 10), 12
on_device = len(sys.argv) > 3 and sys.argv[3] == "device"          # the device builder (noise, mips, grow, water fill as kernels) instead of host threads


def throughput(W, gw, gd, frames=16, launches=6):
    """Mrays/s of svo_trace_frames (primary + shadow, 1080p, `frames` per launch, serialized) over an orbit of cameras."""
    w, h = 1920, 1080
    cams = []
    for f in range(frames):
        a = 2.0 * np.pi * f / frames
        eye = (gw * 64.0 + np.cos(a) * gw * 70.0 + 0.37, 150.0 + 10.0 * np.sin(3 * a), gd * 64.0 + np.sin(a) * gd * 70.0 + 0.41)
        fwd = (gw * 64.0 - eye[0], -110.0, gd * 64.0 - eye[2])
        cams.append(svo.make_camera(eye, fwd, (0, 1, 0), 60.0, w, h))
    # what a ray costs on THIS world: reference steps (tree steps + brick cells tested, the literal kernel's counters) per ray on four of the cameras
    steps = rays4 = 0
    for cam in cams[::max(1, frames // 4)][:4]:
        g, c = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL, counters=True)
        steps += int(c[..., 3].astype(np.int64).sum() + c[..., 1].astype(np.int64).sum()); rays4 += W.last_ray_count()
    per_ray = steps / max(1, rays4)
    out = svo.DeviceBuffer(frames * w * h * 32)
    prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK)
    W.trace_frames(cams, prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)     # warm
    rays = W.last_ray_count()
    t = time.time()
    for _ in range(launches):
        W.trace_frames(cams, prm, (0, 0, w, h), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    dt = time.time() - t
    out.free()
    return rays * launches / dt / 1e6, dt / launches * 1e3, rays, per_ray


t = time.time(); W = svo.World.generate(gw, 1, gd, 128, depth, build_device=0 if on_device else None); tg = time.time() - t
info = W.info
print(f"generated {gw}x1x{gd} depth {depth} in {tg:.1f} s: {info.total_trees/1e6:.0f} M nodes, {info.total_twigs/1e6:.0f} M bricks "
      f"({info.total_twigs*64/2**32:.2f} x 2^32 brick cells)", flush=True)
t = time.time(); W.upload(0); tu = time.time() - t
info = W.info
print(f"upload {tu:.1f} s; HBM pools {(info.tree_pool_bytes+info.twig_pool_bytes+info.mask_pool_bytes)/2**30:.1f} GiB, wide pool {info.wide_pool_bytes/2**30:.2f} GiB "
      f"({info.wide_nodes/1e6:.1f} M wide nodes): {'beyond' if info.wide_pool_bytes >= 2**32 else 'within'} 32-bit byte offsets", flush=True)
assert info.wide_nodes > 0, "no wide trees: the stack kernel cannot run on this world"
# 1. stack (large-world instantiation when the pools ask for it) == literal on a 1080p frame, AUTO picks the stack kernel
cam = svo.default_camera(gw, gd, 128, 1920, 1080)
t = time.time(); a = W.draw(cam, shadow=True, kernel=svo.KERNEL_STACK); ts = time.time() - t
t = time.time(); b = W.draw(cam, shadow=True, kernel=svo.KERNEL_LITERAL); tl = time.time() - t
assert a.tobytes() == b.tobytes() or (assert_gbuffer_equal(a.reshape(-1), b.reshape(-1), "stack vs literal") is None)
c = W.draw(cam, shadow=True, kernel=svo.KERNEL_AUTO)
assert c.tobytes() == a.tobytes()
print(f"1080p frame: stack == literal == auto, {int((a['flags'] & 1).sum())} hits (draw incl. read-back: stack {ts*1e3:.0f} ms, literal {tl*1e3:.0f} ms)", flush=True)
# 2. throughput next to the C3 world's
big = throughput(W, gw, gd)
print(f"stack kernel on this world: {big[0]:.0f} Mrays/s ({big[1]:.2f} ms per 16-frame launch, {big[2]/16e6:.2f} M rays per frame, {big[3]:.1f} reference steps per ray "
      f"= {big[0]*big[3]/1e3:.1f} G reference steps/s)", flush=True)
if not (len(sys.argv) > 4 and sys.argv[4] == "noc3"):
    C3 = svo.World.generate(4, 1, 4, 128, depth, build_device=0)
    c3 = throughput(C3, 4, 4)
    C3.destroy()
    print(f"stack kernel on the C3 world (4x1x4): {c3[0]:.0f} Mrays/s ({c3[1]:.2f} ms per launch, {c3[2]/16e6:.2f} M rays per frame, {c3[3]:.1f} reference steps per ray "
          f"= {c3[0]*c3[3]/1e3:.1f} G reference steps/s); ratio big / C3 = {big[0]/c3[0]:.3f} in rays, {big[0]*big[3]/(c3[0]*c3[3]):.3f} in reference steps", flush=True)
# 3. against the CPU oracle over the same arrays
n = gw * gd
t = time.time()
O = ob.OracleWorld.from_chunks([W.chunk(i, copy=False) for i in range(n)], gw, 1, gd, 128)
print(f"host copies for the oracle fetched in {time.time()-t:.1f} s", flush=True)
w, h = 960, 540
for name, cam in (("far corner", svo.make_camera((gw * 128 - 40.0, 150.0, gd * 128 + 40.0), (-0.4, -0.45, -0.8), (0, 1, 0), 60.0, w, h)),
                  ("default", svo.default_camera(gw, gd, 128, w, h))):
    want = O.trace_image(cam, rect=(0, 0, w, h), params=ob.make_params(shadow=True), threads=os.cpu_count())
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        got = W.draw(cam, shadow=True, kernel=k)
        assert_gbuffer_equal(got.reshape(-1), want.reshape(-1), f"{name} kernel {k}")
    chunks = np.unique(want["chunk"][(want["flags"] & 1) != 0])
    print(f"{name}: {int((want['flags'] & 1).sum())} hits in {len(chunks)} chunks (highest index {chunks.max()}), both kernels bit-identical to the oracle", flush=True)
print("BIGWORLD OK")
