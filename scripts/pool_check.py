"""The experimental over-subscribed kernel (SVO_KERNEL_POOL, csrc/kernel_pool.hip.h) against the oracle and against the stack kernel:
parity on ray lists and frames, then throughput on the C3 world (serialized 16-frame launches over the bench's camera path).
    scripts/build_variants.sh pool:-DSVO_WITH_POOL      (the shipped library does not carry the experiment)
    SVO_AMD_LIB=$PWD/octree-raymarcher_amd/build/libsvo_pool.so python scripts/pool_check.py [parity|speed|both]"""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
svo = importlib.import_module("octree-raymarcher_amd")
what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what in ("parity", "both"):
    import oracle_binding as ob
    from helpers import adversarial_rays, assert_gbuffer_equal, creeping_rays, random_rays
    ccm = (-1, 0, -1)
    W = svo.World.generate(2, 1, 2, 128, 8, chunkcoordmin=ccm)
    O = ob.OracleWorld.from_chunks([W.chunk(i) for i in range(4)], 2, 1, 2, 128, ccm)
    W.upload(0)
    lo, hi = (-128.0, 0.0, -128.0), (128.0, 128.0, 128.0)
    rng = np.random.default_rng(7)
    for name, (o, d) in {"tiny": random_rays(rng, 100, lo, hi), "random": random_rays(rng, 200000, lo, hi), "adversarial": adversarial_rays(rng, 100000, lo, hi),
                         "creeping": creeping_rays(rng, 8000, lo, hi, 0.5)}.items():
        for shadow in (False, True):
            want = O.trace_rays(o, d, params=ob.make_params(shadow=shadow), threads=os.cpu_count())
            got = W.chunkmarch(o, d, shadow=shadow, kernel=svo.KERNEL_POOL)
            assert_gbuffer_equal(got, want, f"pool {name} shadow={shadow}")
            assert W.last_ray_count() == O.last_rays, (name, W.last_ray_count(), O.last_rays)
        print("pool ==", "oracle on", name, len(o), "rays", flush=True)
    cam = svo.make_camera((0.3, 150.0, -170.2), (0.0, -0.5, 0.866), (0, 1, 0), 60.0, 640, 360)
    for nm in (0, 1):
        want = O.trace_image(cam, params=ob.make_params(shadow=True, normal_mode=nm), threads=os.cpu_count())
        got = W.draw(cam, shadow=True, kernel=svo.KERNEL_POOL, normal_mode=nm)
        assert_gbuffer_equal(got, want, f"pool frame normal mode {nm}")
    out = svo.DeviceBuffer(3 * 640 * 360 * 32)
    W.trace_frames([cam] * 3, svo.trace_params(shadow=True, kernel=svo.KERNEL_POOL), (0, 0, 640, 360), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    want = O.trace_image(cam, params=ob.make_params(shadow=True), threads=os.cpu_count())
    three = out.to_numpy(svo.HIT_DTYPE, 3 * 640 * 360).reshape(3, 360, 640)
    for f in range(3):
        assert_gbuffer_equal(three[f], want, f"pool frame {f} of 3")
    print("pool == oracle on frames (both normal modes, three per launch)", flush=True)
    W.destroy()
if what in ("speed", "both"):
    sys.path.insert(0, ROOT)
    import bench
    W = svo.World.generate(4, 1, 4, 128, 12, build_device=0)
    path = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)
    w, h = 1920, 1080
    out = svo.DeviceBuffer(16 * w * h * 32)
    ref = None
    for name, k in (("stack", svo.KERNEL_STACK), ("pool", svo.KERNEL_POOL), ("stack", svo.KERNEL_STACK), ("pool", svo.KERNEL_POOL)):
        prm = svo.trace_params(shadow=True, kernel=k, tiles_per_wave=4)
        W.trace_frames(path[:16], prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)
        rays = W.last_ray_count()
        rec = out.to_numpy(np.uint8, 16 * w * h * 32)
        if ref is None: ref = rec.copy()
        else: assert np.array_equal(rec, ref), f"{name}: records differ from the stack kernel's"
        t = time.time()
        for rep in range(4):
            W.trace_frames(path[:16], prm, (0, 0, w, h), out.ptr); W.trace_frames(path[16:], prm, (0, 0, w, h), out.ptr)
        svo.lib.svo_stream_synchronize(None)
        dt = (time.time() - t) / 8
        print(f"{name:6s} {dt*1e3:7.3f} ms per 16-frame launch  ~{rays/dt/1e6:8.0f} Mrays/s (first half's ray count)", flush=True)
    # one frame
    for name, k in (("stack", svo.KERNEL_STACK), ("pool", svo.KERNEL_POOL)):
        prm = svo.trace_params(shadow=True, kernel=k)
        W.trace(path[5], prm, (0, 0, w, h), out.ptr); svo.lib.svo_stream_synchronize(None)
        t = time.time()
        for c in path: W.trace(c, prm, (0, 0, w, h), out.ptr)
        svo.lib.svo_stream_synchronize(None)
        print(f"{name:6s} one frame per launch: {(time.time()-t)/len(path)*1e3:.3f} ms mean over the path", flush=True)
    W.destroy()
print("POOL CHECK DONE")
