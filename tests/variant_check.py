#!/usr/bin/env python3
"""Runs ONE variant build of libsvo_amd (octree-raymarcher_amd/Makefile: `variants`) against the oracle, in its own process - the
binding loads one library per process (SVO_AMD_LIB).  Test infrastructure: tests/test_variants.py starts it, once per variant,
one after the other.

    python tests/variant_check.py <path to libsvo_*.so> march          goldens + the C1 image + adversarial / creeping rays + a world of mixed depths
    python tests/variant_check.py <path to libsvo_hooks.so> hooks      failed wide-tree rebuilds (SVO_TEST_FAIL_WIDE) behind update, edit and shift
    python tests/variant_check.py <path to libsvo_timing.so> timing    the timing build's counters are there and consistent with the records

exit 0 = every check passed; anything else fails the test with this script's output.
"""
import ctypes as C
import glob
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def load(lib_path):
    os.environ["SVO_AMD_LIB"] = lib_path
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    svo = importlib.import_module("octree-raymarcher_amd")
    assert os.path.samefile(svo.LIB_PATH, lib_path)
    import oracle_binding as ob
    return svo, ob


def march(svo, ob):
    from helpers import adversarial_rays, assert_gbuffer_equal, creeping_rays, guard_boundary_rays, random_rays
    kernels = (svo.KERNEL_STACK, svo.KERNEL_LITERAL)
    # 1. the committed goldens
    for path in sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))):
        z = np.load(path, allow_pickle=False)
        w, h, d, depth, cx, cy, cz = [int(v) for v in z["params"]]
        W = svo.World.generate(w, h, d, 128, depth, chunkcoordmin=(cx, cy, cz))
        W.upload(0)
        for k in kernels:
            assert_gbuffer_equal(W.chunkmarch(z["origins"], z["dirs"], shadow=True, kernel=k), z["hits"], f"{os.path.basename(path)}/{k}")
        assert W.last_ray_count() == int(z["rays"])
        W.destroy()
    # 2. BASELINE configs[0]: the 256x256 image of the depth-8 chunk, primary + shadow, against the oracle
    W = svo.World.generate(1, 1, 1, 128, 8)
    O = ob.OracleWorld.from_chunks([W.chunk(0)], 1, 1, 1, 128)
    W.upload(0)
    cam = svo.default_camera(1, 1, 128, 256, 256)
    want = O.trace_image(cam, params=ob.make_params(shadow=True), threads=8)
    assert int((want["flags"] & 1).sum()) > 5000
    for k in kernels:
        assert_gbuffer_equal(W.draw(cam, shadow=True, kernel=k), want, f"C1 image/{k}")
    for face in (svo.NORMAL_FACE,):
        wantf = O.trace_image(cam, params=ob.make_params(shadow=True, normal_mode=face), threads=8)
        assert_gbuffer_equal(W.draw(cam, shadow=True, kernel=svo.KERNEL_STACK, normal_mode=face), wantf, "C1 image, face normals")
    # several frames per launch
    cams = [svo.default_camera(1, 1, 128, 256, 256), svo.default_camera(1, 1, 128, 256, 256)]
    out = svo.DeviceBuffer(2 * 256 * 256 * 32)
    W.trace_frames(cams, svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK), (0, 0, 256, 256), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    both = out.to_numpy(svo.HIT_DTYPE, 2 * 256 * 256)
    assert_gbuffer_equal(both[:256 * 256], want, "frame 0 of 2"); assert_gbuffer_equal(both[256 * 256:], want, "frame 1 of 2")
    out.free()
    W.destroy()
    # 3. adversarial and creeping ray lists on a multi-chunk world with negative coordinates
    ccm = (-1, 0, -1)
    W = svo.World.generate(2, 1, 2, 128, 7, chunkcoordmin=ccm)
    O = ob.OracleWorld.from_chunks([W.chunk(i) for i in range(4)], 2, 1, 2, 128, ccm)
    W.upload(0)
    lo, hi = (-128, 0, -128), (128, 128, 128)
    rng = np.random.default_rng(404)
    lists = {"random": random_rays(rng, 30000, lo, hi), "adversarial": adversarial_rays(rng, 30000, lo, hi),
             "creeping": creeping_rays(rng, 4000, lo, hi, 128.0 / 2 ** 7),
             "guard": guard_boundary_rays(rng, 24000, lo, hi, 4.0 * 128.0 / 2 ** 7)}        # (around the sure-miss test's conditions; brick edge = 4 voxels)
    for name, (o, d) in lists.items():
        want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
        for k in kernels:
            assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"{name}/{k}")
    for name, (o, d) in lists.items():                      # the GLSL twin's march (svo_trace_params.semantics) through the same variant
        want = O.trace_rays(o, d, params=ob.make_params(shadow=True, semantics=1), threads=8)
        for k in kernels:
            assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k, semantics=svo.SEMANTICS_GLSL), want, f"GLSL {name}/{k}")
    o, d = creeping_rays(rng, 3000, lo, hi, 1.0, chunk_faces=True)
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True, caps=(6, 40, 30)), threads=8)
    for k in kernels:
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k, caps=(6, 40, 30)), want, f"chunk-face creep/{k}")
    W.destroy()
    # 4. chunks of different depths in one world (the descent cache must not carry a cell across chunks)
    gen = {depth: svo.World.generate(2, 1, 2, 128, depth) for depth in (3, 8, 5, 6)}
    chunks = [gen[depth].chunk(i) for i, depth in enumerate((3, 8, 5, 6))]
    W = svo.World.create(chunks, 2, 1, 2, 128)
    O = ob.OracleWorld.from_chunks(chunks, 2, 1, 2, 128)
    W.upload(0)
    o, d = random_rays(rng, 30000, (0, 0, 0), (256, 128, 256))
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    for k in kernels:
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"mixed depths/{k}")
    W.destroy()
    print("march: goldens, C1 image (both normal modes, 2 frames per launch), random / adversarial / creeping / guard-boundary lists (CPU and GLSL semantics), mixed depths: all equal to the oracle")


def hooks(svo, ob):
    from helpers import assert_gbuffer_equal, random_rays
    # (a) svo_world_update - the host path of an edit
    O = ob.OracleWorld.generate(2, 1, 1, 128, 6)
    W = svo.World.create([O.chunk(i) for i in range(2)], 2, 1, 1, 128)
    W.upload(0)
    rng = np.random.default_rng(17)
    o, d = random_rays(rng, 6000, (0, 0, 0), (256, 128, 128))
    assert W.info.wide_nodes > 0

    def edit(lo, hi, mat):
        dt, dw = ob.Delta(), ob.Delta()
        ob.lib.orc_build(C.byref(O.w.chunk[0]), ob.vec3(lo), ob.vec3(hi), mat, C.byref(dt), C.byref(dw))
        c = O.chunk(0)
        return W.update(0, c, tree_range=(0, c["tree"].size), twig_range=(0, c["twig"].size // 64), realloc=True)

    os.environ["SVO_TEST_FAIL_WIDE"] = "1"
    rc = edit((20, 60, 20), (50, 90, 50), 5)
    assert rc == svo.OK_LITERAL_ONLY, rc                   # the edit HAS been applied; only the stack kernel is gone
    assert b"wide tree" in svo.lib.svo_last_error()
    del os.environ["SVO_TEST_FAIL_WIDE"]
    assert W.info.wide_nodes == 0 and W.info.wide_pool_bytes == 0
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    assert (want["material"] == 5).sum() > 0
    assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_AUTO), want, "auto after failed rebuild")
    try:
        W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_STACK)
        raise AssertionError("SVO_KERNEL_STACK must be refused without a wide pool")
    except svo.SvoError as e:
        assert e.code == -6
    assert edit((100, 100, 100), (104, 104, 104), 5) == svo.SVO_OK          # a full rebuild of the wide pool
    assert W.info.wide_nodes > 0
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, "after recovery")
    # (b) svo_world_edit_box on the resident world
    os.environ["SVO_TEST_FAIL_WIDE"] = "1"
    lo, hi = (130.0, 40.0, 30.0), (170.0, 100.0, 60.0)
    assert W.edit_box(1, svo.EDIT_BUILD, lo, hi, 5) == svo.OK_LITERAL_ONLY
    del os.environ["SVO_TEST_FAIL_WIDE"]
    dt, dw = ob.Delta(), ob.Delta()
    ob.lib.orc_build(C.byref(O.w.chunk[1]), ob.vec3(lo), ob.vec3(hi), 5, C.byref(dt), C.byref(dw))
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_AUTO), want, "auto after an edit whose wide rebuild failed")
    got = W.chunk(1)
    assert np.array_equal(got["tree"], O.chunk(1)["tree"]) and np.array_equal(got["twig"], O.chunk(1)["twig"])
    assert W.edit_box(1, svo.EDIT_DESTROY, (131.0, 41.0, 31.0), (133.0, 43.0, 33.0)) == svo.SVO_OK
    assert W.info.wide_nodes > 0
    W.destroy()
    # (b2) svo_world_upload: the world is resident, only the stack kernel is missing; the next upload-or-edit brings it back
    W = svo.World.create([O.chunk(i) for i in range(2)], 2, 1, 1, 128)
    os.environ["SVO_TEST_FAIL_WIDE"] = "1"
    W.upload(0)
    del os.environ["SVO_TEST_FAIL_WIDE"]
    assert W.upload_status == svo.OK_LITERAL_ONLY and W.info.wide_nodes == 0
    assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_AUTO), want, "uploaded, literal only")
    assert W.edit_box(0, svo.EDIT_DESTROY, (1.0, 1.0, 1.0), (1.5, 1.5, 1.5)) == svo.SVO_OK and W.info.wide_nodes > 0
    W.destroy()
    # (c) svo_world_shift on a device-resident world: every install's wide rebuild fails - the window still moves as a whole
    W = svo.World.generate(3, 1, 2, 128, 6, build_device=0)
    os.environ["SVO_TEST_FAIL_WIDE"] = "1"
    assert W.shift((1, 0, 0)) == svo.OK_LITERAL_ONLY
    del os.environ["SVO_TEST_FAIL_WIDE"]
    assert tuple(W.info.chunkcoordmin) == (1, 0, 0) and W.info.wide_nodes == 0
    O = ob.OracleWorld.generate(3, 1, 2, 128, 6, chunkcoordmin=(1, 0, 0))
    o, d = random_rays(rng, 12000, (128, 0, 0), (512, 128, 256))
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_AUTO), want, "shifted, literal only")
    assert W.shift((0, 0, 1)) == svo.SVO_OK and W.info.wide_nodes > 0
    O = ob.OracleWorld.generate(3, 1, 2, 128, 6, chunkcoordmin=(1, 0, 1))
    o, d = random_rays(rng, 12000, (128, 0, 128), (512, 128, 384))
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"shifted again/{k}")
    W.destroy()
    # (d) the wide pool is sized from an estimate (wide nodes per BRANCH node of the largest chunk) and grown when that was too little:
    # with the estimate cut to 2 % the pool is grown - and the chunks built so far moved - several times over
    os.environ["SVO_TEST_WIDE_ESTIMATE"] = "0.02"
    W = svo.World.generate(3, 1, 2, 128, 8, build_device=0, chunkcoordmin=(-1, 0, 0))
    del os.environ["SVO_TEST_WIDE_ESTIMATE"]
    assert W.info.wide_nodes > 0
    O = ob.OracleWorld.from_chunks([W.chunk(i) for i in range(6)], 3, 1, 2, 128, (-1, 0, 0))
    o, d = random_rays(rng, 20000, (-128, 0, 0), (256, 128, 256))
    want = O.trace_rays(o, d, params=ob.make_params(shadow=True), threads=8)
    assert (want["flags"] & 1).sum() > 3000
    for k in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=k), want, f"grown wide pool/{k}")
    grown = W.info.wide_pool_bytes
    W.destroy()
    W = svo.World.generate(3, 1, 2, 128, 8, build_device=0, chunkcoordmin=(-1, 0, 0))
    assert W.info.wide_nodes > 0 and abs(W.info.wide_pool_bytes - grown) < grown     # (the estimate alone: the same order of size)
    assert_gbuffer_equal(W.chunkmarch(o, d, shadow=True, kernel=svo.KERNEL_STACK), want, "estimated wide pool")
    W.destroy()
    print("hooks: failed wide rebuilds behind update / edit_box / shift leave the change applied and the literal kernel marching; the next one recovers")


def timing(svo, ob):
    """-DSVO_STACK_TIMING: six uint4 per wave through counters_dev (kernel_stack.hip.h, end of the kernel)."""
    from helpers import assert_gbuffer_equal
    W = svo.World.generate(2, 1, 2, 128, 7)
    O = ob.OracleWorld.from_chunks([W.chunk(i) for i in range(4)], 2, 1, 2, 128)
    W.upload(0)
    cam = svo.default_camera(2, 2, 128, 320, 200)
    w, h = cam.width, cam.height
    want = O.trace_image(cam, params=ob.make_params(shadow=True), threads=8)
    out = svo.DeviceBuffer(w * h * 32)
    nwaves_max = 256 * 32
    ctr = svo.DeviceBuffer.from_numpy(np.zeros(nwaves_max * 6 * 4, np.uint32))
    prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK, counters_dev=ctr.ptr)
    W.trace(cam, prm, (0, 0, w, h), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    assert_gbuffer_equal(out.to_numpy(svo.HIT_DTYPE, w * h), want, "timing build, records")
    c = ctr.to_numpy(np.uint32, nwaves_max * 6 * 4).reshape(nwaves_max, 6, 4)
    used = c[:, 0, 3] > 0                                   # waves that marched rays
    assert used.sum() > 0
    rays = int(c[used, 0, 3].sum())
    assert rays == W.last_ray_count() == O.last_rays, (rays, W.last_ray_count(), O.last_rays)
    wsteps, lsteps = c[used, 3, 0].astype(np.int64), c[used, 3, 1].astype(np.int64)
    assert np.all(lsteps <= 64 * wsteps) and lsteps.sum() > 0
    asm_steps, asm_lanes = c[used, 5, 0].astype(np.int64), c[used, 5, 1].astype(np.int64)
    assert np.all(asm_lanes <= 64 * asm_steps) and asm_lanes.sum() > 0
    print(f"timing: records equal the oracle's; {int(used.sum())} waves, {rays} rays, {asm_lanes.sum() / max(1, asm_steps.sum()):.1f} marching lanes per asm step")
    W.destroy()


if __name__ == "__main__":
    lib_path, what = os.path.abspath(sys.argv[1]), sys.argv[2]
    svo, ob = load(lib_path)
    if svo.device_count() < 1:
        print("no HIP device"); sys.exit(3)
    {"march": march, "hooks": hooks, "timing": timing}[what](svo, ob)
