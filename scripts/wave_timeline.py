"""Diagnostic (SVO_STACK_TIMING build): per-wave start/end/iterations of one stack-kernel launch."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 12
F = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # frames per launch
G = int(os.environ.get("SVO_WT_GRID", "4"))              # GxG chunks
W = svo.World.generate(G, 1, G, 128, depth, build_device=0)
cam = svo.default_camera(G, G, 128, 1920, 1080)
if os.environ.get("SVO_WT_ORBIT"):                      # camera f of the 16-camera orbit of scripts/bigworld_check.py / slow_world_probe.py
    a = 2.0 * np.pi * int(os.environ["SVO_WT_ORBIT"]) / 16
    eye = (G * 64.0 + np.cos(a) * G * 70.0 + 0.37, 150.0 + 10.0 * np.sin(3 * a), G * 64.0 + np.sin(a) * G * 70.0 + 0.41)
    cam = svo.make_camera(eye, (G * 64.0 - eye[0], -110.0, G * 64.0 - eye[2]), (0, 1, 0), 60.0, 1920, 1080)
if os.environ.get("SVO_BENCH_EYE_DX"):
    cam.eye[0] += float(os.environ["SVO_BENCH_EYE_DX"])
if os.environ.get("SVO_PATH_CAM"):                      # camera k of bench.py's path
    sys.path.insert(0, ROOT)
    import bench
    cam = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)[int(os.environ["SVO_PATH_CAM"])]
nblk = 256 * 32
cams = [cam] * F
if os.environ.get("SVO_PATH_CAMS"):                     # F consecutive cameras of the path from this one on
    import bench
    path = bench.camera_path(svo, "c3_1080p_depth12_4x1x4_shadow", 4, 4, 1920, 1080)
    cams = [path[(int(os.environ["SVO_PATH_CAMS"]) + i) % len(path)] for i in range(F)]
out = svo.DeviceBuffer(F * 1920 * 1080 * 32)
for rep in range(3):
    cnt = svo.DeviceBuffer.from_numpy(np.zeros((nblk, 24), np.uint32))
    prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK, counters_dev=cnt.ptr)
    W.trace_frames(cams, prm, (0, 0, 1920, 1080), out.ptr)
    svo.lib.svo_stream_synchronize(None)
    c8 = cnt.to_numpy(np.uint32, nblk * 24).reshape(nblk, 24); c = c8[:, :4]; e = c8[:, 4:8]; f = c8[:, 8:12]; h = c8[:, 12:20]; st = c8[:, 20:24]
e = e[c[:, 2] > 0]; f = f[c[:, 2] > 0]; h = h[c[:, 2] > 0]; st = st[c[:, 2] > 0]; c = c[c[:, 2] > 0]
t0 = c[:, 0].min()
stats = st.astype(np.int64).sum(axis=0)
per_wave_asm = st.astype(np.int64)                # [steps, marching lanes summed, stalls, chased] of each wave's asm statements
st = (c[:, 0] - t0).astype(np.int64) * 0.01      # us
en = (c[:, 1] - t0).astype(np.int64) * 0.01
print("waves that ran:", len(c), "kernel span %.1f us" % en.max())
print("start  p50 %.1f p99 %.1f max %.1f" % tuple(np.percentile(st, [50, 99, 100])))
print("end    p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % tuple(np.percentile(en, [10, 50, 90, 99, 100])))
it = c[:, 2].astype(np.int64)
print("iters  p50 %d p99 %d max %d   us/iter p50 %.2f" % (np.percentile(it, 50), np.percentile(it, 99), it.max(), np.median((en - st) / it)))
late = np.argsort(en)[-8:]
for i in late:
    print("  wave end %.1f us start %.1f iters %d rays %d us/iter %.2f  creep runs %d rounds %d busiest-lane steps %d dbg ent %d go %d K>0 %d sure %d" % (en[i], st[i], it[i], c[i, 3], (en[i] - st[i]) / it[i], f[i, 0] & 0xFFFF, f[i, 2], f[i, 1], f[i, 3] & 255, (f[i, 3] >> 8) & 255, (f[i, 3] >> 16) & 255, f[i, 3] >> 24))
    print("      asm steps %d (%.1f marching lanes each; %.2f us per step over the wave's life), chunk-step runs %d, hit-block runs %d, refill rounds %d, tiles %d; step bodies after the tiles ran out: %d" % (
        per_wave_asm[i, 0], per_wave_asm[i, 1] / max(1, per_wave_asm[i, 0]), (en[i] - st[i]) / max(1, per_wave_asm[i, 0]), e[i, 0] & 0xFFFF, e[i, 0] >> 16, e[i, 1] & 0xFFF, (e[i, 1] >> 12) & 0xFF,
        int(h[i, 0]) - int(h[i, 4])))
busy = f[:, 1].astype(np.int64)
print("  ... of which started inside a brick (waves ending last): %s" % (f[late, 0] >> 16).tolist())
print("busiest lane's asm steps per wave: p50 %d p90 %d p99 %d max %d; waves ending last: %s" % (*np.percentile(busy, [50, 90, 99, 100]), busy[late].tolist()))
for t in (200, 400, 600, 800, 1000, 1200, 1500, 2000, 2500):
    print("t=%5d us running waves: %d" % (t, ((st <= t) & (en > t)).sum()))

its = it.sum()
tot_wave_cycles = ((en - st) * 1e-6 * 2.38e9).sum()
ex = e[:, 0].astype(np.int64); ey = e[:, 1].astype(np.int64)
print("wave-iterations total %d" % its)
print("block runs per iteration: world %.3f hit %.3f refill-rounds %.3f tilegen %.4f integral-fix %.3f" % (
    (ex & 0xFFFF).sum() / its, (ex >> 16).sum() / its, (ey & 0xFFF).sum() / its, ((ey >> 12) & 0xFF).sum() / its, (ey >> 20).sum() / its))
print("avg lanes per iteration: tree %.1f twig %.1f world %.1f" % (e[:, 2].astype(np.int64).sum() / its, (e[:, 3] & 0xFFFFF).astype(np.int64).sum() / its, (e[:, 3] >> 20).astype(np.int64).sum() / its))
print("cycles per wave-iteration overall: %.0f" % (tot_wave_cycles / its))
print("creep block: runs per iteration %.4f, rounds per run %.1f" % ((f[:, 0] & 0xFFFF).astype(np.int64).sum() / its, f[:, 2].astype(np.int64).sum() / max(1, (f[:, 0] & 0xFFFF).astype(np.int64).sum())))

hs = h.astype(np.int64).sum(axis=0)
print("step bodies executed %d (%.2f per iteration), marching lanes per step body %.2f of 64; wave cycles per step body %.0f" % (hs[0], hs[0] / its, hs[1] / max(1, hs[0]), tot_wave_cycles / max(1, hs[0])))
b = max(1, hs[4])
print("while tiles remain (bulk): %d step bodies (%.1f %% of all), per step body: marching %.2f (in a brick %.2f), waiting for the chunk step %.2f, for the hit vote %.2f, retired %.2f" % (
    hs[4], 100.0 * hs[4] / max(1, hs[0]), hs[5] / b, hs[7] / b, hs[6] / b, hs[2] / b, hs[3] / b))
print("lane-steps total %d" % hs[1])
print("asm steps %d (%.2f per statement): marching lanes per step %.2f, of them %.2f sit a BRANCH out (%.1f %%), %.2f take a level inside the step" % (
    stats[0], stats[0] / max(1, hs[0]), stats[1] / max(1, stats[0]), stats[2] / max(1, stats[0]), 100.0 * stats[2] / max(1, stats[1]), stats[3] / max(1, stats[0])))
