"""Timing probe (GPU box): svo_world_shift and svo_world_edit_box on the resident C3 world (4x1x4 chunks of depth 12)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
W = svo.World.generate(4, 1, 4, 128, 12, build_device=0)
for rep in range(2):
    for axis, sign in ((0, 1), (2, 1), (0, -1), (2, -1)):
        off = [0, 0, 0]; off[axis] = sign
        t = time.time(); rc = W.shift(off); svo.lib.svo_stream_synchronize(None); dt = time.time() - t
        print(f"shift axis {axis} sign {sign:+d}: {dt*1e3:.1f} ms (status {rc})", flush=True)
for k, (lo, hi) in enumerate((((40, 60, 40), (44, 64, 44)), ((100, 50, 100), (110, 70, 110)), ((10, 20, 10), (60, 90, 60)))):
    for op, name in ((svo.EDIT_BUILD, "build"), (svo.EDIT_DESTROY, "destroy")):
        t = time.time(); rc = W.edit_box(W.index(0, 0, 0), op, lo, hi, 5); svo.lib.svo_stream_synchronize(None); dt = time.time() - t
        print(f"edit_box {name} {lo}..{hi}: {dt*1e3:.2f} ms (status {rc})", flush=True)
