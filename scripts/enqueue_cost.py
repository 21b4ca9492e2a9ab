"""Diagnostic: host-side enqueue cost per frame (trace_rows [+ gbuffer_pack]) vs device time, rank-0 share of 1/8 frame."""
import importlib, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
share = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W = svo.World.generate(4, 1, 4, 128, 12); W.upload(0)
iw, ih, BAND = 1920, 1080, 8
cam = svo.default_camera(4, 4, 128, iw, ih)
nb = svo.partition.bands_per_rank(ih, share, BAND)
S = 16
streams = [torch.cuda.Stream() for _ in range(S)]
bufs = [torch.empty((nb, BAND, iw, 32), dtype=torch.uint8, device="cuda") for _ in range(S)]
pb = [torch.empty((nb, BAND, iw, 8), dtype=torch.uint8, device="cuda") for _ in range(S)]
prm = svo.trace_params(shadow=True, kernel=svo.KERNEL_STACK)
for pack in (False, True):
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(400):
            st = streams[i % S].cuda_stream
            W.trace_rows(cam, prm, 0, share, nb, BAND, bufs[i % S].data_ptr(), st)
            if pack: svo.gbuffer_pack(bufs[i % S].data_ptr(), pb[i % S].data_ptr(), nb * BAND * iw, st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    print(f"share 1/{share} pack={pack}: enqueue {1e6*(t1-t0)/400:.1f} us/frame, total {1e6*(t2-t0)/400:.1f} us/frame")
