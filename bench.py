#!/usr/bin/env python3
"""bench.py — Mrays/s of the SVO march hot path on MI355X (BASELINE.json metric).

A "step" is one frame: every primary ray of a 1920x1080 image marched through the depth-12
4x1x4-chunk Simplex world (BASELINE.json configs[2]) plus one shadow ray per primary hit, G-buffer
written to HBM.  World pools are resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Frames are pipelined: F consecutive frames (F cameras - here F views of the benchmark camera) are
marched by ONE launch (svo_trace_frames: the kernel's persistent waves run through all F frames'
tiles and drain once per launch instead of once per frame), and S such launches are in flight on S
HIP streams.  K steps = K frames = ceil(K/F) launches (the last one may be short).

N > 1: the image is partitioned into 8-row bands dealt round-robin to the ranks (rank r traces
bands r, r+N, ...; every rank holds the whole world), and each launch ends with ONE RCCL gather of
its F frames' per-rank G-buffer bands (packed losslessly to 8 B/pixel) to rank 0 (total work fixed
-> "scaling": "strong"); the gather of one launch overlaps the trace of the next.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (k_trace_stack) against the
8 TB/s HBM peak using the ALGORITHMIC bytes of the reference algorithm: per ray
4*node_words + 2*brick_cells + 32*chunk_descriptors (restart-from-root counts, measured for this
exact frame by the literal kernel's counters) + 32 B G-buffer record per pixel.  `cpu_baseline`
is the CPU oracle (oracle/, a port of src/Traverse.cpp) timed on this host on the same frame.
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

WORKLOADS = {
    # name: (grid w,h,d, depth, image w,h, shadow)
    "c3_1080p_depth12_4x1x4_shadow": (4, 1, 4, 12, 1920, 1080, True),      # BASELINE configs[2] (the metric's config)
    "c2_1080p_depth10_1chunk": (1, 1, 1, 10, 1920, 1080, False),           # BASELINE configs[1]
    "c3small_1080p_depth10_4x1x4_shadow": (4, 1, 4, 10, 1920, 1080, True),  # quick rehearsal of c3
    "c4_2160p_depth12_4x1x4_shadow": (4, 1, 4, 12, 3840, 2160, True),      # BASELINE configs[3]
    "c5_1080p_depth16_sparse_shadow": (1, 1, 1, 16, 1920, 1080, True),     # BASELINE configs[4]: full depth in a 4-unit band
    "c3_grazing_1080p_depth12_4x1x4_shadow": (4, 1, 4, 12, 1920, 1080, True),   # SURVEY §8d second camera: inside the world, looking along it
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
BAND = 8                       # rows per band == tile height of the stack kernel
STREAMS_FOR_SHARE = {1: 4, 2: 4, 4: 4, 8: 4}       # launches in flight per rank when a frame is split N ways (measured)
FRAMES_PER_LAUNCH = {1: 4, 2: 8, 4: 16, 8: 16}      # consecutive frames marched by one launch (and shipped by one gather)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3_1080p_depth12_4x1x4_shadow", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "literal", "stack"])
    ap.add_argument("--streams", type=int, default=0,
                    help="frames in flight: frame i is issued on HIP stream i %% S into G-buffer i %% S, so the long-ray "
                         "tail of one frame overlaps the bulk of the next (1 = strictly serialized frames)")
    ap.add_argument("--emulate-share", type=int, default=0,
                    help="tuning aid (N=1 only): trace just the bands rank 0 of this many ranks would get, no exchange")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the measured path). gloo = rehearsal of the N>1 control flow on a box with "
                         "fewer GPUs than ranks: ranks share devices and the gather is staged through host memory")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="F consecutive frames are marched by ONE launch (svo_trace_frames; at N > 1 also gathered to rank 0 "
                         "by one collective): the persistent waves drain once per launch.  1 = one launch per frame")
    ap.add_argument("--frames-per-gather", type=int, default=0, help="older name of --frames-per-launch at N > 1")
    ap.add_argument("--no-gather", action="store_true", help="diagnostic (N > 1): trace only, skip the per-frame gather")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=0, help="time the CPU oracle on a centred NxN crop instead of the full frame")
    args = ap.parse_args()

    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with the default, at most 4 frames'
    # kernels are really in flight and 1/N-frame shares cannot hide their long-ray tails.  Must be set before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import numpy as np
    import torch                                   # before the library: one HIP runtime per process
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    # rehearsal aid: run the N > 1 control flow (bands, pack, collective, self-check) with a single rank, e.g. to
    # exercise the RCCL calls on a one-GPU box
    multi = world_size > 1 or bool(os.environ.get("SVO_BENCH_FORCE_DIST"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the SVO march has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world_size, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world_size)

    svo = importlib.import_module("octree-raymarcher_amd")
    gw, gh, gd, depth, iw, ih, shadow = WORKLOADS[args.workload]
    kernel = {"auto": svo.KERNEL_AUTO, "literal": svo.KERNEL_LITERAL, "stack": svo.KERNEL_STACK}[args.kernel]

    # ---- world: generated on the host (deterministic -> identical on every rank), resident in HBM
    t0 = time.time()
    if args.workload.startswith("c5_"):
        world = svo.World.generate(gw, gh, gd, 128, depth, **svo.c5_scene()["generate"])
    else:
        world = svo.World.generate(gw, gh, gd, 128, depth)
    t_gen = time.time() - t0
    t0 = time.time()
    world.upload(local_rank)
    t_up = time.time() - t0
    info = world.info
    cam = svo.default_camera(gw, gd, 128, iw, ih)
    if "grazing" in args.workload:
        cam = svo.make_camera((250.3, 90.0, 5.0), (0.06, -0.04, 1.0), (0.0, 1.0, 0.0), 60.0, iw, ih)       # long, shallow marches
    if args.workload.startswith("c5_"):
        cam = svo.c5_scene()["camera"](iw, ih)                  # hovering over the refined band
    if os.environ.get("SVO_BENCH_EYE_DX"):                      # experiments only: move the eye off the lattice plane
        cam.eye[0] += float(os.environ["SVO_BENCH_EYE_DX"])
    # a 1/N share of the frame is small: with several frames in flight, waves that keep refilling (>= 4 tiles each) beat
    # one wave per tile (+6 % at 1/8 share); no effect on a full 1080p frame, which has more tiles than resident waves
    prm = svo.trace_params(shadow=shadow, kernel=kernel, tiles_per_wave=4)
    # launches in flight: a launch's critical path is its longest ray, so its tail leaves SIMDs idle that the next
    # launches' bulk fills; with N ranks a rank's share of a frame shrinks N-fold, so more frames ride in one launch
    # (both tables measured with --emulate-share)
    S = args.streams if args.streams > 0 else STREAMS_FOR_SHARE.get(max(world_size, args.emulate_share, 1), 16)
    # F consecutive frames form one group: ONE launch marches them (svo_trace_frames: the persistent waves drain once
    # per launch, not once per frame) and, at N > 1, ONE gather ships them.  Group j runs on stream j % S.
    if args.frames_per_launch > 0:
        G = args.frames_per_launch
    elif multi and args.frames_per_gather > 0:
        G = args.frames_per_gather
    else:
        G = FRAMES_PER_LAUNCH.get(max(world_size, args.emulate_share, 1), 4)
    G = max(1, min(G, svo.MAX_FRAMES))
    if args.frames_per_launch <= 0 and args.frames_per_gather <= 0:
        G = max(1, min(G, args.steps // S))     # a short run: fewer frames per launch rather than idle streams
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    stream = torch.cuda.current_stream().cuda_stream

    nb = svo.partition.bands_per_rank(ih, world_size, BAND)     # bands per rank (last ones may be padding)
    rec = 32
    prec = 8
    share = args.emulate_share if (not multi and args.emulate_share > 1) else 0
    if share:
        nb = svo.partition.bands_per_rank(ih, share, BAND)
    if multi or share:
        bufs = [torch.empty((G, nb, BAND, iw, rec), dtype=torch.uint8, device=dev) for _ in range(S)]
    else:
        bufs = [torch.empty((G, ih, iw, rec), dtype=torch.uint8, device=dev) for _ in range(S)]
    if multi:
        # N > 1: this rank's bands (32-B records) are packed to the lossless 8-B form (t, normal code, material, flags)
        # and THAT is gathered to rank 0: a quarter of the xGMI traffic into rank 0's seven links
        pbufs = [torch.empty((G, nb, BAND, iw, prec), dtype=torch.uint8, device=dev) for _ in range(S)]
        gdev = dev if args.backend == "nccl" else torch.device("cpu")
        gathered = [[torch.empty(pbufs[0].shape, dtype=torch.uint8, device=gdev) for _ in range(world_size)] for _ in range(S)] if rank == 0 else [None] * S

    def trace_group(slot, k):
        """One launch: k <= G frames (here: k views of the same camera) into bufs[slot][:k] on stream `slot`."""
        st = streams[slot].cuda_stream
        cams = [cam] * k
        if multi:
            world.trace_rows_frames(cams, prm, rank, world_size, nb, BAND, bufs[slot].data_ptr(), st)
        elif share:
            world.trace_rows_frames(cams, prm, 0, share, nb, BAND, bufs[slot].data_ptr(), st)
        else:
            world.trace_frames(cams, prm, (0, 0, iw, ih), bufs[slot].data_ptr(), st)

    def frame(i, works, events=None, last=False):
        """Frame i joins its group; the group's last frame issues the launch (and, at N > 1, pack + gather to rank 0)."""
        slot, sub = (i // G) % S, i % G
        if not (sub == G - 1 or last):
            return
        k = sub + 1                                 # frames in this group (the region's last group may be short)
        st = streams[slot]
        with torch.cuda.stream(st):
            if multi and works[slot] is not None:
                works[slot].wait()                  # the gather that last read this buffer has finished
                works[slot] = None
            if events is not None:
                events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), k))
                events[-1][0].record(st)
            trace_group(slot, k)
            if events is not None:
                events[-1][1].record(st)
            if multi:
                svo.gbuffer_pack(bufs[slot].data_ptr(), pbufs[slot].data_ptr(), k * nb * BAND * iw, st.cuda_stream)
                if not args.no_gather:
                    src = pbufs[slot][:k]
                    dst = [g[:k] for g in gathered[slot]] if rank == 0 else None
                    if args.backend == "nccl":
                        works[slot] = dist.gather(src, dst, dst=0, async_op=True)
                    else:                           # rehearsal: staged through the host, synchronous
                        st.synchronize()
                        dist.gather(src.cpu(), dst, dst=0)

    def drain(works):
        for k, wk in enumerate(works):
            if wk is not None:
                with torch.cuda.stream(streams[k]):
                    wk.wait()
        for st in streams:
            st.synchronize()

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- untimed: ray count of one frame (primary + shadow), all ranks
    trace_group(0, G)                              # G identical views: every launch of this run has the same shape
    rays_local = world.last_ray_count(streams[0].cuda_stream) // G
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    rays_t = torch.tensor([rays_local], dtype=torch.int64, device=cdev)
    if multi:
        dist.all_reduce(rays_t)
    rays_frame = int(rays_t.item())

    # ---- untimed, rank 0 at N=1: algorithmic bytes of this frame from the reference work counters
    algo_bytes = None
    counters_sum = None
    if not multi and not args.emulate_share:
        cnt = torch.zeros((ih * iw, 4), dtype=torch.int32, device=dev)
        tmp = torch.empty((ih, iw, rec), dtype=torch.uint8, device=dev)
        cprm = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_LITERAL, counters_dev=cnt.data_ptr())
        world.trace(cam, cprm, (0, 0, iw, ih), tmp.data_ptr(), stream)
        torch.cuda.synchronize()
        csum = cnt.to(torch.int64).sum(dim=0).tolist()
        counters_sum = dict(node_words=csum[0], brick_cells=csum[1], chunk_descs=csum[2], tree_steps=csum[3])
        algo_bytes = 4 * csum[0] + 2 * csum[1] + 32 * csum[2] + rec * iw * ih
        # the fast kernel must have produced the same G-buffer as the literal one (cheap self-check, untimed)
        trace_group(0, G)
        torch.cuda.synchronize()
        if not all(torch.equal(tmp, bufs[0][f]) for f in range(G)):
            raise SystemExit("bench.py: stack and literal kernels disagree on the benchmark frame")
        del tmp, cnt

    # ---- warmup
    works = [None] * S
    for i in range(args.warmup):
        frame(i, works, last=(i == args.warmup - 1))
    drain(works)
    works = [None] * S

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    ev = []                                     # (start, end, frames) per launch
    sync_all()
    t_start = time.perf_counter()
    for i in range(args.steps):
        frame(i, works, ev, last=(i == args.steps - 1))
    drain(works)
    sync_all()
    elapsed = time.perf_counter() - t_start
    if multi:
        et = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())
    assert sum(k for _, _, k in ev) == args.steps
    kernel_ms = [a.elapsed_time(b) for a, b, _ in ev]
    kernel_ms_overlapped = sum(kernel_ms) / len(kernel_ms)      # per launch while S launches share the GPU

    # ---- roofline leg (N=1): the kernel's own launch duration.  With launches in flight the per-launch time above
    # measures co-scheduling (S launches share the SIMDs), so the dominant kernel - one launch of G frames, as in the
    # timed region - is also timed back-to-back on ONE stream with HIP events on that stream; rocprofv3 --kernel-trace
    # of `bench.py --streams 1` must agree.
    kernel_ms_avg = kernel_ms_overlapped
    if not multi and not args.emulate_share:
        reps = max(5, min(args.steps // G, 20))
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        st0 = streams[0]
        torch.cuda.synchronize()
        with torch.cuda.stream(st0):
            for a, b in sev:
                a.record(st0)
                world.trace_frames([cam] * G, prm, (0, 0, iw, ih), bufs[0].data_ptr(), st0.cuda_stream)
                b.record(st0)
        st0.synchronize()
        kernel_ms_avg = sum(a.elapsed_time(b) for a, b in sev) / reps

    if rank == 0 and multi and not args.no_gather:
        # de-interleave once (untimed): frame[(k*N + r)*8 + j] = gathered[r][k][j]
        lastf = args.steps - 1
        frame_full = svo.partition.deinterleave([g[lastf % G] for g in gathered[(lastf // G) % S]], ih, BAND)
        assert frame_full.shape[0] == ih
        # untimed self-check: the gathered, de-interleaved frame equals a single-GPU trace of the whole image
        whole = torch.empty((ih, iw, rec), dtype=torch.uint8, device=dev)
        whole_packed = torch.empty((ih, iw, prec), dtype=torch.uint8, device=dev)
        world.trace(cam, prm, (0, 0, iw, ih), whole.data_ptr(), stream)
        svo.gbuffer_pack(whole.data_ptr(), whole_packed.data_ptr(), ih * iw, stream)
        torch.cuda.synchronize()
        if not torch.equal(whole_packed.cpu(), frame_full.cpu()):
            raise SystemExit("bench.py: gathered multi-GPU frame differs from the single-GPU frame")

    result = None
    if rank == 0:
        mrays = rays_frame * args.steps / elapsed / 1e6
        result = {
            "metric": "Mrays/s (primary+shadow) at 1920x1080, depth-12 SVO",
            "value": round(mrays, 3),
            "unit": "Mrays/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 5),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "image": [iw, ih], "grid": [gw, gh, gd], "depth_per_chunk": depth, "chunksize": 128,
                "shadow_rays": bool(shadow), "rays_per_frame": rays_frame,
                "nodes": int(info.total_trees), "bricks": int(info.total_twigs),
                "hbm_pool_bytes": int(info.tree_pool_bytes + info.twig_pool_bytes + info.mask_pool_bytes),
                "kernel": args.kernel, "launches_in_flight": S, "frames_per_launch": G, "frames_in_flight": S * G, "backend": args.backend if multi else None, "gather": bool(multi and not args.no_gather), "partition": "single" if not multi else f"8-row bands round-robin x{world_size} + RCCL gather of 8-B packed G-buffer records",
                "world_generate_s": round(t_gen, 2), "world_upload_s": round(t_up, 2),
            },
        }
        if not multi and not args.emulate_share:
            achieved = algo_bytes * G / (kernel_ms_avg * 1e-3) / 1e9      # one launch marches G frames
            traffic = None
            valu = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived HBM bytes per launch, if profiled
            if os.path.exists(tpath):
                try:
                    tj = json.load(open(tpath))
                    traffic = tj.get(args.workload, {}).get("hbm_bytes_per_frame")      # PMC passes run one frame per launch
                    traffic = traffic * G if traffic else None
                    valu = tj.get(args.workload, {}).get("valu_insts_per_frame")
                except Exception:
                    traffic = None
            result["roofline"] = {
                "bound": "hbm", "kernel": "k_trace_stack" if args.kernel != "literal" else "k_trace_literal",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                # the same bytes over whole-job time: what the overlapped launches deliver together
                "achieved_throughput": round(algo_bytes * args.steps / elapsed / 1e9, 2),
                "frac_throughput": round(algo_bytes * args.steps / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                "algorithmic_bytes_per_launch": int(algo_bytes) * G, "frames_per_launch": G,
                "bytes_per_ray": round(algo_bytes / rays_frame, 2),
                "kernel_ms_avg": round(kernel_ms_avg, 5),
                "kernel_ms_avg_with_frames_in_flight": round(kernel_ms_overlapped, 5),
                "counters": counters_sum,
            }
            if valu:
                # what the march is really bound by: a wave64 VALU instruction holds one of the
                # 1024 SIMDs for 4 cycles; profiled instruction count per launch over this run's time per frame
                simd_cycles = 1024 * 2.4e9 * (elapsed / args.steps)
                result["roofline"]["valu_issue"] = {"insts_per_frame": int(valu), "frac_of_issue_slots": round(valu * 4 / simd_cycles, 4),
                                                    "note": "SQ_INSTS_VALU per one-frame launch from profiles/ (PMC pass) x 4 cycles / (1024 SIMDs x 2.4 GHz x s per frame)"}
            if not args.no_cpu_baseline:
                ob = importlib.import_module("oracle_binding")      # the oracle: checker/baseline only
                n = gw * gh * gd
                O = ob.OracleWorld.from_chunks([world.chunk(i, copy=False) for i in range(n)], gw, gh, gd, 128)
                cores = os.cpu_count() or 1
                if args.cpu_crop:
                    c = args.cpu_crop
                    rect = ((iw - c) // 2, (ih - c) // 2, c, c)
                else:
                    rect = (0, 0, iw, ih)
                t0 = time.perf_counter()
                ref = O.trace_image(cam, rect=rect, params=ob.make_params(shadow=shadow), threads=cores)
                dt = time.perf_counter() - t0
                result["cpu_baseline"] = {
                    "value": round(O.last_rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
                    "sample": f"{rect[2]}x{rect[3]} pixels of the same frame (rect x0={rect[0]}, y0={rect[1]}), "
                              f"{O.last_rays} rays incl. shadow, {dt:.2f} s wall on {cores} threads, oracle/svo_oracle.c -O2",
                }
                # parity of the timed product output against the oracle on that sample
                lastf = args.steps - 1
                got = bufs[(lastf // G) % S][lastf % G].cpu().numpy().view(svo.HIT_DTYPE).reshape(ih, iw)[rect[1]:rect[1] + rect[3], rect[0]:rect[0] + rect[2]]
                same = all(np.array_equal(got[f], ref[f]) for f in ("flags", "material", "chunk", "node", "cell")) and \
                    np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
                result["cpu_baseline"]["parity_with_gpu"] = bool(same)
        print(json.dumps(result), flush=True)

    world.destroy()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
