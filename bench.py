#!/usr/bin/env python3
"""bench.py — Mrays/s of the SVO march hot path on MI355X (BASELINE.json metric).

A "step" is one frame: every primary ray of a 1920x1080 image marched through the depth-12
4x1x4-chunk Simplex world (BASELINE.json configs[2]) plus one shadow ray per primary hit, G-buffer
written to HBM.  World pools are resident in HBM before the timed region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Without a launcher (`WORLD_SIZE` unset) and N > 1 the script starts its own N rank processes (multiprocessing
"spawn", before anything touches HIP) and exits non-zero if any of them fails.

The frames follow a deterministic CAMERA PATH (`--camera-path orbit`, default): frame i is seen by camera
i mod P of P = 32 distinct views - the SURVEY.md §8d camera (eye on the chunk seam x = 256) first, then an orbit
whose eyes sit off the voxel lattice (x offset 0.31 + ..., the positions where rays creep along lattice planes), and
the "grazing" camera inside the world last.  `value` = all rays of the K timed frames / wall time.  The per-camera
spread (each camera timed alone, serialized) and the identical-view figure of round 1 (`--camera-path fixed`: every
frame the same camera) are reported beside it as diagnostics.

Frames are pipelined: F consecutive frames (F cameras of the path) are marched by ONE launch (svo_trace_frames: the
kernel's persistent waves run through all F frames' tiles and drain once per launch instead of once per frame), and
S such launches are in flight on S HIP streams.  K steps = K frames = ceil(K/F) launches (the last one may be short).

N > 1: the image is partitioned into 8-row bands dealt round-robin to the ranks (rank r traces
bands r, r+N, ...; every rank holds the whole world), and each launch ends with ONE RCCL gather of
its F frames' per-rank G-buffer bands (packed losslessly to 8 B/pixel) to rank 0 (total work fixed
-> "scaling": "strong"); the gather of one launch overlaps the trace of the next.

Rank 0 prints one JSON line.  `roofline` prices the dominant kernel (k_trace_stack) against the
8 TB/s HBM peak using the ALGORITHMIC bytes of the reference algorithm: per ray
4*node_words + 2*brick_cells + 32*chunk_descriptors (restart-from-root counts, measured for these
exact frames by the literal kernel's counters) + 32 B G-buffer record per pixel.  `cpu_baseline`
is the CPU oracle (oracle/, a port of src/Traverse.cpp) timed on this host on one frame of the path.
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import statistics
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
    "ref_default_1080p_depth8_4x4x4_shadow": (4, 4, 4, 8, 1920, 1080, True),    # the reference's own default scene: world.init(4, 4, 4, 128), TREE_MAX_DEPTH 8 (src/Main.cpp:80, src/World.cpp:10)
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_STREAM_GBS = 6300.0        # ... and what a streaming copy achieves (6.29 TB/s measured)
BAND = 8                       # rows per band == tile height of the stack kernel
# A launch ends with a tail (its longest rays, each wave alone on its SIMD); the frames of one launch, behind one set of cursors,
# pay it once, and the next launch's bulk runs under it.  The N > 1 rows come from one-GPU emulation of a rank's share
# (--emulate-share) and are unmeasured on real multi-GPU hardware.
# (re-measured at the end of round 3 with scripts/sweep_share.sh: 1/4 share 8 x 16 -> 6 675 against 6 205 Mrays/s with 4 x 16, 1/8 share 6 283 against 6 139)
STREAMS_FOR_SHARE = {1: 4, 2: 4, 4: 8, 8: 8}       # launches in flight per rank when a frame is split N ways
FRAMES_PER_LAUNCH = {1: 8, 2: 16, 4: 16, 8: 16}     # consecutive frames marched by one launch (and shipped by one gather)
# N = 1, per workload (launches in flight, frames per launch).  The depth-12 4x1x4 worlds: two launches of up to 16 frames give the
# throughput of four of eight (6.9 Grays/s either way, 5.9 at 20 steps) and spread a launch's one drain over twice the frames
# (serialized launch: 0.566 against 0.640 ms per frame).  C2's and C5's frames are short (0.2 - 0.3 ms): they want four launches in
# flight (C2 with two: 6.5 against 11.1 Grays/s) and lose 2 - 7 % with 16 frames per launch.
LAUNCH_SHAPE = {"c3_1080p_depth12_4x1x4_shadow": (2, 16), "c4_2160p_depth12_4x1x4_shadow": (2, 16),
                "ref_default_1080p_depth8_4x4x4_shadow": (2, 16)}     # (7.9 - 8.4 against 7.1 - 7.3 Grays/s with 4 x 8, gpurun_out/s2_refdefault_shapes.txt)
PATH_CAMERAS = 32


def camera_path(svo, workload, gw, gd, iw, ih, count=PATH_CAMERAS):
    """The deterministic camera path of the run: camera 0 is the workload's base view, cameras 1.. an
    orbit around it with eyes off the voxel lattice and a swinging view direction; C3-family workloads end with the
    grazing camera.  Plain float64 arithmetic, rounded to float32 by make_camera: the same path on every host."""
    if workload.startswith("c5_"):
        base_eye, base_fwd = (64.3, 12.7, 96.5), (0.0, -0.8, 0.6)          # svo.c5_scene(): hovering over the refined band
        span, rise, reach = 1.3, 0.6, 2.0
    elif "grazing" in workload:
        base_eye, base_fwd = (250.3, 90.0, 5.0), (0.06, -0.04, 1.0)
        span, rise, reach = 30.0, 4.0, 15.0
    else:
        base_eye, base_fwd = (gw * 128 * 0.5, 150.0, -40.0), (0.0, -0.5, 0.866)
        span, rise, reach = 0.29 * gw * 128 * 0.5, 5.0, 21.0
    cams = [svo.make_camera(base_eye, base_fwd, (0.0, 1.0, 0.0), 60.0, iw, ih)]
    for k in range(1, count):
        th = 2.0 * math.pi * k / count
        eye = (base_eye[0] + 0.31 + span * math.sin(th), base_eye[1] + 0.07 + rise * math.sin(2.0 * th),
               base_eye[2] + 0.13 + reach * (1.0 - math.cos(th)))
        yaw = 0.22 * math.sin(th)
        fx, fy, fz = base_fwd
        fwd = (fx * math.cos(yaw) + fz * math.sin(yaw), fy + 0.08 * (math.cos(2.0 * th) - 1.0), -fx * math.sin(yaw) + fz * math.cos(yaw))
        cams.append(svo.make_camera(eye, fwd, (0.0, 1.0, 0.0), 60.0, iw, ih))
    if workload.startswith(("c3_1080p", "c3small", "c4_")):
        cams[-1] = svo.make_camera((250.3, 90.0, 5.0), (0.06, -0.04, 1.0), (0.0, 1.0, 0.0), 60.0, iw, ih)   # grazing: long, shallow marches
    return cams


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3_1080p_depth12_4x1x4_shadow", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel", default="auto", choices=["auto", "literal", "stack"])
    ap.add_argument("--camera-path", default="orbit", choices=["orbit", "fixed"],
                    help="orbit = 32 distinct views incl. off-lattice eyes and the grazing camera (the metric); fixed = every frame "
                         "is the SURVEY camera (round-1 figure: identical rays in flight share every cache line)")
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
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the per-camera spread and the identical-view leg")
    ap.add_argument("--cpu-crop", type=int, default=0, help="time the CPU oracle on a centred NxN crop instead of the full frame")
    ap.add_argument("--spawn-selftest", action="store_true",
                    help="no GPU, no tracing: run the N-rank control flow (self-spawn, rendezvous, ranks_seen, band partition, "
                         "gather to rank 0, de-interleave) on synthetic band buffers over gloo; used by the CPU test suite")
    ap.add_argument("--selftest-fail-rank", type=int, default=-1, help="--spawn-selftest: this rank exits with an error (exit-code propagation test)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# self-spawn: `python bench.py --gpus N` without a launcher
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_entry(rank, world_size, port, argv):
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world_size), "LOCAL_WORLD_SIZE": str(world_size),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SVO_BENCH_SPAWNED": "1"})
    run(parse_args(argv))


def spawn_ranks(args, argv):
    """Start args.gpus fresh rank processes (spawn: nothing of this process's state - in particular no HIP context - is
    inherited; never exec) and wait; any failing rank fails the run."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_entry, args=(r, args.gpus, port, argv), name=f"bench-rank{r}") for r in range(args.gpus)]
    for p in procs:
        p.start()
    failed = None
    while failed is None and any(p.is_alive() for p in procs):
        for p in procs:
            p.join(timeout=0.2)
            if p.exitcode not in (None, 0):
                failed = p
                break
    if failed is None:
        for p in procs:
            p.join()
        bad = [p for p in procs if p.exitcode != 0]
        failed = bad[0] if bad else None
    if failed is not None:
        for p in procs:                         # the others would wait in a collective forever: stop exactly the processes started here
            if p.is_alive():
                p.terminate()
        for p in procs:
            p.join(timeout=10)
        raise SystemExit(f"bench.py: rank process {failed.name} exited with code {failed.exitcode}")


def spawn_selftest(args, rank, world_size):
    """The N-rank control flow without a GPU: rendezvous, ranks_seen, band partition, gather, de-interleave."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.join(ROOT, "octree-raymarcher_amd"))
    part = importlib.import_module("partition")                 # pure host logic; the package itself needs the HIP library
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    if rank == args.selftest_fail_rank:
        raise SystemExit(3)
    seen = torch.ones(1, dtype=torch.int64)
    dist.all_reduce(seen)
    ih, iw, prec = 116, 40, 8                                   # ragged: 14.5 bands
    nb = part.bands_per_rank(ih, world_size, BAND)
    mine = torch.zeros((nb, BAND, iw, prec), dtype=torch.uint8)
    for k in range(nb):
        for j, y in enumerate(part.band_rows(rank, world_size, k, BAND)):
            mine[k, j] = (y * 7 + rank) % 251                  # a pattern the de-interleaved frame can be checked against
    gathered = [torch.empty_like(mine) for _ in range(world_size)] if rank == 0 else None
    dist.gather(mine, gathered, dst=0)
    if rank == 0:
        frame = part.deinterleave(gathered, ih, BAND)
        ok = frame.shape[0] == ih
        for y in range(ih):
            ok = ok and bool((frame[y] == (y * 7 + (y // BAND) % world_size) % 251).all())
        if not ok:
            raise SystemExit("bench.py --spawn-selftest: de-interleaved frame is wrong")
        print(json.dumps({"spawn_selftest": True, "ranks_seen": int(seen.item()), "n_gpus": world_size,
                          "launcher": "self-spawned" if os.environ.get("SVO_BENCH_SPAWNED") else "external",
                          "frame_rows": ih, "bands_per_rank": nb}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main(argv=None):
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args, list(sys.argv[1:] if argv is None else argv))
    return run(args)


def run(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus} (or without a launcher)")
    if args.spawn_selftest:
        return spawn_selftest(args, rank, world_size)

    # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): with the default, at most 4 frames'
    # kernels are really in flight and 1/N-frame shares cannot hide their long-ray tails.  Must be set before HIP starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
    import numpy as np
    import torch                                   # before the library: one HIP runtime per process
    import torch.distributed as dist

    # rehearsal aid: run the N > 1 control flow (bands, pack, collective, self-check) with a single rank, e.g. to
    # exercise the RCCL calls on a one-GPU box
    multi = world_size > 1 or bool(os.environ.get("SVO_BENCH_FORCE_DIST"))
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

    # ---- world: generated by every rank for itself, on its own GPU (noise, mips, grow() and the water fill as kernels,
    # pools left in HBM: deterministic -> identical on every rank).  Generated twice: the first call of a process also
    # pays the HIP runtime's first-use costs (code object load, first allocations), the second is the builder alone.
    local_ranks = int(os.environ.get("LOCAL_WORLD_SIZE", str(world_size)))
    gen_threads = max(1, (os.cpu_count() or 1) // max(1, local_ranks))
    gen_kw = svo.c5_scene()["generate"] if args.workload.startswith("c5_") else {}
    t0 = time.time()
    world = svo.World.generate(gw, gh, gd, 128, depth, threads=gen_threads, build_device=local_rank, **gen_kw)
    t_gen = time.time() - t0
    # the second generation: the library hands the first world's pools on to it (include/svo.h svo_device_cache_trim: destroy
    # keeps the large device buffers), so it is the builder alone - until round 4 it also paid hipFree + hipMalloc of 12 GB,
    # which on this pool stalled for ~4 s about once in ten cycles (scripts/alloc_probe.py) and was papered over by a retry here
    world.destroy()
    t0 = time.time()
    world = svo.World.generate(gw, gh, gd, 128, depth, threads=gen_threads, build_device=local_rank, **gen_kw)
    t_gen_warm = time.time() - t0
    gen_stalled = t_gen_warm > max(2.0 * t_gen, 0.5)        # slower than the process's FIRST call: not the builder's doing (reported, not retried)
    t0 = time.time()
    world.upload(local_rank)                       # already resident where it was built: a no-op
    t_up = time.time() - t0
    info = world.info
    path = camera_path(svo, args.workload, gw, gd, iw, ih)
    if os.environ.get("SVO_BENCH_EYE_DX"):                      # experiments only: move the base eye off the lattice plane
        path[0].eye[0] += float(os.environ["SVO_BENCH_EYE_DX"])
    if args.camera_path == "fixed":
        path = [path[0]]
    P = len(path)
    # a 1/N share of the frame is small: with several frames in flight, waves that keep refilling (>= 4 tiles each) beat
    # one wave per tile (+6 % at 1/8 share); no effect on a full 1080p frame, which has more tiles than resident waves
    prm = svo.trace_params(shadow=shadow, kernel=kernel, tiles_per_wave=4)        # (launches_in_flight is set below, once S is known)
    # launches in flight: a launch's critical path is its longest ray, so its tail leaves SIMDs idle that the next
    # launches' bulk fills; with N ranks a rank's share of a frame shrinks N-fold, so more frames ride in one launch
    # (both tables measured with --emulate-share on ONE GPU; unmeasured on real multi-GPU hardware)
    one_gpu_whole_frame = max(world_size, args.emulate_share, 1) == 1
    shape = LAUNCH_SHAPE.get(args.workload) if one_gpu_whole_frame else None
    S = args.streams if args.streams > 0 else (shape[0] if shape else STREAMS_FOR_SHARE.get(max(world_size, args.emulate_share, 1), 16))
    # F consecutive frames form one group: ONE launch marches them (svo_trace_frames: the persistent waves drain once
    # per launch, not once per frame) and, at N > 1, ONE gather ships them.  Group j runs on stream j % S.
    if args.frames_per_launch > 0:
        G = args.frames_per_launch
    elif multi and args.frames_per_gather > 0:
        G = args.frames_per_gather
    else:
        G = shape[1] if shape else FRAMES_PER_LAUNCH.get(max(world_size, args.emulate_share, 1), 4)
    G = max(1, min(G, svo.MAX_FRAMES))
    if args.frames_per_launch <= 0 and args.frames_per_gather <= 0:
        # a short run: fewer frames per launch rather than idle streams - except for the whole frame on one GPU once the run is longer than one
        # launch (the driver's --steps 20: a launch of 16 and one of 4 - 6 158 - 6 258 Mrays/s against 6 050 - 6 183 for two of 10, and the
        # launch the run is made of is the one the 200-step figure is made of; profiles/r04_driver_shapes_sure_miss.txt).  The roofline leg below
        # times the launch sizes of the timed region, the short one included.
        if not (one_gpu_whole_frame and args.steps > G):
            G = max(1, min(G, args.steps // S))
    if args.kernel == "literal":
        G = 1                                   # the literal kernel is one launch per frame (svo_trace_last_ray_count reports one frame)
    prm.launches_in_flight = S                  # the timed launches share the wave slots (svo_trace_params.launches_in_flight)
    prm_alone = svo.trace_params(shadow=shadow, kernel=kernel, tiles_per_wave=4)       # serialized legs: one launch owns the device
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
    cdev = dev if args.backend == "nccl" else torch.device("cpu")

    def cams_of(first, k, fixed=None):
        return [fixed] * k if fixed is not None else [path[(first + f) % P] for f in range(k)]

    def trace_group(slot, cams):
        """One launch: len(cams) <= G frames into bufs[slot][:k] on stream `slot`."""
        st = streams[slot].cuda_stream
        if multi:
            world.trace_rows_frames(cams, prm, rank, world_size, nb, BAND, bufs[slot].data_ptr(), st)
        elif share:
            world.trace_rows_frames(cams, prm, 0, share, nb, BAND, bufs[slot].data_ptr(), st)
        else:
            world.trace_frames(cams, prm, (0, 0, iw, ih), bufs[slot].data_ptr(), st)

    def frame(i, works, events=None, last=False, gather=True, fixed=None):
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
            trace_group(slot, cams_of(i - sub, k, fixed))
            if events is not None:
                events[-1][1].record(st)
            if multi:
                svo.gbuffer_pack(bufs[slot].data_ptr(), pbufs[slot].data_ptr(), k * nb * BAND * iw, st.cuda_stream)
                if gather:
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

    def timed_region(steps, gather=True, fixed=None, events=None):
        """Exactly `steps` frames, barrier + synchronize on both sides; returns max-over-ranks seconds."""
        works = [None] * S
        sync_all()
        t_start = time.perf_counter()
        for i in range(steps):
            frame(i, works, events, last=(i == steps - 1), gather=gather, fixed=fixed)
        drain(works)
        sync_all()
        dt = time.perf_counter() - t_start
        if multi:
            et = torch.tensor([dt], dtype=torch.float64, device=cdev)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            dt = float(et.item())
        return dt

    # ---- untimed preamble.  N = 1: every camera of the path is traced once by the literal kernel (its reference work
    # counters price the algorithmic bytes; its launch also counts the frame's rays), then the path is traced group by
    # group by the timed kernel - launches of the same shape as the timed ones - and every frame must equal the literal
    # kernel's G-buffer.  N > 1: this rank's share of every camera is traced once and the ray counts are summed over ranks.
    rays_local = []
    algo_cam = None
    counters_sum = None
    exempt_records = 0                          # records outside the stack-vs-literal comparison (SVO_ERR_FLAG in either kernel's output)
    if not multi and not args.emulate_share:
        algo_cam = []
        counters_sum = dict(node_words=0, brick_cells=0, chunk_descs=0, tree_steps=0)
        cnt = torch.zeros((ih * iw, 4), dtype=torch.int32, device=dev)
        tmp = torch.empty((G, ih, iw, rec), dtype=torch.uint8, device=dev)
        cprm = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_LITERAL, counters_dev=cnt.data_ptr())
        for first in range(0, P, G):
            group = list(range(first, min(first + G, P)))
            for f, ci in enumerate(group):
                world.trace(path[ci], cprm, (0, 0, iw, ih), tmp[f].data_ptr(), stream)
                rays_local.append(world.last_ray_count(stream))
                csum = cnt.to(torch.int64).sum(dim=0).tolist()
                for key, v in zip(list(counters_sum), csum):
                    counters_sum[key] += v
                algo_cam.append(4 * csum[0] + 2 * csum[1] + 32 * csum[2] + rec * iw * ih)
            trace_group(0, [path[ci] for ci in group])          # the fast kernel must produce the literal kernel's G-buffers
            torch.cuda.synchronize()
            for f, ci in enumerate(group):
                if not torch.equal(tmp[f], bufs[0][f]):
                    # records flagged SVO_ERR_FLAG (a ray given up after 2^22 steps of the KERNEL's own counting) are outside the
                    # cross-kernel contract: the two kernels count their steps differently near that bound (include/svo.h)
                    a16, b16 = tmp[f].view(torch.int16).reshape(ih * iw, 16), bufs[0][f].view(torch.int16).reshape(ih * iw, 16)
                    err = (a16[:, 9] < 0) | (b16[:, 9] < 0)                  # flags halfword, bit 15
                    if not torch.equal(a16[~err], b16[~err]):
                        raise SystemExit(f"bench.py: stack and literal kernels disagree on path camera {ci}")
                    # ... but they are counted, and a kernel that gives up on more than a handful of rays - or the stack kernel
                    # giving up where the literal kernel does not - is a defect, not an exemption
                    n_err = int(err.sum().item())
                    n_stack_only = int(((b16[:, 9] < 0) & ~(a16[:, 9] < 0)).sum().item())
                    exempt_records += n_err
                    if n_err > max(8, int(1e-5 * ih * iw)) or n_stack_only:
                        raise SystemExit(f"bench.py: {n_err} records of path camera {ci} carry SVO_ERR_FLAG ({n_stack_only} from the stack kernel alone)")
        del tmp, cnt
    else:
        for c in path:
            trace_group(0, [c])
            rays_local.append(world.last_ray_count(streams[0].cuda_stream))
    rays_t = torch.tensor(rays_local, dtype=torch.int64, device=cdev)
    seen_t = torch.ones(1, dtype=torch.int64, device=cdev)
    if multi:
        dist.all_reduce(rays_t)
        dist.all_reduce(seen_t)                     # how many ranks the collective backend really connects
    rays_cam = [int(v) for v in rays_t.tolist()]
    ranks_seen = int(seen_t.item())
    devices = [torch.cuda.current_device()]
    if multi:
        dl = [torch.zeros(1, dtype=torch.int64, device=cdev) for _ in range(world_size)]
        dist.all_gather(dl, torch.tensor([torch.cuda.current_device()], dtype=torch.int64, device=cdev))
        devices = [int(d.item()) for d in dl]

    def rays_of(steps, fixed=False):
        return steps * rays_cam[0] if fixed else sum(rays_cam[i % P] for i in range(steps))

    # ---- warmup
    works = [None] * S
    for i in range(args.warmup):
        frame(i, works, last=(i == args.warmup - 1))
    drain(works)

    # ---- timed region: exactly K steps, barrier + synchronize on both sides
    ev = []                                     # (start, end, frames) per launch
    elapsed = timed_region(args.steps, gather=not args.no_gather, events=ev)
    assert sum(k for _, _, k in ev) == args.steps
    kernel_ms = [a.elapsed_time(b) for a, b, _ in ev]
    kernel_ms_overlapped = sum(kernel_ms) / len(kernel_ms)      # per launch while S launches share the GPU
    lastf = args.steps - 1
    last_slot, last_sub = (lastf // G) % S, lastf % G
    last_out = bufs[last_slot][:last_sub + 1].clone()            # the last timed launch's G-buffers (parity sample below)
    if rank == 0 and multi and not args.no_gather:
        last_gathered = [g[last_sub].clone() for g in gathered[last_slot]]

    # ---- diagnostics, untimed by the contract
    diag = {}
    if multi and not args.no_gather:
        dt = timed_region(args.steps, gather=False)             # the same frames without the exchange step
        diag["trace_only_mrays"] = round(rays_of(args.steps) / dt / 1e6, 3)
    if not multi and not share and not args.no_diagnostics:
        if P > 1:
            dt = timed_region(args.steps, fixed=path[0])        # round-1 figure: every frame in flight is the same view
            diag["identical_view_mrays"] = round(rays_of(args.steps, fixed=True) / dt / 1e6, 3)
        # each camera alone: one serialized launch of G frames of that camera (second of two)
        per_cam = []
        st0 = streams[0]
        for ci, c in enumerate(path):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(st0):
                world.trace_frames([c] * G, prm_alone, (0, 0, iw, ih), bufs[0].data_ptr(), st0.cuda_stream)
                a.record(st0)
                world.trace_frames([c] * G, prm_alone, (0, 0, iw, ih), bufs[0].data_ptr(), st0.cuda_stream)
                b.record(st0)
            st0.synchronize()
            per_cam.append(rays_cam[ci] * G / (a.elapsed_time(b) * 1e-3) / 1e6)
        diag["per_camera_serialized_mrays"] = {
            "min": round(min(per_cam), 1), "median": round(statistics.median(per_cam), 1), "max": round(max(per_cam), 1),
            "camera0_on_lattice": round(per_cam[0], 1), "slowest_camera": int(per_cam.index(min(per_cam))),
            "all": [round(v) for v in per_cam],
            "how": f"each of the {P} path cameras alone: one launch of {G} frames of that camera, back to back on one stream"}

    # ---- single-frame latency (N=1): the reference's caller issues ONE World::draw per displayed frame (src/Main.cpp:190-222).
    # One svo_trace per frame, strictly serialized, over the camera path; plain, and with the tiles handed out longest-first
    # by the previous frame's per-tile step maxima (svo_trace_params.tile_cost_dev -> svo_tile_order -> tile_order_dev).
    # The ordered figure includes the sort.  Records must be byte-identical either way.
    single = None
    if not multi and not share and not args.no_diagnostics and args.kernel != "literal" and world.info.wide_nodes > 0:
        st0 = streams[0]
        ntl = ((iw + 7) // 8) * ((ih + 7) // 8)
        cost = torch.zeros(ntl * 2, dtype=torch.int32, device=dev)
        order = torch.zeros(ntl, dtype=torch.int32, device=dev)
        prm1 = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_STACK)                 # tiles_per_wave 0: as many waves as the device keeps resident
        prm1o = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_STACK, tile_cost_dev=cost.data_ptr(), tile_order_dev=order.data_ptr())
        prm1c = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_STACK, tile_cost_dev=cost.data_ptr())
        ob = bufs[0][0]

        def lap(cams, ordered):
            ms = []
            with torch.cuda.stream(st0):
                world.trace(cams[-1], prm1c if ordered else prm1, (0, 0, iw, ih), ob.data_ptr(), st0.cuda_stream)      # primes caches / the cost record
                for c in cams:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(st0)
                    if ordered:
                        world.tile_order(cost.data_ptr(), order.data_ptr(), ntl, st0.cuda_stream)
                        world.trace(c, prm1o, (0, 0, iw, ih), ob.data_ptr(), st0.cuda_stream)
                    else:
                        world.trace(c, prm1, (0, 0, iw, ih), ob.data_ptr(), st0.cuda_stream)
                    b.record(st0)
                    ms.append((a, b))
            st0.synchronize()
            return [a.elapsed_time(b) for a, b in ms]

        plain_path, ord_path = lap(path, False), lap(path, True)
        ord_frame = ob.clone()
        lap(path[-1:], False)
        same = bool(torch.equal(ord_frame, ob))
        plain0, ord0 = lap([path[0]] * 8, False), lap([path[0]] * 8, True)
        # the same frames under the GLSL twin's semantics (svo_trace_params.semantics = SVO_SEMANTICS_GLSL: shaders/Chunkmarch.glsl, the
        # march the reference renders with - EPS 1/4096, caps 256 / 512 / 64, the BIGEPS guard: no ray creeps by EPS along a lattice
        # plane).  Another march, other records: reported next to the headline semantics (src/Traverse.cpp), never instead of it.
        prm1 = svo.trace_params(shadow=shadow, kernel=svo.KERNEL_STACK, semantics=svo.SEMANTICS_GLSL)
        glsl_path = lap(path, False)
        glsl_rays = []
        for c in path[:4]:
            world.trace(c, prm1, (0, 0, iw, ih), ob.data_ptr(), st0.cuda_stream)
            glsl_rays.append(world.last_ray_count(st0.cuda_stream))
        single = {
            "how": "one svo_trace per frame, serialized on one stream, every camera of the path in turn (mean) and the SURVEY camera repeated; "
                   "ordered = tiles handed out longest-first by the previous frame's tile costs, svo_tile_order's device sort included",
            "plain_ms": {"mean": round(statistics.mean(plain_path), 4), "max": round(max(plain_path), 4), "survey_camera": round(statistics.median(plain0), 4)},
            "ordered_ms": {"mean": round(statistics.mean(ord_path), 4), "max": round(max(ord_path), 4), "survey_camera": round(statistics.median(ord0), 4)},
            "records_identical": same,
            "glsl_semantics_ms": {"mean": round(statistics.mean(glsl_path), 4), "max": round(max(glsl_path), 4),
                                  "rays_per_frame_first_4_cameras": glsl_rays,
                                  "note": "svo_trace_params.semantics = SVO_SEMANTICS_GLSL (the shader twin's march, parity-tested against the oracle's "
                                          "restatement of it in tests/test_gpu_glsl.py); the headline metric stays on the CPU march's semantics"}}
        if not same:
            raise SystemExit("bench.py: tile ordering changed the G-buffer - refusing to report")

    # ---- roofline leg (N=1): the kernel's own launch duration.  With launches in flight the per-launch time above
    # measures co-scheduling (S launches share the SIMDs), so the dominant kernel - one launch of G consecutive path
    # frames, as in the timed region - is also timed back-to-back on ONE stream with HIP events on that stream;
    # rocprofv3 --kernel-trace of `bench.py --streams 1` must agree.
    kernel_ms_avg = kernel_ms_overlapped
    roof_bytes = None
    if not multi and not args.emulate_share:
        # the launch sizes of the timed region, in its order and proportion (a run that is no multiple of G ends with a shorter launch, whose
        # drain weighs more: it is timed too, as often as it occurs), repeated until at least five launches are timed
        sizes = [k for _, _, k in ev]
        if len(set(sizes)) == 1:
            reps = max(5, min(args.steps // G, 2 * P // G if P > 1 else 20))
            sizes = sizes[:1] * reps
        else:
            sizes = sizes * max(1, -(-5 // len(sizes)))
            reps = len(sizes)
        sev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
        st0 = streams[0]
        torch.cuda.synchronize()
        roof_bytes = 0
        first = 0
        with torch.cuda.stream(st0):
            for (a, b), k in zip(sev, sizes):
                a.record(st0)
                world.trace_frames(cams_of(first, k), prm_alone, (0, 0, iw, ih), bufs[0].data_ptr(), st0.cuda_stream)
                b.record(st0)
                roof_bytes += sum(algo_cam[(first + f) % P] for f in range(k))
                first += k
        st0.synchronize()
        kernel_ms_sum = sum(a.elapsed_time(b) for a, b in sev)
        kernel_ms_avg = kernel_ms_sum / reps
        roof_sizes = sizes

    if rank == 0 and multi and not args.no_gather:
        # de-interleave once (untimed): frame[(k*N + r)*8 + j] = gathered[r][k][j]
        frame_full = svo.partition.deinterleave(last_gathered, ih, BAND)
        assert frame_full.shape[0] == ih
        # untimed self-check (mandatory): the gathered, de-interleaved frame equals a single-GPU trace of the whole image
        whole = torch.empty((ih, iw, rec), dtype=torch.uint8, device=dev)
        whole_packed = torch.empty((ih, iw, prec), dtype=torch.uint8, device=dev)
        world.trace(path[lastf % P], prm, (0, 0, iw, ih), whole.data_ptr(), stream)
        svo.gbuffer_pack(whole.data_ptr(), whole_packed.data_ptr(), ih * iw, stream)
        torch.cuda.synchronize()
        if not torch.equal(whole_packed.cpu(), frame_full.cpu()):
            raise SystemExit("bench.py: gathered multi-GPU frame differs from the single-GPU frame")

    if rank == 0:
        rays_total = rays_of(args.steps)
        mrays = rays_total / elapsed / 1e6
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
                "shadow_rays": bool(shadow), "rays_per_frame": round(rays_total / args.steps, 1),
                "camera_path": {"kind": args.camera_path, "cameras": P, "rays_per_frame_min": min(rays_cam), "rays_per_frame_max": max(rays_cam),
                                "note": "frame i is seen by camera i mod P; camera 0 = SURVEY §8d view (eye on the chunk seam), the others off the voxel lattice, the last one the grazing view"},
                "nodes": int(info.total_trees), "bricks": int(info.total_twigs),
                "hbm_pool_bytes": int(info.tree_pool_bytes + info.twig_pool_bytes + info.mask_pool_bytes + info.wide_pool_bytes),
                "wide_tree": {"nodes": int(info.wide_nodes), "pool_bytes": int(info.wide_pool_bytes)},
                "kernel": args.kernel, "launches_in_flight": S, "frames_per_launch": G, "frames_in_flight": S * G,
                "backend": args.backend if multi else None, "gather": bool(multi and not args.no_gather),
                "partition": "single" if not multi else f"8-row bands round-robin x{world_size} + RCCL gather of 8-B packed G-buffer records",
                "ranks_seen": ranks_seen, "devices": devices,
                "launcher": "self-spawned" if os.environ.get("SVO_BENCH_SPAWNED") else ("external" if "WORLD_SIZE" in os.environ else "single process"),
                "world_generate_s": round(t_gen, 3), "world_generate_warm_s": round(t_gen_warm, 3), "world_generate_allocator_stall_seen": gen_stalled, "stack_vs_literal_exempt_records": exempt_records,
                "world_generate": "on the rank's GPU (noise, mips, grow, water fill), pools left in HBM; _s = first call of the process, _warm_s = second", "world_generate_threads": gen_threads, "world_upload_s": round(t_up, 3),
            },
        }
        if single:
            result["single_frame_ms"] = single
        if diag:
            result["diagnostics"] = diag
        if not multi and not args.emulate_share:
            achieved = roof_bytes / (kernel_ms_sum * 1e-3) / 1e9         # algorithmic bytes of the timed launches / their durations
            algo_total = sum(algo_cam[i % P] for i in range(args.steps))
            prof = {}
            tpath = os.path.join(ROOT, "profiles", "traffic.json")     # PMC-derived figures of an earlier profiled run, if any
            if os.path.exists(tpath):
                try:
                    prof = json.load(open(tpath)).get(args.workload, {})
                except Exception:
                    prof = {}
            traffic = prof.get("fabric_bytes_per_frame", prof.get("hbm_bytes_per_frame"))
            result["roofline"] = {
                "bound": "hbm", "kernel": "k_trace_stack" if args.kernel != "literal" else "k_trace_literal",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "frac_of_achievable_6300": round(achieved / HBM_STREAM_GBS, 5),
                # PMC figures are NOT measured in this run: replayed from the committed profile of the same workload
                "traffic": traffic * sum(roof_sizes) / reps if traffic else None,       # (per launch of the average size timed, like achieved)
                "traffic_kind": "L2-miss fabric bytes per launch (FETCH_SIZE + WRITE_SIZE; Infinity-Cache hits are counted, so an upper bound on HBM bytes)",
                "traffic_source": prof.get("source", "none") + " (replayed, not measured in this run)",
                "l2_hit_rate": prof.get("l2_hit_rate"),
                # what the memory system really moves against the chip's peak: the profiled fabric bytes per frame over THIS run's time per frame
                "hbm_frac_measured": round(traffic / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                # the same bytes over whole-job time: what the overlapped launches deliver together
                "achieved_throughput": round(algo_total / elapsed / 1e9, 2),
                "frac_throughput": round(algo_total / elapsed / 1e9 / HBM_PEAK_GBS, 5),
                "algorithmic_bytes_per_launch": int(roof_bytes // reps), "frames_per_launch": G, "launch_sizes_timed": roof_sizes,
                "bytes_per_ray": round(sum(algo_cam) / sum(rays_cam), 2),
                "kernel_ms_avg": round(kernel_ms_avg, 5), "kernel_launches_timed": reps,
                "kernel_ms_avg_with_frames_in_flight": round(kernel_ms_overlapped, 5),
                "note_serialized_vs_ms_per_step": "kernel_ms_avg / frames_per_launch is the SERIALIZED launch's time per frame (its drain - the launch's longest rays - "
                                                  "included once per launch); ms_per_step is lower because launches in flight hide each other's drain",
                "counters_all_path_cameras": counters_sum,
            }
            if prof.get("issue"):
                # instruction issue: ONE figure with its spread, measured by scripts/prof_round3.sh + scripts/microbench/valu_issue
                # on the serialized launch of the reported shape and replayed here; "this_run" prices the profiled instruction count per frame
                # against this run's time per frame at the nominal 2.4 GHz shader clock
                iss = prof["issue"]
                simd_cycles = 1024 * 2.4e9 * (elapsed / args.steps)
                result["roofline"]["issue"] = {
                    "instructions_per_frame": int(iss["instructions_per_frame"]), "valu_per_frame": int(iss["valu_per_frame"]), "salu_per_frame": int(iss["salu_per_frame"]),
                    "cycles_per_instruction": iss["cycles_per_instruction_weighted"], "cycles_per_instruction_spread": iss["spread_p10_p90_over_simds"],
                    "issue_utilisation_profiled_launch": iss["issue_utilisation"], "issue_utilisation_spread": iss["issue_utilisation_spread"],
                    "issue_utilisation_direct_range": iss.get("issue_utilisation_direct_range"),
                    "issue_utilisation_this_run_at_2.4GHz": round(iss["instructions_per_frame"] * iss["cycles_per_instruction_weighted"] / simd_cycles, 4),
                    "wave_wait_fraction": iss["wave_wait_fraction"], "wave_active_fraction": iss["wave_active_fraction"], "lane_utilisation_valu": iss["lane_utilisation_valu"],
                    "source": prof.get("source", "") + " (replayed)",
                    "note": "issue_utilisation_profiled_launch is a MODEL (PMC instruction counts x the asm step's static mix priced by scripts/microbench/valu_issue); "
                            "issue_utilisation_direct_range = the same counts priced between the cheapest class and the weighted mix: quote the range (DESIGN.md §5); "
                            "the prices come from one-class instruction streams and bound a mixed stream from above, so a model figure above 1 reads as: issue slots full"}
            if not args.no_cpu_baseline:
                ob = importlib.import_module("oracle_binding")      # the oracle: checker/baseline only
                n = gw * gh * gd
                O = ob.OracleWorld.from_chunks([world.chunk(i, copy=False) for i in range(n)], gw, gh, gd, 128)
                cores = os.cpu_count() or 1
                # the sample: the last timed launch's last frame that an off-lattice camera saw (any frame of a fixed path)
                sub = last_sub
                while P > 1 and sub > 0 and (lastf - (last_sub - sub)) % P == 0:
                    sub -= 1
                fidx = lastf - (last_sub - sub)
                cam = path[fidx % P]
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
                    "sample": f"path camera {fidx % P} (eye x = {cam.eye[0]:.3f}), {rect[2]}x{rect[3]} pixels (rect x0={rect[0]}, y0={rect[1]}), "
                              f"{O.last_rays} rays incl. shadow, {dt:.2f} s wall on {cores} threads, oracle/svo_oracle.c -O2",
                }
                # parity of the timed product output against the oracle on that sample
                got = last_out[sub].cpu().numpy().view(svo.HIT_DTYPE).reshape(ih, iw)[rect[1]:rect[1] + rect[3], rect[0]:rect[0] + rect[2]]
                same = all(np.array_equal(got[f], ref[f]) for f in ("flags", "material", "chunk", "node", "cell")) and \
                    np.array_equal(got["t"].view(np.uint32), ref["t"].view(np.uint32))
                result["cpu_baseline"]["parity_with_gpu"] = bool(same)
                result["cpu_baseline"]["parity_camera_off_lattice"] = bool(P > 1 and fidx % P != 0)
                # one thread, centred 256x256 crop of the same frame
                c1 = 256
                r1 = ((iw - c1) // 2, (ih - c1) // 2, c1, c1)
                t0 = time.perf_counter()
                O.trace_image(cam, rect=r1, params=ob.make_params(shadow=shadow), threads=1)
                dt1 = time.perf_counter() - t0
                result["cpu_baseline"]["single_thread"] = {"value": round(O.last_rays / dt1 / 1e6, 4), "unit": "Mrays/s", "cores": 1,
                                                           "sample": f"centred {c1}x{c1} crop of the same frame, {O.last_rays} rays, {dt1:.2f} s"}
        print(json.dumps(result), flush=True)

    world.destroy()
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
