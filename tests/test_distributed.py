"""The N>1 path on CPU: world_size-2 gloo processes partition a frame into round-robin 8-row bands
(octree-raymarcher_amd/partition.py, the same helpers bench.py uses with RCCL), trace their bands, gather to
rank 0, de-interleave, and the result must equal the single-process frame.  The per-band tracing here is done
by the CPU oracle (this is a test of the partition + exchange logic; GPU tracing of bands is covered by
test_gpu_parity.py::test_camera_rect_and_bands)."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world_size, port, width, height, out_path):
    for p in (HERE, ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    svo = importlib.import_module("octree-raymarcher_amd")
    ob = importlib.import_module("oracle_binding")
    part = svo.partition
    O = ob.OracleWorld.generate(2, 1, 2, 128, 5)
    cam = svo.default_camera(2, 2, 128, width, height)
    prm = ob.make_params(shadow=True)
    nb = part.bands_per_rank(height, world_size)
    mine = np.zeros((nb, part.BAND, width), dtype=ob.HIT_DTYPE)
    for k in range(nb):
        rows = [y for y in part.band_rows(rank, world_size, k) if y < height]
        if rows:
            mine[k, :len(rows)] = O.trace_image(cam, rect=(0, rows[0], width, len(rows)), params=prm)
    buf = torch.from_numpy(mine.view(np.uint8).reshape(nb, part.BAND, width, 32))
    gathered = [torch.empty_like(buf) for _ in range(world_size)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0)
    if rank == 0:
        frame = part.deinterleave(gathered, height).numpy().view(ob.HIT_DTYPE).reshape(height, width)
        full = O.trace_image(cam, params=prm)
        ok = all(np.array_equal(frame[f], full[f]) for f in ("flags", "material", "chunk", "node", "cell")) and \
            np.array_equal(frame["t"].view(np.uint32), full["t"].view(np.uint32))
        np.save(out_path, np.array([int(ok), int((full["flags"] & 1).sum())]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("height", [64, 52])        # 52: not a multiple of the band height / of 2 bands
def test_band_partition_gather_world_size_2(tmp_path, height):
    out = str(tmp_path / "result.npy")
    port = 29500 + (os.getpid() % 1000) + height
    mp.spawn(_worker, args=(2, port, 96, height, out), nprocs=2, join=True)
    ok, hits = np.load(out)
    assert ok == 1 and hits > 200


def test_partition_helpers_cover_every_row_once():
    part = importlib.import_module("octree-raymarcher_amd").partition
    for height in (1080, 2160, 52, 8, 7):
        for n in (1, 2, 4, 8):
            nb = part.bands_per_rank(height, n)
            seen = np.zeros(nb * n * part.BAND, int)
            for r in range(n):
                for k in range(nb):
                    for y in part.band_rows(r, n, k):
                        seen[y] += 1
            assert np.all(seen[:height] == 1) and seen.size >= height
