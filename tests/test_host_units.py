"""Host-only checks of product arithmetic that has no GPU test of its own at scale:

* the multi-GPU exchange's band / pitch / offset arithmetic (host/band_layout.hpp, used by host/multi_gpu.hpp) for N = 2, 4, 8 -
  the N > 1 exchange has never run on hardware (one-GPU boxes), so the copies MultiGpuWorld::draw_frames would issue are applied
  with memcpy on tagged buffers, and the result is held against partition.deinterleave (bench.py's Python twin);
* the device fill's per-node word (csrc/svo_format.h fill_pack) at the frontier sizes the builder admits.
"""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "octree-raymarcher_amd", "host")
BUILD = os.path.join(ROOT, "octree-raymarcher_amd", "build")


def compile_host(src, extra=()):
    os.makedirs(BUILD, exist_ok=True)
    exe = os.path.join(BUILD, os.path.splitext(src)[0])
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I.", *extra, src, "-o", exe], cwd=HOST, check=True)
    return exe


def tags(r, f, k, j, x):
    return (np.uint64(r) << np.uint64(56)) | (np.uint64(f) << np.uint64(48)) | (k.astype(np.uint64) << np.uint64(32)) | \
           (j.astype(np.uint64) << np.uint64(24)) | x.astype(np.uint64)


@pytest.mark.parametrize("ranks", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("width,height", [(40, 1080), (24, 2160), (16, 52), (8, 8), (8, 1), (8, 129)])
def test_exchange_copies_put_every_row_of_every_frame_in_place_once(ranks, width, height):
    partition = __import__("octree-raymarcher_amd.partition", fromlist=["partition"])
    exe = compile_host("band_layout_check.cpp")
    frames = 3
    r = subprocess.run([exe, str(ranks), str(width), str(height), str(frames)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()            # written once, and every row where the layout says (checked in C++)
    nb = partition.bands_per_rank(height, ranks)
    stride = nb * ranks * partition.BAND * width
    got = np.frombuffer(r.stdout, dtype=np.uint64).reshape(frames, stride)
    k, j, x = np.meshgrid(np.arange(nb), np.arange(partition.BAND), np.arange(width), indexing="ij")
    for f in range(frames):
        gathered = [tags(rk, f, k, j, x) for rk in range(ranks)]            # [nb, band, width] per rank, as svo_trace_rows stacks them
        want = partition.deinterleave(gathered, height)                     # the Python twin bench.py uses behind the RCCL gather
        assert np.array_equal(got[f, :height * width].reshape(height, width), want)
        for rk in range(ranks):
            for kk in range(nb):
                assert list(partition.band_rows(rk, ranks, kk)) == list(range((kk * ranks + rk) * 8, (kk * ranks + rk) * 8 + 8))


def test_fill_word_keeps_child_block_indices_up_to_the_frontier_limit():
    exe = compile_host("format_check.cpp", extra=["-I../csrc"])
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0 and "0 failures" in r.stdout, r.stdout
