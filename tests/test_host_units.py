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


def test_world_update_host_copy_patch_and_rollback(svo, oracle):
    """svo_world_update's host half, no device needed (a world that was never uploaded takes updates too): without `realloc` the
    library's copy of the chunk is patched over the dirty ranges + the appended tail, and equals the caller's pools after every one
    of a sequence of oracle edits (Ocdelta ranges, src/Octree.h:47-54); ranges that lie (or a malformed node inside them) are refused
    by the whole-chunk validation and the copy is restored word for word; a shrunken pool or `realloc` takes the copying path."""
    import ctypes as C
    O = oracle.OracleWorld.generate(1, 1, 1, 128, 6)
    W = svo.World.create([O.chunk(0)], 1, 1, 1, 128)
    rng = np.random.default_rng(11)
    kinds = 0
    for k in range(30):
        lo = rng.uniform(2, 110, 3); hi = lo + rng.uniform(0.3, 30, 3)
        dt, dw = oracle.Delta(), oracle.Delta()
        root = C.byref(O.w.chunk[0])
        if k % 3 == 1:
            oracle.lib.orc_destroy(root, oracle.vec3(lo), oracle.vec3(hi), C.byref(dt), C.byref(dw))
        else:
            oracle.lib.orc_build(root, oracle.vec3(lo), oracle.vec3(hi), 2 + k % 5, C.byref(dt), C.byref(dw))
        c = O.chunk(0)
        realloc = bool(dt.realloc_ or dw.realloc_)
        kinds |= 2 if realloc else 1
        assert W.update(0, c, tree_range=(min(dt.left, c["tree"].size), dt.right), twig_range=(min(dw.left, c["twig"].size // 64), dw.right), realloc=realloc) == 0
        mine = W.chunk(0)
        assert np.array_equal(mine["tree"], c["tree"]) and np.array_equal(mine["twig"], c["twig"]), f"host copy after update {k} (realloc {realloc})"
    assert kinds == 3, "the sequence exercised both the patching and the copying path"
    good = O.chunk(0)
    # a BRANCH sent past the pool inside the dirty range
    bad = dict(good); bad["tree"] = good["tree"].copy()
    victim = int(np.nonzero((bad["tree"] >> 30) == 2)[0][-1])
    bad["tree"][victim] = (2 << 30) | (bad["tree"].size + 8)
    with pytest.raises(svo.SvoError) as e:
        W.update(0, bad, tree_range=(victim, victim + 1), twig_range=(0, 0))
    assert e.value.code == -4
    # the root turned into a TWIG that points past the brick pool, in an update that also appends a block (the patch grows the copy first)
    bad2 = dict(good); bad2["tree"] = np.concatenate([good["tree"], np.zeros(8, np.uint32)])
    bad2["tree"][0] = (3 << 30) | 0x3FFFFFF
    with pytest.raises(svo.SvoError):
        W.update(0, bad2, tree_range=(0, 1), twig_range=(0, 0))
    mine = W.chunk(0)
    assert np.array_equal(mine["tree"], good["tree"]) and np.array_equal(mine["twig"], good["twig"]), "refused patches are taken back"
    # empty and inverted ranges change nothing; ranges past the end are clamped
    assert W.update(0, good, tree_range=(5, 5), twig_range=(7, 3)) == 0
    assert W.update(0, good, tree_range=(good["tree"].size + 100, good["tree"].size + 200), twig_range=(10 ** 9, 10 ** 9 + 5)) == 0
    mine = W.chunk(0)
    assert np.array_equal(mine["tree"], good["tree"]) and np.array_equal(mine["twig"], good["twig"])
    # a pool that has shrunk (a re-grown chunk) cannot be patched: the copying path takes it
    small = oracle.OracleWorld.generate(1, 1, 1, 128, 4).chunk(0)
    assert small["tree"].size < good["tree"].size
    assert W.update(0, small, tree_range=(0, 1), twig_range=(0, 0)) == 0
    mine = W.chunk(0)
    assert np.array_equal(mine["tree"], small["tree"]) and np.array_equal(mine["twig"], small["twig"]) and mine["depth"] == 4
    W.destroy()
