"""Timing probe (GPU box): the host path of a world - svo_world_generate on host threads, svo_world_upload of its pools (H2D, masks, wide
trees), svo_world_update of one chunk - at C3 size (4x1x4 chunks of depth 12)."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
svo = importlib.import_module("octree-raymarcher_amd")
t = time.time(); W = svo.World.generate(4, 1, 4, 128, 12); print(f"host generate: {time.time()-t:.2f} s", flush=True)
i = W.info; nbytes = i.total_trees * 4 + i.total_twigs * 128
for rep in range(2):
    t = time.time(); W.upload(0); dt = time.time() - t
    print(f"upload: {dt*1e3:.0f} ms for {nbytes/2**30:.2f} GiB of pools ({nbytes/dt/1e9:.1f} GB/s incl. masks and wide trees)", flush=True)
    W2 = svo.World.create([W.chunk(k, copy=False) for k in range(16)], 4, 1, 4, 128)
    t = time.time(); W2.upload(0); dt = time.time() - t
    print(f"create + upload of the same pools: {dt*1e3:.0f} ms", flush=True)
    W2.destroy()
c = W.chunk(5)
nb = c["twig"].size // 64
for what, tr, br, realloc in (("a small dirty range (64 nodes, 2 bricks)", (1000, 1064), (10, 12), False),
                              ("the whole chunk as dirty range", (0, c["tree"].size), (0, nb), False),
                              ("the whole chunk, realloc (the pools were reallocated on the caller's side)", (0, c["tree"].size), (0, nb), True)):
    for rep in range(2):
        t = time.time(); W.update(5, c, tree_range=tr, twig_range=br, realloc=realloc); dt = time.time() - t
    print(f"svo_world_update of a depth-12 chunk ({(c['tree'].nbytes + c['twig'].nbytes)/2**20:.0f} MiB), {what}: {dt*1e3:.1f} ms", flush=True)
