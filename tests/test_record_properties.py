"""What a G-buffer record MEANS, checked against the pools it points into - without any march: a third line of evidence beside the
oracle and its Python twin (which both restate the march), and the size-independent property the full-size configs can be held to.

For every hit record (chunk, node, cell, material, t, normal, flags) of a frame or ray list:
  * the node word at tree[node] of that chunk is a LEAF (cell == 0xFF) whose offset is the material, or a TWIG whose brick holds
    the material at `cell` (src/Octree.h:8-45) - never EMPTY, never a BRANCH;
  * the voxel box - the node's box, rebuilt from the node INDEX alone through parent links (position + child-slot bits, exact
    arithmetic: no float descent), or the brick cell's sub-box - contains the hit position alpha + beta * t to within the march's
    step EPS (plus the rounding of the position itself), and so does the box of the chunk the record names;
  * the normal is one of cubeNormal's values (shaders/Chunkmarch.glsl:128-136: components of an integer vector in {-1, 0, 1}
    normalised, or NaN), or - SVO_NORMAL_FACE - a signed unit axis; the flags are consistent; a miss is an all-zero record.
CPU: the C oracle's records.  GPU (-m gpu): both kernels' records of the full C3 frame (BASELINE configs[2], 1920x1080, 16 chunks
of depth 12, primary + shadow) and of C2's, under both semantics."""
import numpy as np
import pytest

from helpers import random_rays

EPS = 1.0 / 8192.0


def parents_of(tree):
    """parent[n], slot[n], level[n] of every node reachable from the root, from the BRANCH words alone (children of node p live at
    offset(p) .. offset(p) + 7, slot = x + 2y + 4z, src/Octree.cpp:55-65)."""
    tree = np.asarray(tree, dtype=np.uint32)
    n = tree.size
    parent = np.full(n, -1, np.int64); slot = np.zeros(n, np.int8); level = np.zeros(n, np.int8)
    frontier = np.array([0], np.int64)
    lv = 0
    while frontier.size:
        words = tree[frontier]
        br = frontier[(words >> 30) == 2]
        if br.size == 0:
            break
        first = (tree[br] & 0x3FFFFFFF).astype(np.int64)
        kids = (first[:, None] + np.arange(8)[None, :]).reshape(-1)
        parent[kids] = np.repeat(br, 8); slot[kids] = np.tile(np.arange(8, dtype=np.int8), br.size); level[kids] = lv + 1
        frontier = kids
        lv += 1
    return parent, slot, level


def node_boxes(chunk, parent, slot, level, nodes):
    """(lo, size) of the given nodes: walk up the parent links, adding slot bits * size / 2^level (exact in float64)."""
    nodes = np.asarray(nodes, np.int64)
    lo = np.zeros((nodes.size, 3), np.float64)
    size = float(chunk["size"]) / (2.0 ** level[nodes].astype(np.float64))
    cur = nodes.copy()
    while True:
        live = cur > 0
        if not live.any():
            break
        s = slot[cur[live]].astype(np.int64)
        edge = float(chunk["size"]) / (2.0 ** level[cur[live]].astype(np.float64))
        lo[live, 0] += (s & 1) * edge; lo[live, 1] += ((s >> 1) & 1) * edge; lo[live, 2] += ((s >> 2) & 1) * edge
        cur[live] = parent[cur[live]]
    return lo + np.asarray(chunk["position"], np.float64)[None, :], size


def check_records(rec, origins, dirs, chunks, shadow, normal_mode=0, eps=EPS, leaf_backoff=True, what=""):
    rec = np.asarray(rec).reshape(-1)
    o = np.asarray(origins, np.float64).reshape(-1, 3); d = np.asarray(dirs, np.float64).reshape(-1, 3)
    hit = (rec["flags"] & 1) != 0
    miss = ~hit
    # a miss is an all-zero record (but for the error flag, which no test world raises)
    assert not rec["t"][miss].any() and not rec["material"][miss].any() and not rec["node"][miss].any() and not rec["flags"][miss].any(), what
    f = rec["flags"][hit]
    assert np.all(((f & 2) != 0) == bool(shadow)), f"{what}: SHADOW_TRACED on exactly the hits of a shadow-casting launch"
    assert np.all((f & 4) <= ((f & 2) << 1)), f"{what}: SHADOWED only where a shadow ray was traced"
    assert np.all(((f & 8) != 0) == (normal_mode == 1)), what
    nrm = rec["normal"][hit].astype(np.float64)
    if normal_mode == 1:
        assert np.all(np.sort(np.abs(nrm), axis=1) == np.array([0.0, 0.0, 1.0])), f"{what}: a face normal is a signed unit axis"
    else:
        ok = np.isnan(nrm).all(axis=1)
        for k, c in ((1, 1.0), (2, np.float32(1.0) / np.sqrt(np.float32(2.0))), (3, np.float32(1.0) / np.sqrt(np.float32(3.0)))):
            a = np.abs(nrm)
            ok |= ((a == 0) | (np.abs(a - float(c)) < 1e-7)).all(axis=1) & ((a != 0).sum(axis=1) == k)
        assert ok.all(), f"{what}: cubeNormal yields a normalised vector of -1 / 0 / 1 components or NaN"
    idx = np.nonzero(hit)[0]
    total = 0
    for ci in np.unique(rec["chunk"][idx]):
        c = chunks[int(ci)]
        tree = np.asarray(c["tree"], np.uint32); twig = np.asarray(c["twig"], np.uint16).reshape(-1, 64)
        sel = idx[rec["chunk"][idx] == ci]
        node = rec["node"][sel].astype(np.int64); cell = rec["cell"][sel].astype(np.int64); mat = rec["material"][sel].astype(np.int64)
        assert node.max() < tree.size
        word = tree[node]; typ = word >> 30; off = (word & 0x3FFFFFFF).astype(np.int64)
        leaf = cell == 0xFF
        assert np.all(typ[leaf] == 1) and np.all((off[leaf] & 0xFFFF) == mat[leaf]), f"{what}: a LEAF hit names a LEAF node and its material"
        assert np.all(typ[~leaf] == 3) and np.all(cell[~leaf] < 64), f"{what}: a brick hit names a TWIG node and a cell"
        assert np.all(twig[off[~leaf], cell[~leaf]] == mat[~leaf]) and np.all(mat[~leaf] != 0), f"{what}: the brick cell holds the material, and it is solid"
        parent, slot, level = parents_of(tree)
        assert np.all((parent[node] >= 0) | (node == 0)), f"{what}: the node is reachable from the root"
        lo, size = node_boxes(c, parent, slot, level, node)
        vox = size / 4.0
        cz, cy, cx = cell >> 4, (cell >> 2) & 3, cell & 3
        lo[~leaf] += np.stack([cx, cy, cz], axis=1)[~leaf] * vox[~leaf, None]
        size = np.where(leaf, size, vox)
        # the hit position: LEAF hits are reported EPS early by the CPU march (src/Traverse.cpp:93), brick hits and the shader's where they are
        t = rec["t"][sel].astype(np.float64) + (eps if leaf_backoff else 0.0) * leaf
        q = o[sel] + d[sel] * t[:, None]
        tol = 3.0 * eps * np.abs(d[sel]).max(axis=1) + 64.0 * np.spacing(np.abs(q).max(axis=1).astype(np.float32)).astype(np.float64) + 1e-6
        outside = np.maximum(np.maximum(lo - q, q - (lo + size[:, None])), 0.0).max(axis=1)
        bad = outside > tol
        assert not bad.any(), f"{what}: {bad.sum()} hit positions lie outside their voxel (worst {outside.max():.3e}, tolerance {tol[bad][:3]})"
        clo = np.asarray(c["position"], np.float64)
        assert np.all(np.maximum(np.maximum(clo - q, q - (clo + float(c["size"]))), 0.0).max(axis=1) <= tol), f"{what}: the hit lies in the chunk the record names"
        total += sel.size
    return total


def test_oracle_records_mean_what_they_say(oracle):
    ccm = (-1, 0, -1)
    O = oracle.OracleWorld.generate(2, 1, 2, 128, 7, chunkcoordmin=ccm)
    chunks = [O.chunk(i) for i in range(4)]
    o, d = random_rays(np.random.default_rng(31), 30000, (-128, 0, -128), (128, 128, 128))
    for shadow in (False, True):
        for nm in (0, 1):
            rec = O.trace_rays(o, d, params=oracle.make_params(shadow=shadow, normal_mode=nm), threads=4)
            assert check_records(rec, o, d, chunks, shadow, nm, what=f"oracle shadow={shadow} normals={nm}") > 5000
    rec = O.trace_rays(o, d, params=oracle.make_params(shadow=True, semantics=1), threads=4)
    assert check_records(rec, o, d, chunks, True, 0, eps=1.0 / 4096.0, leaf_backoff=False, what="oracle, GLSL twin") > 5000


def pixel_rays(cam):
    """Origins and directions of a camera's pixels: the product's own camera formula in float32, vectorised (include/svo.h svo_camera) - a wrong
    ray would put the hit positions outside their voxels."""
    w, h = cam.width, cam.height
    fx = (np.arange(w, dtype=np.float32) + np.float32(0.5)); fy = (np.arange(h, dtype=np.float32) + np.float32(0.5))
    u = ((fx / np.float32(w)) * np.float32(2.0) - np.float32(1.0)) * np.float32(cam.tan_half_x)
    v = (np.float32(1.0) - (fy / np.float32(h)) * np.float32(2.0)) * np.float32(cam.tan_half_y)
    fwd, right, up = (np.array(list(x), np.float32) for x in (cam.forward, cam.right, cam.up))
    dirs = (fwd[None, None, :] + right[None, None, :] * u[None, :, None]) + up[None, None, :] * v[:, None, None]
    dirs = dirs * (np.float32(1.0) / np.sqrt((dirs * dirs).sum(axis=2, dtype=np.float32)))[:, :, None]
    dirs = dirs.reshape(-1, 3).astype(np.float32)
    return np.broadcast_to(np.array(list(cam.eye), np.float32), dirs.shape), dirs


@pytest.mark.gpu
def test_full_size_frames_mean_what_they_say(svo, oracle):
    """BASELINE configs[2] at full size: every one of the ~1.1 M hit records of the 1920x1080 C3 frame (both kernels, CPU semantics and
    the GLSL twin's) against the world's pools - node words, materials, voxel boxes from node indices, chunk boxes; C2's frame; and C5's (depth 16, sparse)."""
    import bench
    for workload, (gw, gd, depth) in (("c3_1080p_depth12_4x1x4_shadow", (4, 4, 12)), ("c2_1080p_depth10_1chunk", (1, 1, 10))):
        W = svo.World.generate(gw, 1, gd, 128, depth, build_device=0)
        chunks = [W.chunk(i, copy=False) for i in range(gw * gd)]
        cam = bench.camera_path(svo, workload, gw, gd, 1920, 1080)[7]
        origins, dirs = pixel_rays(cam)
        shadow = "shadow" in workload
        for kernel in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
            for sem, eps, back in ((0, EPS, True), (1, 1.0 / 4096.0, False)):
                rec = W.draw(cam, shadow=shadow, kernel=kernel, semantics=sem)
                n = check_records(rec, origins, dirs, chunks, shadow, 0, eps=eps, leaf_backoff=back, what=f"{workload} kernel {kernel} semantics {sem}")
                assert n > 300000, (workload, n)
        W.destroy()
    # BASELINE configs[4]: one depth-16 chunk, refined to full depth only in a band (bricks under nodes of level 8 and of level 14 in one tree)
    scene = svo.c5_scene()
    W = svo.World.generate(1, 1, 1, 128, 16, **scene["generate"])
    chunks = [W.chunk(0, copy=False)]
    W.upload(0)
    cam = scene["camera"](1920, 1080)
    origins, dirs = pixel_rays(cam)
    for kernel in (svo.KERNEL_STACK, svo.KERNEL_LITERAL):
        rec = W.draw(cam, shadow=True, kernel=kernel)
        n = check_records(rec, origins, dirs, chunks, True, 0, what=f"c5 kernel {kernel}")
        assert n > 300000, n
    W.destroy()
