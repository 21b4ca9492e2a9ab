"""svo_oracle_py.py — SECOND, independent restatement of the reference's CPU march, in pure Python.

TEST INFRASTRUCTURE ONLY (like svo_oracle.c; only tests/ may import it).  PARITY UNPINNED against the
reference itself (no reference tests/fixtures exist, GLM is missing so the reference cannot be built here);
its purpose is to catch transcription slips in oracle/svo_oracle.c: the two were written separately, in
different languages, from the same source lines, and tests/test_oracle_cross_check.py requires them to agree
bit for bit on small worlds.

Every function follows src/Traverse.cpp (and src/World.cpp:276-293,323-332) line by line, with numpy float32
scalars so that each operation rounds to single precision exactly where the C++ does.  Pure-Python loops:
small cases only.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32
EPS = f32(1.0) / f32(8192.0)                      # src/Traverse.cpp:8
EMPTY, LEAF, BRANCH, TWIG = 0, 1, 2, 3            # src/Octree.h:8-14
TWIG_LEVELS, TWIG_SIZE = 2, 4

np.seterr(all="ignore")                           # inf / NaN arithmetic is part of the semantics (axis-parallel rays)


def vec3(x, y, z):
    return np.array([x, y, z], dtype=f32)


def glm_min(x, y):                                # glm::min(x, y) = (y < x) ? y : x
    return y if y < x else x


def glm_max(x, y):                                # glm::max(x, y) = (x < y) ? y : x
    return y if x < y else x


def is_inside_cube(p, cmin, cmax):                # src/Traverse.cpp:18-23
    geq = bool(p[0] >= cmin[0]) and bool(p[1] >= cmin[1]) and bool(p[2] >= cmin[2])
    leq = bool(cmax[0] >= p[0]) and bool(cmax[1] >= p[1]) and bool(cmax[2] >= p[2])
    return geq and leq


def cube_escape_distance(a, b, cmin, cmax):       # src/Traverse.cpp:25-32
    gamma = np.array([f32(np.float64(1.0) / np.float64(b[k])) for k in range(3)], dtype=f32)   # 1.0 / b.x in double
    tmin = ((cmin - a).astype(f32) * gamma).astype(f32)
    tmax = ((cmax - a).astype(f32) * gamma).astype(f32)
    t = [glm_max(tmin[k], tmax[k]) for k in range(3)]
    return glm_min(t[0], glm_min(t[1], t[2]))


class Chunk:                                      # Ocroot, src/Octree.h:56-76
    def __init__(self, position, size, depth, tree, twig):
        self.position = np.asarray(position, dtype=f32)
        self.size = f32(size)
        self.depth = int(depth)
        self.tree = np.asarray(tree, dtype=np.uint32)
        self.twig = np.asarray(twig, dtype=np.uint16).reshape(-1, 64)


def node_type(v):
    return int(v) >> 30                           # src/Octree.cpp:50-53


def node_offset(v):
    return int(v) & 0x3FFFFFFF                    # src/Octree.cpp:45-48


def traverse(p, root: Chunk):                     # src/Traverse.cpp:34-48
    bmin, size, offset = root.position.copy(), root.size, 0
    while True:
        if node_type(root.tree[offset]) != BRANCH:
            return bmin, size, offset
        halfsize = f32(size * f32(0.5))
        mid = (bmin + halfsize).astype(f32)
        ge = [bool(p[k] >= mid[k]) for k in range(3)]
        bmin = (bmin + (np.array(ge, dtype=f32) * halfsize).astype(f32)).astype(f32)
        i = int(ge[0]) + int(ge[1]) * 2 + int(ge[2]) * 4          # Octree::branch, src/Octree.cpp:55-58
        offset = node_offset(root.tree[offset]) + i
        size = halfsize


def twigmarch(a, b, bmin, size, leafsize, cells):  # src/Traverse.cpp:50-72
    bmax = (bmin + size).astype(f32)
    t = f32(0.0)
    for _ in range(1000):
        p = (a + (b * t).astype(f32)).astype(f32)
        if not is_inside_cube(p, bmin, bmax):
            return False, None, None
        q = ((p - bmin).astype(f32) / leafsize).astype(f32)
        off = [int(q[k]) for k in range(3)]                       # ivec3(): truncation toward zero
        if not is_inside_cube(np.array(off, dtype=f32), vec3(0, 0, 0), vec3(TWIG_SIZE - 1, TWIG_SIZE - 1, TWIG_SIZE - 1)):
            return False, None, None
        word = off[2] * TWIG_SIZE * TWIG_SIZE + off[1] * TWIG_SIZE + off[0]      # Octwig::word, src/Octree.cpp:22-30
        if cells[word] != 0:
            return True, t, word
        leafmin = (bmin + (np.array(off, dtype=f32) * leafsize).astype(f32)).astype(f32)
        leafmax = (leafmin + leafsize).astype(f32)
        escape = cube_escape_distance(p, b, leafmin, leafmax)
        t = f32(t + f32(escape + EPS))
    return False, None, None


def treemarch(a, b, root: Chunk):                 # src/Traverse.cpp:74-113 -> (hit, s, node, cell)
    rmin = root.position
    rmax = (root.position + root.size).astype(f32)
    t = f32(0.0)
    for _ in range(1000):
        p = (a + (b * t).astype(f32)).astype(f32)
        if not is_inside_cube(p, rmin, rmax):
            return False, None, None, None
        bmin, size, offset = traverse(p, root)
        kind = node_type(root.tree[offset])
        if kind == EMPTY:
            escape = cube_escape_distance(p, b, bmin, (bmin + size).astype(f32))
            t = f32(t + f32(escape + EPS))
        elif kind == LEAF:
            return True, f32(t - EPS), offset, 0xFF
        elif kind == TWIG:
            leafsize = f32(size / f32(1 << TWIG_LEVELS))
            hit, s, word = twigmarch(p, b, bmin, size, leafsize, root.twig[node_offset(root.tree[offset])])
            if hit:
                return True, f32(s + t), offset, word
            escape = cube_escape_distance(p, b, bmin, (bmin + size).astype(f32))
            t = f32(t + f32(escape + EPS))
        else:
            raise AssertionError("BRANCH returned by traverse")
    return False, None, None, None


def intersect_cube(a, b, cmin, cmax):             # src/Traverse.cpp:115-125
    tmin = ((cmin - a).astype(f32) / b).astype(f32)
    tmax = ((cmax - a).astype(f32) / b).astype(f32)
    t1 = [glm_min(tmin[k], tmax[k]) for k in range(3)]
    t2 = [glm_max(tmin[k], tmax[k]) for k in range(3)]
    tnear = glm_max(glm_max(t1[0], t1[1]), t1[2])
    tfar = glm_min(glm_min(t2[0], t2[1]), t2[2])
    return tnear, bool(tfar > tnear)


class World:                                      # the fields chunkmarch reads, src/World.h:44-57
    def __init__(self, chunks, width, height, depth, chunksize, chunkcoordmin=(0, 0, 0)):
        self.chunk = chunks
        self.width, self.height, self.depth = width, height, depth
        self.chunksize = int(chunksize)
        self.chunkcoordmin = tuple(int(v) for v in chunkcoordmin)

    @staticmethod
    def _modulo(n, m):                            # src/World.cpp:276-279, C++ % truncates toward zero
        r = int(np.fmod(n, m))
        return (m + r) % m

    def index(self, x, y, z):                     # src/World.cpp:288-293
        return self._modulo(y, self.height) * self.width * self.depth + self._modulo(z, self.depth) * self.width + self._modulo(x, self.width)

    def index_float(self, p):                     # src/World.cpp:323-332
        q = (p / f32(self.chunksize)).astype(f32)
        out = []
        for k in range(3):
            v = q[k]
            if v < 0.0:
                v = f32(np.float64(v) - 1.0)
            out.append(int(v))
        return out


def chunkmarch(alpha, beta, world: World):        # src/Traverse.cpp:127-171 -> (hit, t, chunk, node, cell)
    alpha = np.asarray(alpha, dtype=f32)
    beta = np.asarray(beta, dtype=f32)
    chunksize = f32(world.chunksize)
    ccm = world.chunkcoordmin
    ccmax = np.array([ccm[0] + world.width, ccm[1] + world.height, ccm[2] + world.depth], dtype=f32)
    chunkmin = np.array([ccm[k] * int(chunksize) for k in range(3)], dtype=f32)
    chunkmax = (ccmax * chunksize).astype(f32)

    t = f32(0.0)
    intersect = True
    if not is_inside_cube(alpha, chunkmin, chunkmax):
        tnear, intersect = intersect_cube(alpha, beta, chunkmin, chunkmax)
        t = f32(tnear + EPS)
    if not intersect:
        return False, None, None, None, None

    for _ in range(1000):
        p = (alpha + (beta * t).astype(f32)).astype(f32)
        if not is_inside_cube(p, chunkmin, chunkmax):
            return False, None, None, None, None
        q = world.index_float(p)
        i = world.index(q[0], q[1], q[2])
        cmin = world.chunk[i].position
        cmax = (cmin + chunksize).astype(f32)
        if not is_inside_cube(p, cmin, cmax):
            return False, None, None, None, None
        hit, s, node, cell = treemarch(p, beta, world.chunk[i])
        if hit:
            t = f32(t + s)
            return True, t, i, node, cell
        escape = cube_escape_distance(p, beta, cmin, cmax)
        t = f32(t + f32(escape + EPS))
    return False, None, None, None, None
