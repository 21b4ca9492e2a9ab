// terrain.cpp — host-side world generation for libsvo_amd (the World::init half of the boundary).
//
// What the reference does (and this file reproduces bit-for-bit on the same parameters):
//   World::g_pyramid   src/World.cpp:296-306      one Simplex height pyramid per (x,z) chunk column
//   BoundsPyramid      src/BoundsPyramid.cpp:47-174  base noise + min/max mips + bound lookup
//   grow()             src/Octree.cpp:74-176      BFS build of one chunk's SVO from height bounds
//   Ocroot::build      src/Octree.cpp:320-436     the water plane (World::g_chunk, src/World.cpp:316-320)
//
// How it is built here: the pyramid is ONE flat array per bound (level offsets (4^lv-1)/3,
// row-major per level) instead of an array of per-level allocations; the BFS is level-synchronous
// (a frontier vector per level — identical node order to the reference's FIFO queue, and the shape
// a device builder needs); chunk columns are generated in parallel on host threads.
//
// Compile with -ffp-contract=off: every float op must round separately, as in the reference.
#include "terrain.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>

namespace svo {

// ---------------------------------------------------------------------------------------------
// 2-D simplex noise (Ashima Arts / McEwan, the algorithm behind glm::simplex(vec2),
// src/BoundsPyramid.cpp:99).  Written from the published algorithm; GLM itself is not available.
namespace {
struct Noise2 {
    static float wrap289(float x) { return x - std::floor(x * (1.0f / 289.0f)) * 289.0f; }
    static float perm(float x) { return wrap289(((x * 34.0f) + 1.0f) * x); }
    static float frac(float x) { return x - std::floor(x); }
    static float clamp0(float x) { return (x < 0.0f) ? 0.0f : x; }

    static float eval(float vx, float vy)
    {
        constexpr float SKEW = 0.366025403784439f, UNSKEW = 0.211324865405187f;
        constexpr float OFF2 = -0.577350269189626f, INV41 = 0.024390243902439f;

        const float skew = vx * SKEW + vy * SKEW;
        float cx = std::floor(vx + skew), cy = std::floor(vy + skew);
        const float unskew = cx * UNSKEW + cy * UNSKEW;
        const float d0x = vx - cx + unskew, d0y = vy - cy + unskew;
        const bool lower = d0x > d0y;
        const float sx = lower ? 1.0f : 0.0f, sy = lower ? 0.0f : 1.0f;
        const float d1x = (d0x + UNSKEW) - sx, d1y = (d0y + UNSKEW) - sy;
        const float d2x = d0x + OFF2, d2y = d0y + OFF2;

        cx = cx - 289.0f * std::floor(cx / 289.0f);
        cy = cy - 289.0f * std::floor(cy / 289.0f);
        const float h[3] = {
            perm(perm(cy + 0.0f) + cx + 0.0f),
            perm(perm(cy + sy) + cx + sx),
            perm(perm(cy + 1.0f) + cx + 1.0f) };
        const float dx[3] = { d0x, d1x, d2x }, dy[3] = { d0y, d1y, d2y };

        float w[3], g[3];
        for (int k = 0; k < 3; ++k) {
            float m = clamp0(0.5f - (dx[k] * dx[k] + dy[k] * dy[k]));
            m = m * m;
            m = m * m;
            const float gx = 2.0f * frac(h[k] * INV41) - 1.0f;
            const float gh = std::fabs(gx) - 0.5f;
            const float ga = gx - std::floor(gx + 0.5f);
            m *= 1.79284291400159f - 0.85373472095314f * (ga * ga + gh * gh);
            w[k] = m;
            g[k] = ga * dx[k] + gh * dy[k];
        }
        return 130.0f * (w[0] * g[0] + w[1] * g[1] + w[2] * g[2]);
    }
};
} // namespace

float simplex2(float x, float y) { return Noise2::eval(x, y); }

// ---------------------------------------------------------------------------------------------
void HeightPyramid::build(uint32_t res, float ampl, float period, float xshift, float yshift, float zshift)
{
    size = res;
    levels = 0;
    while ((1u << levels) < res) ++levels;
    amplitude = ampl;
    shift = yshift;
    const size_t total = level_offset(levels + 1);
    lo.assign(total, 1.0f);             // reference initial values, src/BoundsPyramid.cpp:60-69
    hi.assign(total, -1.0f);

    // base level == both bounds (src/BoundsPyramid.cpp:70,92-104)
    float *blo = lo.data() + level_offset(levels), *bhi = hi.data() + level_offset(levels);
    for (size_t z = 0; z < size; ++z)
        for (size_t x = 0; x < size; ++x) {
            const float n = simplex2(((float)x + xshift) * period, ((float)z + zshift) * period);
            blo[z * size + x] = n;
            bhi[z * size + x] = n;
        }
    // mips, finest to coarsest (src/BoundsPyramid.cpp:72-77,106-135).  min/max of the 2x2 footprint
    // folded into the level's initial value (+1 / -1), which is what the reference's pairwise
    // accumulation amounts to.
    for (uint32_t lv = levels; lv > 0; --lv) {
        const size_t s = (size_t)1 << lv, up = s / 2;
        const float *flo = lo.data() + level_offset(lv), *fhi = hi.data() + level_offset(lv);
        float *clo = lo.data() + level_offset(lv - 1), *chi = hi.data() + level_offset(lv - 1);
        for (size_t z = 0; z < s; ++z)
            for (size_t x = 0; x < s; x += 2) {
                const size_t k = (z / 2) * up + x / 2;
                const float a = flo[z * s + x], b = flo[z * s + x + 1];
                const float ab = (b < a) ? b : a;
                clo[k] = (ab < clo[k]) ? ab : clo[k];
                const float c = fhi[z * s + x], d = fhi[z * s + x + 1];
                const float cd = (c < d) ? d : c;
                chi[k] = (chi[k] < cd) ? cd : chi[k];
            }
    }
}

float HeightPyramid::bound(const std::vector<float> &q, float x, float z, uint32_t lv) const
{   // src/BoundsPyramid.cpp:146-174
    const size_t a = (size_t)(x * (float)size);
    const size_t b = (size_t)(z * (float)size);
    if (lv <= levels) {
        const size_t d = (size_t)1 << (levels - lv);
        return q[level_offset(lv) + (b / d) * (size / d) + a / d] * amplitude + shift;
    }
    // finer than the base: wrapped bilinear interpolation of the base samples
    const float *base = lo.data() + level_offset(levels);
    const size_t m = size - 1;
    const size_t a1 = (a + 1) & m, b1 = (b + 1) & m;
    const float t = (float)(x * (float)size) - (float)a;
    const float s = (float)(z * (float)size) - (float)b;
    auto mix = [](float v0, float v1, float w) { return (float)((double)(v1 * w) + (1.0 - (double)w) * (double)v0); };
    const float r0 = mix(base[b * size + a], base[b * size + a1], t);
    const float r1 = mix(base[b1 * size + a], base[b1 * size + a1], t);
    return mix(r0, r1, s) * amplitude + shift;
}

// ---------------------------------------------------------------------------------------------
static uint16_t height_material(float y)
{   // src/Octree.cpp:69-72: (uint16_t) clamp(y / 0.03, 1.0, 4.0) evaluated in double
    double v = (double)y / 0.03;
    if (v < 1.0) v = 1.0;
    if (4.0 < v) v = 4.0;
    return (uint16_t)v;
}

void ChunkPools::reserve_tree(uint64_t need)
{   // capacity doubling as in src/Octree.cpp:160-161 (kept: capacity is part of Ocroot)
    while (need >= tree_capacity) tree_capacity *= 2;
    if (tree.capacity() < tree_capacity) tree.reserve(tree_capacity);
}

void grow_chunk(ChunkPools &c, const float position[3], float size, uint32_t depth, const HeightPyramid &pyr,
                const TerrainParams *sparse)
{   // src/Octree.cpp:74-176, level-synchronous
    const bool coarse = sparse && sparse->coarse_depth >= TWIG_LEVELS && sparse->coarse_depth < depth;
    c.position[0] = position[0]; c.position[1] = position[1]; c.position[2] = position[2];
    c.size = size;
    c.depth = depth;
    c.tree_capacity = 16;
    c.twig_capacity = 16;
    c.tree.assign(1, 0u);
    c.twig.clear();

    struct Cell { float x, y, z; uint32_t slot; };
    std::vector<Cell> frontier{ { position[0], position[1], position[2], 0u } }, next;
    float edge = size;
    for (uint32_t level = 0; !frontier.empty(); ++level) {
        next.clear();
        const float half = edge / 2;
        for (const Cell &e : frontier) {
            const float px = (e.x - position[0]) / size;
            const float py = (e.y - position[1]) / size;
            const float pz = (e.z - position[2]) / size;
            const float low = pyr.min(px, pz, level);
            const float high = pyr.max(px, pz, level);
            if (high < e.y) {
                c.tree[e.slot] = node_make(EMPTY, 0);
            } else if (low > e.y + edge) {
                c.tree[e.slot] = node_make(LEAF, height_material(py));
            } else if (level == depth - TWIG_LEVELS ||
                       (coarse && level == sparse->coarse_depth - TWIG_LEVELS &&
                        !(e.x + edge >= sparse->refine_min[0] && e.y + edge >= sparse->refine_min[1] && e.z + edge >= sparse->refine_min[2] &&
                          sparse->refine_max[0] >= e.x && sparse->refine_max[1] >= e.y && sparse->refine_max[2] >= e.z))) {
                // a brick: at depth-2 (the reference), or earlier outside the refine box (sparse extension)
                const float voxel = edge / (float)(1 << TWIG_LEVELS);
                const uint16_t mat = height_material(py);
                const size_t at = c.twig.size();
                while (c.twig_count() >= c.twig_capacity) c.twig_capacity *= 2;
                c.twig.resize(at + TWIG_WORDS);
                for (uint32_t z = 0; z < TWIG_SIZE; ++z)
                    for (uint32_t x = 0; x < TWIG_SIZE; ++x) {
                        // one column lookup serves all four y layers (src/Octree.cpp:131-144)
                        const float dx = ((float)x * voxel) / size;
                        const float dz = ((float)z * voxel) / size;
                        const float h = pyr.max(px + dx, pz + dz, level + TWIG_LEVELS);
                        for (uint32_t y = 0; y < TWIG_SIZE; ++y)
                            c.twig[at + z * 16 + y * 4 + x] = (h >= e.y + (float)y * voxel) ? mat : (uint16_t)0;
                    }
                c.tree[e.slot] = node_make(TWIG, (uint32_t)(at / TWIG_WORDS));
            } else {
                const uint64_t first = c.tree.size();
                c.reserve_tree(first + 8);
                c.tree.resize(first + 8, 0u);
                for (uint32_t i = 0; i < 8; ++i) {
                    const float ox = (i & 1) ? 1.0f : 0.0f, oy = (i & 2) ? 1.0f : 0.0f, oz = (i & 4) ? 1.0f : 0.0f;
                    next.push_back({ e.x + ox * half, e.y + oy * half, e.z + oz * half, (uint32_t)first + i });
                }
                c.tree[e.slot] = node_make(BRANCH, (uint32_t)first);
            }
        }
        frontier.swap(next);
        edge = half;
    }
}

// ---------------------------------------------------------------------------------------------
// Box fill (Ocroot::build, src/Octree.cpp:320-436).  Depth-first in child-slot order so that
// appended blocks/bricks land at the same indices as in the reference.
namespace {
struct Box { float lo[3], hi[3]; };
inline bool touches(const Box &a, const Box &b)
{   // cubesIntersect, src/Traverse.cpp:173-178 (closed boxes: touching counts)
    return a.hi[0] >= b.lo[0] && a.hi[1] >= b.lo[1] && a.hi[2] >= b.lo[2] &&
           b.hi[0] >= a.lo[0] && b.hi[1] >= a.lo[1] && b.hi[2] >= a.lo[2];
}
inline bool contains(const Box &outer, const Box &inner)
{   // cubeIsInside, src/Traverse.cpp:180-185
    return inner.lo[0] >= outer.lo[0] && inner.lo[1] >= outer.lo[1] && inner.lo[2] >= outer.lo[2] &&
           outer.hi[0] >= inner.hi[0] && outer.hi[1] >= inner.hi[1] && outer.hi[2] >= inner.hi[2];
}
inline Box cube(float x, float y, float z, float edge) { return { { x, y, z }, { x + edge, y + edge, z + edge } }; }

struct Filler {
    ChunkPools &c;
    Box region;
    uint16_t material;
    DirtyRange &dt, &dw;

    void touch_node(uint64_t i) { dt.left = std::min(dt.left, i); dt.right = std::max(dt.right, i + 1); }
    void touch_brick(uint64_t i) { dw.left = std::min(dw.left, i); dw.right = std::max(dw.right, i + 1); }

    void visit(uint64_t slot, float x, float y, float z, float edge, uint32_t level)
    {
        const Box box = cube(x, y, z, edge);
        if (!touches(box, region)) return;
        const uint32_t word = c.tree[slot];
        switch (node_type(word)) {
        case LEAF:
            return;
        case EMPTY:
            if (contains(region, box)) {
                touch_node(slot);
                c.tree[slot] = node_make(LEAF, material);
            } else if (level == c.depth - TWIG_LEVELS) {
                if (c.twig_count() >= c.twig_capacity) { c.twig_capacity *= 2; dw.realloc = true; }
                const uint64_t brick = c.twig_count();
                c.twig.resize(c.twig.size() + TWIG_WORDS, 0);
                touch_brick(brick);
                touch_node(slot);
                c.tree[slot] = node_make(TWIG, (uint32_t)brick);
                visit(slot, x, y, z, edge, level);
            } else {
                const uint64_t first = c.tree.size();
                if (first + 8 >= c.tree_capacity) { c.tree_capacity *= 2; dt.realloc = true; }
                dt.left = std::min(dt.left, slot);
                c.tree[slot] = node_make(BRANCH, (uint32_t)first);
                dt.right = std::max(dt.right, first + 8);
                c.tree.resize(first + 8, node_make(EMPTY, 0));
                visit(slot, x, y, z, edge, level);
            }
            return;
        case TWIG: {
            const float voxel = edge / (float)(1 << TWIG_LEVELS);
            const uint64_t brick = node_offset(word);
            touch_brick(brick);
            uint16_t *cells = c.twig.data() + brick * TWIG_WORDS;
            for (uint32_t cz = 0; cz < TWIG_SIZE; ++cz)
                for (uint32_t cy = 0; cy < TWIG_SIZE; ++cy)
                    for (uint32_t cx = 0; cx < TWIG_SIZE; ++cx) {
                        uint16_t &cell = cells[cz * 16 + cy * 4 + cx];
                        if (cell != 0) continue;
                        const Box vb = cube(x + (float)cx * voxel, y + (float)cy * voxel, z + (float)cz * voxel, voxel);
                        if (touches(vb, region)) cell = material;
                    }
            return;
        }
        default: {
            const float half = edge * 0.5f;
            const uint64_t first = node_offset(word);
            for (uint32_t i = 0; i < 8; ++i) {
                const float ox = (i & 1) ? 1.0f : 0.0f, oy = (i & 2) ? 1.0f : 0.0f, oz = (i & 4) ? 1.0f : 0.0f;
                visit(first + i, x + ox * half, y + oy * half, z + oz * half, half, level + 1);
            }
        }
        }
    }
};
} // namespace

void fill_box(ChunkPools &c, const float lo[3], const float hi[3], uint16_t material, DirtyRange &dtree, DirtyRange &dtwig)
{
    dtree = DirtyRange(); dtwig = DirtyRange();
    Filler f{ c, { { lo[0], lo[1], lo[2] }, { hi[0], hi[1], hi[2] } }, material, dtree, dtwig };
    f.visit(0, c.position[0], c.position[1], c.position[2], c.size, 0);
}

// ---------------------------------------------------------------------------------------------
static int positive_mod(int n, int m) { return (m + (n % m)) % m; }   // src/World.cpp:276-279

int generate_world(int w, int h, int d, int chunksize, const int ccm[3], const TerrainParams &tp,
                   std::vector<ChunkPools> &chunks)
{   // World::init, src/World.cpp:19-43; g_pyramid :296-306; g_chunk :308-321
    chunks.assign((size_t)w * h * d, ChunkPools());
    const uint32_t res = tp.pyramid_resolution ? tp.pyramid_resolution : (1u << tp.depth);
    const int columns = w * d;
    int nthreads = tp.threads > 0 ? tp.threads : (int)std::thread::hardware_concurrency();
    nthreads = std::max(1, std::min(nthreads, columns));

    std::atomic<int> cursor{ 0 };
    std::atomic<int> failed{ 0 };               // a worker ran out of memory: nothing may escape a std::thread (std::terminate)
    auto worker = [&]() {
        try {
        HeightPyramid pyr;
        for (;;) {
            const int col = cursor.fetch_add(1);
            if (col >= columns) break;
            const int xi = col % w, zi = col / w;
            const int cx = ccm[0] + xi, cz = ccm[2] + zi;
            const float period = 1.0f / (float)res;
            pyr.build(res, tp.amplitude, period,
                      (float)cx * (float)res + (float)tp.seed, tp.yshift,
                      (float)cz * (float)res + (float)tp.seed);
            for (int yi = 0; yi < h; ++yi) {
                const int cy = ccm[1] + yi;
                const int idx = positive_mod(cy, h) * w * d + positive_mod(cz, d) * w + positive_mod(cx, w);
                ChunkPools &c = chunks[(size_t)idx];
                const float pos[3] = { (float)cx * (float)chunksize, (float)cy * (float)chunksize, (float)cz * (float)chunksize };
                grow_chunk(c, pos, (float)chunksize, tp.depth, pyr, &tp);
                if (tp.water) {
                    const float hi[3] = { c.position[0] + c.size, tp.water_level, c.position[2] + c.size };
                    DirtyRange a, b;
                    fill_box(c, c.position, hi, (uint16_t)tp.water_material, a, b);
                }
            }
        }
        } catch (...) {
            failed.store(1);
            cursor.store(columns);              // the other workers stop at their next column
        }
    };
    std::vector<std::thread> pool;
    for (int i = 1; i < nthreads; ++i) pool.emplace_back(worker);
    worker();
    for (auto &t : pool) t.join();
    return failed.load() ? -1 : 0;
}

} // namespace svo
