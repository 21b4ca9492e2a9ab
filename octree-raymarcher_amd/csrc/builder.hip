// builder.hip — device-side world generation: the World::init half of the boundary on the GPU.
//
// Produces pools bit-identical to terrain.cpp (and therefore to the reference's algorithm) for the same
// parameters:
//   BoundsPyramid::computeBase        src/BoundsPyramid.cpp:92-104   -> k_noise_base   (one thread per texel)
//   BoundsPyramid::computeBoundsAbove src/BoundsPyramid.cpp:106-135  -> k_mip_level    (one thread per coarse texel)
//   grow()                            src/Octree.cpp:74-176          -> level-synchronous BFS:
//        k_classify (EMPTY / LEAF / TWIG / BRANCH per frontier node, src/Octree.cpp:105-121)
//        exclusive scans of the BRANCH and TWIG flags (rocPRIM) == the reference queue's append order
//        k_emit     (node words, the 8 children of every BRANCH into the next frontier, :155-174)
//        k_bricks_rows (one thread per brick z-row: 4 column lookups, 16 cells, one 32-byte store, :122-154)
// The water fill (Ocroot::build) appends blocks in depth-first order and stays on the host (74 ms at depth 12).
//
// Layout on the device: the pyramid is the same flat array per bound as on the host (level lv at (4^lv-1)/3,
// row-major): mip kernels read/write whole rows coalesced; frontier entries are 16 B {x, y, z, slot}.
// Compile with -ffp-contract=off: every float op must round separately, exactly like terrain.cpp.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "terrain.h"
#include "world.h"

namespace svo {

namespace {

#define BUILD_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return e_ == hipErrorOutOfMemory ? SVO_ERR_OUT_OF_MEMORY : SVO_ERR_HIP; } \
    } while (0)

// ---- 2-D simplex noise, the same sequence of float operations as terrain.cpp's Noise2::eval ------------
__device__ __forceinline__ float d_wrap289(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
__device__ __forceinline__ float d_perm(float x) { return d_wrap289(((x * 34.0f) + 1.0f) * x); }
__device__ __forceinline__ float d_frac(float x) { return x - floorf(x); }

__device__ float d_simplex2(float vx, float vy)
{
    const float SKEW = 0.366025403784439f, UNSKEW = 0.211324865405187f;
    const float OFF2 = -0.577350269189626f, INV41 = 0.024390243902439f;
    const float skew = vx * SKEW + vy * SKEW;
    float cx = floorf(vx + skew), cy = floorf(vy + skew);
    const float unskew = cx * UNSKEW + cy * UNSKEW;
    const float d0x = vx - cx + unskew, d0y = vy - cy + unskew;
    const bool lower = d0x > d0y;
    const float sx = lower ? 1.0f : 0.0f, sy = lower ? 0.0f : 1.0f;
    const float d1x = (d0x + UNSKEW) - sx, d1y = (d0y + UNSKEW) - sy;
    const float d2x = d0x + OFF2, d2y = d0y + OFF2;
    cx = cx - 289.0f * floorf(cx / 289.0f);
    cy = cy - 289.0f * floorf(cy / 289.0f);
    const float h[3] = { d_perm(d_perm(cy + 0.0f) + cx + 0.0f), d_perm(d_perm(cy + sy) + cx + sx), d_perm(d_perm(cy + 1.0f) + cx + 1.0f) };
    const float dx[3] = { d0x, d1x, d2x }, dy[3] = { d0y, d1y, d2y };
    float w[3], g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float m = 0.5f - (dx[k] * dx[k] + dy[k] * dy[k]);
        m = (m < 0.0f) ? 0.0f : m;
        m = m * m;
        m = m * m;
        const float gx = 2.0f * d_frac(h[k] * INV41) - 1.0f;
        const float gh = fabsf(gx) - 0.5f;
        const float ga = gx - floorf(gx + 0.5f);
        m *= 1.79284291400159f - 0.85373472095314f * (ga * ga + gh * gh);
        w[k] = m;
        g[k] = ga * dx[k] + gh * dy[k];
    }
    return 130.0f * (w[0] * g[0] + w[1] * g[1] + w[2] * g[2]);
}

__global__ __launch_bounds__(256) void k_noise_base(float *lo, float *hi, uint32_t size, float period, float xshift, float zshift)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (uint64_t)size * size) return;
    const uint32_t x = (uint32_t)(i % size), z = (uint32_t)(i / size);
    const float n = d_simplex2(((float)x + xshift) * period, ((float)z + zshift) * period);
    lo[i] = n;
    hi[i] = n;
}

// coarse texel k = min / max over its 2x2 footprint, folded into the initial +1 / -1 in the reference's order
// (row z even: pair (x, x+1), then row z odd) — src/BoundsPyramid.cpp:115-134
__global__ __launch_bounds__(256) void k_mip_level(const float *flo, const float *fhi, float *clo, float *chi, uint32_t s)
{
    const uint32_t up = s / 2;
    const uint64_t k = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= (uint64_t)up * up) return;
    const uint32_t cx = (uint32_t)(k % up), cz = (uint32_t)(k / up);
    float mn = 1.0f, mx = -1.0f;
#pragma unroll
    for (uint32_t r = 0; r < 2; ++r) {
        const uint64_t at = (uint64_t)(2 * cz + r) * s + 2 * cx;
        const float a = flo[at], b = flo[at + 1];
        const float ab = (b < a) ? b : a;
        mn = (ab < mn) ? ab : mn;
        const float c = fhi[at], d = fhi[at + 1];
        const float cd = (c < d) ? d : c;
        mx = (mx < cd) ? cd : mx;
    }
    clo[k] = mn;
    chi[k] = mx;
}

struct DevPyramid {
    const float *lo, *hi;       // flat, level lv at (4^lv - 1) / 3
    uint32_t size, levels;
    float amplitude, shift;
};

__device__ __forceinline__ uint64_t d_level_offset(uint32_t lv) { return ((1ull << (2 * lv)) - 1) / 3; }

// BoundsPyramid::bound, src/BoundsPyramid.cpp:146-174 (== HeightPyramid::bound in terrain.cpp)
__device__ float d_bound(const DevPyramid &P, const float *q, float x, float z, uint32_t lv)
{
    const uint64_t a = (uint64_t)(x * (float)P.size);
    const uint64_t b = (uint64_t)(z * (float)P.size);
    if (lv <= P.levels) {
        const uint32_t sh = P.levels - lv;              // d = 2^sh: the reference's divisions by d are shifts
        return q[d_level_offset(lv) + ((b >> sh) << lv) + (a >> sh)] * P.amplitude + P.shift;
    }
    const float *base = P.lo + d_level_offset(P.levels);
    const uint64_t m = P.size - 1;
    const uint64_t a1 = (a + 1) & m, b1 = (b + 1) & m;
    const float t = (float)(x * (float)P.size) - (float)a;
    const float s = (float)(z * (float)P.size) - (float)b;
    auto mix = [](float v0, float v1, float w) { return (float)((double)(v1 * w) + (1.0 - (double)w) * (double)v0); };
    const float r0 = mix(base[b * P.size + a], base[b * P.size + a1], t);
    const float r1 = mix(base[b1 * P.size + a], base[b1 * P.size + a1], t);
    return mix(r0, r1, s) * P.amplitude + P.shift;
}

__device__ __forceinline__ uint32_t d_height_material(float y)
{   // src/Octree.cpp:69-72, in double
    double v = (double)y / 0.03;
    if (v < 1.0) v = 1.0;
    if (4.0 < v) v = 4.0;
    return (uint32_t)(uint16_t)v;
}

struct Cell { float x, y, z; uint32_t slot; };

struct GrowArgs {
    float px, py, pz, size;     // chunk position / edge
    float edge;                 // node edge at this level
    uint32_t level, depth;
    uint32_t coarse_depth;      // 0 = off
    float rmin[3], rmax[3];     // refine box
};

// type of every frontier node (src/Octree.cpp:105-121) + flags for the scans
__global__ __launch_bounds__(256) void k_classify(const Cell *frontier, uint32_t n, GrowArgs G, DevPyramid P,
                                                  uint32_t *word, uint32_t *is_branch, uint32_t *is_twig)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Cell e = frontier[i];
    const float px = (e.x - G.px) / G.size, py = (e.y - G.py) / G.size, pz = (e.z - G.pz) / G.size;
    const float low = d_bound(P, P.lo, px, pz, G.level);
    const float high = d_bound(P, P.hi, px, pz, G.level);
    uint32_t w, br = 0, tw = 0;
    if (high < e.y) {
        w = node_make(EMPTY, 0);
    } else if (low > e.y + G.edge) {
        w = node_make(LEAF, d_height_material(py));
    } else {
        const bool coarse_brick = G.coarse_depth != 0 && G.level == G.coarse_depth - TWIG_LEVELS &&
            !(e.x + G.edge >= G.rmin[0] && e.y + G.edge >= G.rmin[1] && e.z + G.edge >= G.rmin[2] &&
              G.rmax[0] >= e.x && G.rmax[1] >= e.y && G.rmax[2] >= e.z);
        if (G.level == G.depth - TWIG_LEVELS || coarse_brick) { w = node_make(TWIG, 0); tw = 1; }
        else { w = node_make(BRANCH, 0); br = 1; }
    }
    word[i] = w;
    is_branch[i] = br;
    is_twig[i] = tw;
}

// node words; children of every BRANCH appended to the next frontier in parent order (== FIFO queue order);
// brick jobs listed in TWIG order
__global__ __launch_bounds__(256) void k_emit(const Cell *frontier, uint32_t n, float half, const uint32_t *word,
                                              const uint32_t *branch_rank, const uint32_t *twig_rank,
                                              uint32_t trees, uint32_t twigs, uint32_t *tree, Cell *next, Cell *brick_jobs)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Cell e = frontier[i];
    uint32_t w = word[i];
    const uint32_t type = node_type(w);
    if (type == BRANCH) {
        const uint32_t first = trees + 8 * branch_rank[i];
        w = node_make(BRANCH, first);
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const float ox = (c & 1) ? 1.0f : 0.0f, oy = (c & 2) ? 1.0f : 0.0f, oz = (c & 4) ? 1.0f : 0.0f;
            Cell ch; ch.x = e.x + ox * half; ch.y = e.y + oy * half; ch.z = e.z + oz * half; ch.slot = first + c;
            next[8 * (uint64_t)branch_rank[i] + c] = ch;
        }
    } else if (type == TWIG) {
        const uint32_t brick = twigs + twig_rank[i];
        w = node_make(TWIG, brick);
        Cell job = e; job.slot = brick;
        brick_jobs[twig_rank[i]] = job;
    }
    tree[e.slot] = w;
}

// Bricks (src/Octree.cpp:122-147), one thread per (brick, z-row): the row's four column heights are looked up with
// the reference's float operations (one lookup per column serves its four y-layers, :131-144), then the row's 16
// cells (index z*16 + y*4 + x) go out as one 32-byte store.  4 threads per brick, 128 B per brick contiguous.
__global__ __launch_bounds__(256) void k_bricks_rows(const Cell *jobs, uint32_t n, GrowArgs G, DevPyramid P, uint16_t *twig)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 4u) return;
    const uint32_t j = i >> 2, z = i & 3u;
    const Cell e = jobs[j];
    const float px = (e.x - G.px) / G.size, py = (e.y - G.py) / G.size, pz = (e.z - G.pz) / G.size;
    const float voxel = G.edge / (float)(1 << TWIG_LEVELS);
    const float dz = ((float)z * voxel) / G.size;
    const uint32_t mat = d_height_material(py);
    float h[4];
#pragma unroll
    for (uint32_t x = 0; x < 4; ++x) {
        const float dx = ((float)x * voxel) / G.size;
        h[x] = d_bound(P, P.hi, px + dx, pz + dz, G.level + TWIG_LEVELS);
    }
    uint32_t w[8];                                   // cells (y, x) of this z-row: index y*4 + x, two per dword
#pragma unroll
    for (uint32_t y = 0; y < 4; ++y) {
        const float floor_y = e.y + (float)y * voxel;
        const uint32_t c0 = (h[0] >= floor_y) ? mat : 0u, c1 = (h[1] >= floor_y) ? mat : 0u;
        const uint32_t c2 = (h[2] >= floor_y) ? mat : 0u, c3 = (h[3] >= floor_y) ? mat : 0u;
        w[2 * y] = c0 | (c1 << 16);
        w[2 * y + 1] = c2 | (c3 << 16);
    }
    uint4 *dst = reinterpret_cast<uint4 *>(twig + (uint64_t)e.slot * TWIG_WORDS + z * 16);
    dst[0] = make_uint4(w[0], w[1], w[2], w[3]);
    dst[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int reserve(size_t n, bool keep, hipStream_t s)
    {
        if (n <= cap) return SVO_OK;
        size_t nc = std::max(n, cap * 2);
        T *q = nullptr;
        if (hipMalloc((void **)&q, nc * sizeof(T)) != hipSuccess) { set_error("device builder: hipMalloc failed"); return SVO_ERR_OUT_OF_MEMORY; }
        if (keep && p && cap) { if (hipMemcpyAsync(q, p, cap * sizeof(T), hipMemcpyDeviceToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) { (void)hipFree(q); return SVO_ERR_HIP; } }
        if (p) (void)hipFree(p);
        p = q; cap = nc;
        return SVO_OK;
    }
};

inline unsigned blocks_for(uint64_t n, unsigned per) { return (unsigned)((n + per - 1) / per); }

} // namespace

// One chunk column's pyramid on the device.
struct DevicePyramidBuilder {
    DevBuf<float> lo, hi;
    DevPyramid view{};
    int build(uint32_t res, float ampl, float period, float xshift, float yshift, float zshift, hipStream_t s)
    {
        uint32_t levels = 0;
        while ((1u << levels) < res) ++levels;
        const size_t total = HeightPyramid::level_offset(levels + 1);
        int rc;
        if ((rc = lo.reserve(total, false, s)) != SVO_OK || (rc = hi.reserve(total, false, s)) != SVO_OK) return rc;
        float *blo = lo.p + HeightPyramid::level_offset(levels), *bhi = hi.p + HeightPyramid::level_offset(levels);
        hipLaunchKernelGGL(k_noise_base, dim3(blocks_for((uint64_t)res * res, 256)), dim3(256), 0, s, blo, bhi, res, period, xshift, zshift);
        for (uint32_t lv = levels; lv > 0; --lv) {
            const uint32_t sdim = 1u << lv;
            hipLaunchKernelGGL(k_mip_level, dim3(blocks_for((uint64_t)(sdim / 2) * (sdim / 2), 256)), dim3(256), 0, s,
                               lo.p + HeightPyramid::level_offset(lv), hi.p + HeightPyramid::level_offset(lv),
                               lo.p + HeightPyramid::level_offset(lv - 1), hi.p + HeightPyramid::level_offset(lv - 1), sdim);
        }
        BUILD_TRY(hipGetLastError());
        view.lo = lo.p; view.hi = hi.p; view.size = res; view.levels = levels; view.amplitude = ampl; view.shift = yshift;
        return SVO_OK;
    }
};

struct DeviceGrower {
    DevBuf<Cell> frontier, next, jobs;
    DevBuf<uint32_t> word, is_branch, is_twig, branch_rank, twig_rank, tree;
    DevBuf<uint16_t> twig;
    DevBuf<unsigned char> scan_tmp;

    // The node words go to the host copy (the water fill edits them there); the bricks stay in HBM: *bricks_dev receives the
    // device array (caller owns it, hipFree), c.twigs_on_device their number.
    int grow(ChunkPools &c, const float position[3], float size, uint32_t depth, const DevPyramid &P, const TerrainParams &tp, hipStream_t s,
             uint16_t **bricks_dev)
    {
        c.position[0] = position[0]; c.position[1] = position[1]; c.position[2] = position[2];
        c.size = size; c.depth = depth;
        c.tree_capacity = 16; c.twig_capacity = 16;
        uint64_t trees = 1, twigs = 0;
        int rc;
        if ((rc = frontier.reserve(1, false, s)) != SVO_OK || (rc = tree.reserve(1024, false, s)) != SVO_OK) return rc;
        const Cell root = { position[0], position[1], position[2], 0u };
        BUILD_TRY(hipMemcpyAsync(frontier.p, &root, sizeof root, hipMemcpyHostToDevice, s));
        uint32_t n = 1;
        float edge = size;
        GrowArgs G{};
        G.px = position[0]; G.py = position[1]; G.pz = position[2]; G.size = size; G.depth = depth;
        const bool coarse = tp.coarse_depth >= TWIG_LEVELS && tp.coarse_depth < depth;
        G.coarse_depth = coarse ? tp.coarse_depth : 0;
        for (int a = 0; a < 3; ++a) { G.rmin[a] = tp.refine_min[a]; G.rmax[a] = tp.refine_max[a]; }

        for (uint32_t level = 0; n > 0; ++level) {
            const float half = edge / 2;
            G.level = level; G.edge = edge;
            if ((rc = word.reserve(n, false, s)) != SVO_OK || (rc = is_branch.reserve(n, false, s)) != SVO_OK || (rc = is_twig.reserve(n, false, s)) != SVO_OK ||
                (rc = branch_rank.reserve(n, false, s)) != SVO_OK || (rc = twig_rank.reserve(n, false, s)) != SVO_OK) return rc;
            hipLaunchKernelGGL(k_classify, dim3(blocks_for(n, 256)), dim3(256), 0, s, frontier.p, n, G, P, word.p, is_branch.p, is_twig.p);
            size_t need = 0;
            BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, is_branch.p, branch_rank.p, (int)n, s));
            if ((rc = scan_tmp.reserve(need + 16, false, s)) != SVO_OK) return rc;
            BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, need, is_branch.p, branch_rank.p, (int)n, s));
            BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, need, is_twig.p, twig_rank.p, (int)n, s));
            uint32_t tail[4];       // last rank + last flag of both scans
            BUILD_TRY(hipMemcpyAsync(&tail[0], branch_rank.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipMemcpyAsync(&tail[1], is_branch.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipMemcpyAsync(&tail[2], twig_rank.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipMemcpyAsync(&tail[3], is_twig.p + (n - 1), 4, hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipStreamSynchronize(s));
            const uint64_t nb = (uint64_t)tail[0] + tail[1], nt = (uint64_t)tail[2] + tail[3];
            if (trees + 8 * nb >= (1ull << 30) || twigs + nt >= (1ull << 30)) { set_error("device builder: chunk exceeds the 30-bit node offset"); return SVO_ERR_UNSUPPORTED; }
            if ((rc = tree.reserve(trees + 8 * nb, true, s)) != SVO_OK || (rc = twig.reserve((twigs + nt) * TWIG_WORDS, true, s)) != SVO_OK ||
                (rc = next.reserve(std::max<uint64_t>(8 * nb, 1), false, s)) != SVO_OK || (rc = jobs.reserve(std::max<uint64_t>(nt, 1), false, s)) != SVO_OK) return rc;
            hipLaunchKernelGGL(k_emit, dim3(blocks_for(n, 256)), dim3(256), 0, s, frontier.p, n, half, word.p, branch_rank.p, twig_rank.p,
                               (uint32_t)trees, (uint32_t)twigs, tree.p, next.p, jobs.p);
            if (nt) hipLaunchKernelGGL(k_bricks_rows, dim3(blocks_for(nt * 4, 256)), dim3(256), 0, s, jobs.p, (uint32_t)nt, G, P, twig.p);
            BUILD_TRY(hipGetLastError());
            // capacity bookkeeping exactly as the host builder (src/Octree.cpp:149-150,160-161)
            if (nb) while (trees + 8 * nb >= c.tree_capacity) c.tree_capacity *= 2;
            while (twigs + nt > c.twig_capacity) c.twig_capacity *= 2;
            trees += 8 * nb; twigs += nt;
            std::swap(frontier.p, next.p); std::swap(frontier.cap, next.cap);
            n = (uint32_t)(8 * nb);
            edge = half;
        }
        c.tree.resize(trees);
        c.twig.clear();
        BUILD_TRY(hipMemcpyAsync(c.tree.data(), tree.p, trees * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        BUILD_TRY(hipStreamSynchronize(s));
        c.twigs_on_device = twigs;
        *bricks_dev = twig.p;                       // hand the brick array over; the next chunk gets a fresh one
        twig.p = nullptr; twig.cap = 0;
        return SVO_OK;
    }
};

static int positive_mod_b(int n, int m) { return (m + (n % m)) % m; }

// Ocroot::build on bricks that live in the pool (fill_box_plan recorded which): cell empty and its voxel box touches the
// region -> material (src/Octree.cpp:395-410; cubesIntersect on closed boxes, the host Filler's expressions).
struct DevBrickOp { uint64_t brick; float x, y, z, voxel; };
static_assert(sizeof(DevBrickOp) == sizeof(BrickOp), "BrickOp is uploaded as it is");
__global__ __launch_bounds__(256) void k_brick_fill(uint16_t *twig, const DevBrickOp *ops, uint32_t n, uint64_t first_brick,
                                                    float rlx, float rly, float rlz, float rhx, float rhy, float rhz, uint32_t material)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 64u) return;
    const DevBrickOp op = ops[i >> 6];
    const uint32_t cell = i & 63u, cx = cell & 3u, cy = (cell >> 2) & 3u, cz = cell >> 4;
    uint16_t *p = twig + (first_brick + op.brick) * TWIG_WORDS + cell;
    if (*p != 0) return;
    const float lx = op.x + (float)cx * op.voxel, ly = op.y + (float)cy * op.voxel, lz = op.z + (float)cz * op.voxel;
    const float hx = lx + op.voxel, hy = ly + op.voxel, hz = lz + op.voxel;
    const bool touch = hx >= rlx && hy >= rly && hz >= rlz && rhx >= lx && rhy >= ly && rhz >= lz;
    if (touch) *p = (uint16_t)material;
}

// World::init on the device, pools left in HBM: noise, mips and grow() as kernels (above); the node words visit the host
// for the water fill (Ocroot::build appends depth-first: order-dependent, 25 MB per depth-12 chunk), the bricks - 10x the
// bytes - never leave the device: the fill's brick edits are applied in place by k_brick_fill.  The pools are packed
// exactly as svo_world_upload packs them; the world is uploaded to `device` when this returns.
static int generate_world_resident_impl(svo_world &w, int device)
{
    const bool timing = std::getenv("SVO_BUILD_TIMING") != nullptr;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    auto lapt = [&](const char *what) { if (timing) { (void)hipDeviceSynchronize(); const double t1 = now(); std::fprintf(stderr, "[svo build] %-28s %.1f ms\n", what, (t1 - t0) * 1e3); t0 = t1; } };
    const TerrainParams &tp = w.terrain;
    const int gw = w.width, gh = w.height, gd = w.depth, chunksize = w.chunksize;
    const int *ccm = w.chunkcoordmin;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("svo_world_generate: no HIP device for the device builder"); return SVO_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("svo_world_generate: build_device out of range"); return SVO_ERR_INVALID_ARG; }
    BUILD_TRY(hipSetDevice(device));
    std::vector<ChunkPools> &chunks = w.chunks;
    chunks.assign((size_t)gw * gh * gd, ChunkPools());
    std::vector<uint16_t *> bricks(chunks.size(), nullptr);             // per chunk: its bricks as grow() left them in HBM
    std::vector<uint64_t> grown(chunks.size(), 0);
    std::vector<std::vector<BrickOp>> ops(chunks.size());
    struct Cleanup { std::vector<uint16_t *> &b; ~Cleanup() { for (uint16_t *p : b) if (p) (void)hipFree(p); } } cleanup{ bricks };
    const uint32_t res = tp.pyramid_resolution ? tp.pyramid_resolution : (1u << tp.depth);
    hipStream_t s = nullptr;
    {
        DevicePyramidBuilder pyr;
        DeviceGrower grower;
        for (int zi = 0; zi < gd; ++zi)
            for (int xi = 0; xi < gw; ++xi) {
                const int cx = ccm[0] + xi, cz = ccm[2] + zi;
                int rc = pyr.build(res, tp.amplitude, 1.0f / (float)res, (float)cx * (float)res + (float)tp.seed, tp.yshift,
                                   (float)cz * (float)res + (float)tp.seed, s);
                if (rc != SVO_OK) return rc;
                for (int yi = 0; yi < gh; ++yi) {
                    const int cy = ccm[1] + yi;
                    const int idx = positive_mod_b(cy, gh) * gw * gd + positive_mod_b(cz, gd) * gw + positive_mod_b(cx, gw);
                    ChunkPools &c = chunks[(size_t)idx];
                    const float pos[3] = { (float)cx * (float)chunksize, (float)cy * (float)chunksize, (float)cz * (float)chunksize };
                    rc = grower.grow(c, pos, (float)chunksize, tp.depth, pyr.view, tp, s, &bricks[(size_t)idx]);
                    if (rc != SVO_OK) return rc;
                    grown[(size_t)idx] = c.twigs_on_device;
                }
            }
    }
    lapt("noise + mips + grow (device)");
    if (tp.water) {     // Ocroot::build on the node words (host threads, all chunks in parallel); brick edits are recorded
        int nthreads = tp.threads > 0 ? tp.threads : (int)std::thread::hardware_concurrency();
        nthreads = std::max(1, std::min<int>(nthreads, (int)chunks.size()));
        std::atomic<size_t> cursor{ 0 };
        std::atomic<int> failed{ 0 };           // nothing may escape a std::thread (std::terminate): report after join
        auto worker = [&]() {
            try {
                for (;;) {
                    const size_t i = cursor.fetch_add(1);
                    if (i >= chunks.size()) break;
                    ChunkPools &c = chunks[i];
                    const float hi[3] = { c.position[0] + c.size, tp.water_level, c.position[2] + c.size };
                    fill_box_plan(c, c.position, hi, (uint16_t)tp.water_material, ops[i]);
                }
            } catch (...) {
                failed.store(1);
                cursor.store(chunks.size());
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
        if (failed.load()) { set_error("svo_world_generate: out of host memory in the water fill"); return SVO_ERR_OUT_OF_MEMORY; }
    }
    lapt("water fill plan (host)");
    // pack: the layout of svo_world_upload
    int rc = plan_pools(w);
    if (rc != SVO_OK) return rc;
    if ((rc = alloc_pools(w, device)) != SVO_OK) return rc;
    lapt("alloc pools");
    DevBuf<DevBrickOp> d_ops;
    for (size_t i = 0; i < chunks.size(); ++i) {
        ChunkPools &c = chunks[i];
        const DevChunk &e = w.table[i];
        BUILD_TRY(hipMemcpyAsync(w.d_tree + e.tree_off, c.tree.data(), c.tree.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        const uint64_t before = grown[i], now = c.twig_count();
        uint16_t *slot = w.d_twig + e.twig_off * TWIG_WORDS;
        if (before) BUILD_TRY(hipMemcpyAsync(slot, bricks[i], before * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        if (now > before) BUILD_TRY(hipMemsetAsync(slot + before * TWIG_WORDS, 0, (now - before) * TWIG_WORDS * sizeof(uint16_t), s));
        if (!ops[i].empty()) {
            if ((rc = d_ops.reserve(ops[i].size(), false, s)) != SVO_OK) return rc;
            BUILD_TRY(hipMemcpyAsync(d_ops.p, ops[i].data(), ops[i].size() * sizeof(BrickOp), hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(k_brick_fill, dim3(blocks_for((uint64_t)ops[i].size() * 64, 256)), dim3(256), 0, s, w.d_twig,
                               d_ops.p, (uint32_t)ops[i].size(), (uint64_t)e.twig_off,
                               c.position[0], c.position[1], c.position[2], c.position[0] + c.size, tp.water_level, c.position[2] + c.size,
                               (uint32_t)tp.water_material);
            BUILD_TRY(hipGetLastError());
            BUILD_TRY(hipStreamSynchronize(s));             // d_ops and ops[i] are reused / freed
        }
        if ((rc = launch_brick_masks(w, e.twig_off, now, s)) != SVO_OK) return rc;
        BUILD_TRY(hipStreamSynchronize(s));
        (void)hipFree(bricks[i]); bricks[i] = nullptr;
    }
    lapt("pack + brick fill + masks");
    BUILD_TRY(hipMemcpy(w.d_chunks, w.table.data(), chunks.size() * sizeof(DevChunk), hipMemcpyHostToDevice));
    if ((rc = build_wide_all(w, s)) != SVO_OK) return rc;
    lapt("wide trees");
    BUILD_TRY(hipDeviceSynchronize());
    return SVO_OK;
}

int generate_world_resident(svo_world &w, int device)
{
    try {
        const int rc = generate_world_resident_impl(w, device);
        if (rc != SVO_OK) release_device(w);
        return rc;
    }
    catch (const std::bad_alloc &) { release_device(w); set_error("svo_world_generate (device builder): out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
}

} // namespace svo
