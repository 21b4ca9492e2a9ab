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
//   Ocroot::build (the water plane)   src/Octree.cpp:320-436         -> DeviceFiller below: the depth-first fill as three
//        level-synchronous sweeps (classify top-down, count bottom-up, number top-down); the node words never visit the host
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

// Consecutive values from *ctr for the threads of a block that raise `pred` - ONE global atomic per block.  The level-synchronous
// sweeps number millions of nodes through a handful of counters, and same-address atomics are served one after the other (≈ 5.7 ns
// each here, even at the one per wave the compiler already folds a uniform atomicAdd to: a sweep over 4.8 M nodes took 0.86 ms).
// Every thread of the block calls this (no early return before it); sh holds FILL_BLOCK / 64 + 1 words.
constexpr unsigned FILL_BLOCK = 1024;
__device__ __forceinline__ uint32_t block_take(uint32_t *ctr, bool pred, uint32_t *sh)
{
    constexpr unsigned WAVES = FILL_BLOCK / 64;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const unsigned long long m = __ballot(pred);
    if (lane == 0u) sh[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0u) {
        uint32_t tot = 0u;
        for (unsigned w = 0; w < WAVES; ++w) { const uint32_t c = sh[w]; sh[w] = tot; tot += c; }
        sh[WAVES] = tot ? atomicAdd(ctr, tot) : 0u;
    }
    __syncthreads();
    const uint32_t r = sh[WAVES] + sh[wv] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();                                        // (sh may serve a second call)
    return r;
}

struct GrowArgs {
    float px, py, pz, size;     // chunk position / edge
    float edge;                 // node edge at this level
    uint32_t level, depth;
    uint32_t coarse_depth;      // 0 = off
    float rmin[3], rmax[3];     // refine box
};

// type of every frontier node (src/Octree.cpp:105-121) + its flags for the scan (BRANCH in the low half, TWIG in the high half of
// one 64-bit word: one scan ranks both; the level's totals follow from the scan, k_level_totals)
__global__ __launch_bounds__(256) void k_classify(const Cell *frontier, uint32_t n, GrowArgs G, DevPyramid P,
                                                  uint32_t *word, unsigned long long *flags)
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
    flags[i] = (unsigned long long)br | ((unsigned long long)tw << 32);
}

// the level's totals, from the scan instead of a counter: exclusive rank of the last node + its own flags
__global__ void k_level_totals(const unsigned long long *flags, const unsigned long long *rank, uint32_t n, uint32_t *totals /* [0] BRANCH, [1] TWIG */)
{
    const unsigned long long t = rank[n - 1] + flags[n - 1];
    totals[0] = (uint32_t)t;
    totals[1] = (uint32_t)(t >> 32);
}

// node words; children of every BRANCH appended to the next frontier in parent order (== FIFO queue order);
// brick jobs listed in TWIG order
__global__ __launch_bounds__(256) void k_emit(const Cell *frontier, uint32_t n, float half, const uint32_t *word,
                                              const unsigned long long *rank /* BRANCH rank | TWIG rank << 32 */,
                                              uint32_t trees, uint32_t twigs, uint32_t *tree, Cell *next, Cell *brick_jobs)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Cell e = frontier[i];
    uint32_t w = word[i];
    const uint32_t type = node_type(w);
    if (type == BRANCH) {
        const uint32_t branch_rank = (uint32_t)rank[i];
        const uint32_t first = trees + 8 * branch_rank;
        w = node_make(BRANCH, first);
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const float ox = (c & 1) ? 1.0f : 0.0f, oy = (c & 2) ? 1.0f : 0.0f, oz = (c & 4) ? 1.0f : 0.0f;
            Cell ch; ch.x = e.x + ox * half; ch.y = e.y + oy * half; ch.z = e.z + oz * half; ch.slot = first + c;
            next[8 * (uint64_t)branch_rank + c] = ch;
        }
    } else if (type == TWIG) {
        const uint32_t twig_rank = (uint32_t)(rank[i] >> 32);
        const uint32_t brick = twigs + twig_rank;
        w = node_make(TWIG, brick);
        Cell job = e; job.slot = brick;
        brick_jobs[twig_rank] = job;
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

// ---- Ocroot::build / Ocroot::destroy on the device (src/Octree.cpp:320-436 / :203-318; terrain.cpp's Filler is build's host twin) ----
// The reference edits depth-first in child-slot order and APPENDS as it goes: a node the region cuts - EMPTY for build, LEAF for
// destroy - becomes a BRANCH whose 8-block lands at the pool's tail (a "split"), or - at the brick level - a TWIG whose brick
// lands at the brick pool's tail.  So the index a new block / brick gets is the number of splits / new bricks that precede it
// in depth-first preorder.  Three level-synchronous sweeps over the nodes the region touches give exactly that:
//   A (top-down)   k_fill_classify: what the visit does at each node (its action), the 8 children of every node the visit
//                  descends into appended to the next level's list (any order: the lists only link parents to children)
//   B (bottom-up)  k_fill_count: splits / new bricks in each node's subtree
//   C (top-down)   k_fill_number: preorder prefix of those counts = each event's index; node words written, new blocks
//                  initialised (EMPTY for build, the split LEAF's material for destroy), brick edits listed for k_brick_edit
// A node the edit creates has no slot until sweep C: until its parent numbers it, its list entry carries FILL_VIRTUAL and what
// the reference would read there (EMPTY, or LEAF + material).
enum FillAction : uint32_t { FILL_NONE = 0, FILL_SET, FILL_NEW_BRICK, FILL_SPLIT, FILL_BRICK, FILL_DESCEND };
static_assert(FILL_DESCEND < (1u << FILL_ACTION_BITS), "every FillAction fits fill_pack's action field");
enum EditOp : uint32_t { EDIT_BUILD = 0, EDIT_DESTROY = 1 };
constexpr uint32_t FILL_VIRTUAL = 0x80000000u, FILL_VIRTUAL_LEAF = 0x00010000u;

struct FillArgs {
    float rlo[3], rhi[3];       // the region (closed box)
    float edge;                 // node edge at this level
    uint32_t level, maxlevel;   // maxlevel = depth - TWIG_LEVELS: nodes cut there become bricks
    uint32_t material;          // build: what the region is filled with
    uint32_t op;                // EDIT_BUILD / EDIT_DESTROY
};

// act[i] = fill_pack(action, kids) (kids = index of the node's child block in the next level's list; svo_format.h)
__global__ __launch_bounds__(FILL_BLOCK) void k_fill_classify(const Cell *cells, uint32_t n, FillArgs F, const uint32_t *tree,
                                                              uint32_t *act, Cell *next, uint32_t *counters /* [0] child blocks, [1] brick edits */)
{
    __shared__ uint32_t sh[FILL_BLOCK / 64 + 1];
    const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
    const bool live = i < n;
    const Cell e = live ? cells[i] : Cell{ 0.0f, 0.0f, 0.0f, 0u };
    const float hx = e.x + F.edge, hy = e.y + F.edge, hz = e.z + F.edge;
    uint32_t a = FILL_NONE, word = node_make(EMPTY, 0);
    // cubesIntersect on closed boxes (src/Traverse.cpp:173-178), the host Filler's expressions
    const bool touch = live && hx >= F.rlo[0] && hy >= F.rlo[1] && hz >= F.rlo[2] && F.rhi[0] >= e.x && F.rhi[1] >= e.y && F.rhi[2] >= e.z;
    if (touch) {
        if (!(e.slot & FILL_VIRTUAL)) word = tree[e.slot];
        else if (e.slot & FILL_VIRTUAL_LEAF) word = node_make(LEAF, e.slot & 0xFFFFu);
        // cubeIsInside (src/Traverse.cpp:180-185)
        const bool inside = e.x >= F.rlo[0] && e.y >= F.rlo[1] && e.z >= F.rlo[2] && F.rhi[0] >= hx && F.rhi[1] >= hy && F.rhi[2] >= hz;
        const uint32_t type = node_type(word);
        if (F.op == EDIT_BUILD) {           // buildCube, src/Octree.cpp:338-430
            if (type == EMPTY) a = inside ? FILL_SET : (F.level == F.maxlevel ? FILL_NEW_BRICK : FILL_SPLIT);
            else if (type == TWIG) a = FILL_BRICK;
            else if (type == BRANCH) a = FILL_DESCEND;
        } else {                            // destroyCube, src/Octree.cpp:220-312: whatever lies inside goes, LEAF nodes the region cuts are split
            if (type == EMPTY) a = FILL_NONE;
            else if (inside) a = FILL_SET;
            else if (type == LEAF) a = F.level == F.maxlevel ? FILL_NEW_BRICK : FILL_SPLIT;
            else if (type == TWIG) a = FILL_BRICK;
            else a = FILL_DESCEND;
        }
    }
    const bool has_kids = a == FILL_SPLIT || a == FILL_DESCEND;
    const uint32_t kids = block_take(&counters[0], has_kids, sh);
    if (has_kids) {
        const float half = F.edge * 0.5f;
        const uint32_t first = node_offset(word);
        const uint32_t inherit = FILL_VIRTUAL | (F.op == EDIT_DESTROY ? FILL_VIRTUAL_LEAF | (node_offset(word) & 0xFFFFu) : 0u);
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const float ox = (c & 1) ? 1.0f : 0.0f, oy = (c & 2) ? 1.0f : 0.0f, oz = (c & 4) ? 1.0f : 0.0f;
            Cell ch; ch.x = e.x + ox * half; ch.y = e.y + oy * half; ch.z = e.z + oz * half;
            ch.slot = a == FILL_DESCEND ? first + c : inherit;
            next[8 * (uint64_t)kids + c] = ch;
        }
    }
    (void)block_take(&counters[1], a == FILL_NEW_BRICK || a == FILL_BRICK, sh);
    if (live) act[i] = fill_pack(a, has_kids ? kids : 0u);
}

// cnt[i] = {splits, new bricks} in the subtree of node i, itself included
__global__ __launch_bounds__(256) void k_fill_count(const uint32_t *act, uint32_t n, const uint2 *cnt_next, uint2 *cnt)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = fill_action(act[i]), kids = fill_kids(act[i]);
    uint2 v = make_uint2(a == FILL_SPLIT ? 1u : 0u, a == FILL_NEW_BRICK ? 1u : 0u);
    if (a == FILL_SPLIT || a == FILL_DESCEND) {
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) { const uint2 k = cnt_next[8 * (uint64_t)kids + c]; v.x += k.x; v.y += k.y; }
    }
    cnt[i] = v;
}

struct DevBrickOp { uint64_t brick; float x, y, z, voxel; uint32_t fresh, init; };     // fresh: a brick this edit appends, every cell starts as `init`

// On entry cnt[i] = {splits, new bricks} that precede node i's own events in preorder (the root: 0, 0); the node's children
// get theirs (their subtree counts are replaced by the running prefix), new nodes get their slots, node words are written.
__global__ __launch_bounds__(FILL_BLOCK) void k_fill_number(Cell *cells, uint32_t n, FillArgs F, const uint32_t *act, const uint2 *cnt,
                                                            Cell *next, uint2 *cnt_next, uint32_t trees0, uint32_t twigs0,
                                                            uint32_t *tree, DevBrickOp *ops, uint32_t *op_cursor)
{
    __shared__ uint32_t sh[FILL_BLOCK / 64 + 1];
    const uint32_t i = blockIdx.x * FILL_BLOCK + threadIdx.x;
    const bool live = i < n;
    const uint32_t a = live ? fill_action(act[i]) : (uint32_t)FILL_NONE, kids = live ? fill_kids(act[i]) : 0u;
    const uint32_t op_slot = block_take(op_cursor, a == FILL_NEW_BRICK || a == FILL_BRICK, sh);     // (any order: k_brick_edit treats the ops alike)
    if (a == FILL_NONE) return;
    const Cell e = cells[i];                // (its slot is a real one by now: the parent's turn came a launch earlier)
    const uint2 base = cnt[i];
    switch (a) {
    case FILL_SET:
        tree[e.slot] = F.op == EDIT_BUILD ? node_make(LEAF, F.material) : node_make(EMPTY, 0);
        break;
    case FILL_NEW_BRICK:
    case FILL_BRICK: {
        const uint32_t word = tree[e.slot];
        DevBrickOp op;
        op.fresh = a == FILL_NEW_BRICK ? 1u : 0u;
        op.init = F.op == EDIT_DESTROY ? node_offset(word) & 0xFFFFu : 0u;        // Octwig(material of the LEAF that is cut) / Octwig(0)
        if (a == FILL_NEW_BRICK) { op.brick = twigs0 + base.y; tree[e.slot] = node_make(TWIG, (uint32_t)op.brick); }
        else op.brick = node_offset(word);
        op.x = e.x; op.y = e.y; op.z = e.z; op.voxel = F.edge / (float)(1 << TWIG_LEVELS);
        ops[op_slot] = op;
        break;
    }
    default: {
        uint2 run = base;
        uint32_t first = 0, inherit = node_make(EMPTY, 0);
        if (a == FILL_SPLIT) {
            if (F.op == EDIT_DESTROY) inherit = tree[e.slot];                     // 8 LEAF children of the same material
            first = trees0 + 8u * base.x;
            tree[e.slot] = node_make(BRANCH, first);
            run.x += 1;
        }
#pragma unroll
        for (uint32_t c = 0; c < 8; ++c) {
            const uint64_t k = 8 * (uint64_t)kids + c;
            if (a == FILL_SPLIT) { tree[first + c] = inherit; next[k].slot = first + c; }
            const uint2 sub = cnt_next[k];
            cnt_next[k] = run;
            run.x += sub.x; run.y += sub.y;
        }
    }
    }
}

// The brick half of both edits (src/Octree.cpp:395-410 / :285-300): build - a cell that is empty and whose voxel box touches the
// region takes the material; destroy - a cell whose voxel box touches the region is emptied (cubesIntersect on closed boxes,
// the host Filler's expressions).
__global__ __launch_bounds__(256) void k_brick_edit(uint16_t *twig, const DevBrickOp *ops, uint32_t n, uint32_t edit,
                                                    float rlx, float rly, float rlz, float rhx, float rhy, float rhz, uint32_t material)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n * 64u) return;
    const DevBrickOp op = ops[i >> 6];
    const uint32_t cell = i & 63u, cx = cell & 3u, cy = (cell >> 2) & 3u, cz = cell >> 4;
    uint16_t *p = twig + op.brick * TWIG_WORDS + cell;
    const uint32_t before = op.fresh ? op.init : (uint32_t)*p;
    const float lx = op.x + (float)cx * op.voxel, ly = op.y + (float)cy * op.voxel, lz = op.z + (float)cz * op.voxel;
    const float hx = lx + op.voxel, hy = ly + op.voxel, hz = lz + op.voxel;
    const bool touch = hx >= rlx && hy >= rly && hz >= rlz && rhx >= lx && rhy >= ly && rhz >= lz;
    uint32_t after = before;
    if (edit == EDIT_BUILD) { if (before == 0u && touch) after = material; }
    else if (touch) after = 0u;
    if (op.fresh || after != before) *p = (uint16_t)after;
}

struct DeviceFiller {
    struct Level { DevBuf<Cell> cells; DevBuf<uint32_t> act; DevBuf<uint2> cnt; uint32_t n = 0; };
    std::vector<Level> lv;
    DevBuf<uint32_t> counters;      // [2 * level + {0, 1}] of sweep A, [64] the op cursor of sweep C
    DevBuf<DevBrickOp> ops;
    uint32_t *h_counters = nullptr; // pinned
    ~DeviceFiller() { if (h_counters) (void)hipHostFree(h_counters); }

    // Ocroot::build (edit = EDIT_BUILD: region [lo, hi] filled with `material`) or Ocroot::destroy (EDIT_DESTROY: emptied) applied
    // to the chunk in `tree` (trees nodes) and `twig` (twigs bricks); both buffers grow as needed, c's capacities follow the
    // reference's doubling.
    int fill(ChunkPools &c, const float lo[3], const float hi[3], uint32_t material, DevBuf<uint32_t> &tree, uint64_t &trees,
             DevBuf<uint16_t> &twig, uint64_t &twigs, hipStream_t s, uint32_t edit = EDIT_BUILD)
    {
        int rc;
        const uint32_t maxlevel = c.depth - TWIG_LEVELS;
        if (lv.size() < maxlevel + 2) lv.resize(maxlevel + 2);
        if ((rc = counters.reserve(80, false, s)) != SVO_OK) return rc;
        if (!h_counters) BUILD_TRY(hipHostMalloc((void **)&h_counters, 80 * sizeof(uint32_t)));
        BUILD_TRY(hipMemsetAsync(counters.p, 0, 80 * sizeof(uint32_t), s));
        FillArgs F{};
        for (int a = 0; a < 3; ++a) { F.rlo[a] = lo[a]; F.rhi[a] = hi[a]; }
        F.maxlevel = maxlevel; F.material = material; F.op = edit;
        // sweep A
        if ((rc = lv[0].cells.reserve(1, false, s)) != SVO_OK) return rc;
        const Cell root = { c.position[0], c.position[1], c.position[2], 0u };
        BUILD_TRY(hipMemcpyAsync(lv[0].cells.p, &root, sizeof root, hipMemcpyHostToDevice, s));
        lv[0].n = 1;
        float edge = c.size;
        uint32_t last = 0, brick_edits = 0;
        std::vector<float> edges(maxlevel + 1);
        for (uint32_t level = 0; level <= maxlevel; ++level) {
            Level &L = lv[level], &N = lv[level + 1];
            const uint32_t n = L.n;
            edges[level] = edge;
            last = level;
            if ((rc = L.act.reserve(n, false, s)) != SVO_OK || (rc = L.cnt.reserve(n, false, s)) != SVO_OK ||
                (rc = N.cells.reserve((uint64_t)n * 8, false, s)) != SVO_OK) return rc;
            F.level = level; F.edge = edge;
            hipLaunchKernelGGL(k_fill_classify, dim3(blocks_for(n, FILL_BLOCK)), dim3(FILL_BLOCK), 0, s, L.cells.p, n, F, tree.p, L.act.p, N.cells.p, counters.p + 2 * level);
            BUILD_TRY(hipGetLastError());
            BUILD_TRY(hipMemcpyAsync(h_counters, counters.p + 2 * level, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipStreamSynchronize(s));
            brick_edits += h_counters[1];
            // (also what fill_pack can hold: child-block indices stay below FILL_KIDS_LIMIT = 2^28)
            if ((uint64_t)h_counters[0] >= FILL_KIDS_LIMIT) { set_error("device builder: the fill's frontier exceeds 2^31 nodes"); return SVO_ERR_UNSUPPORTED; }
            N.n = h_counters[0] * 8u;
            edge = edge * 0.5f;
            if (N.n == 0) break;
            // a node at maxlevel is never split and grow() puts no BRANCH there
            if (level == maxlevel) { set_error("device builder: BRANCH below level depth-2"); return SVO_ERR_MALFORMED_TREE; }
        }
        // sweep B
        for (int level = (int)last; level >= 0; --level) {
            Level &L = lv[(size_t)level], &N = lv[(size_t)level + 1];
            hipLaunchKernelGGL(k_fill_count, dim3(blocks_for(L.n, 256)), dim3(256), 0, s, L.act.p, L.n, N.cnt.p, L.cnt.p);
        }
        BUILD_TRY(hipGetLastError());
        uint2 total;
        BUILD_TRY(hipMemcpyAsync(h_counters, lv[0].cnt.p, sizeof(uint2), hipMemcpyDeviceToHost, s));
        BUILD_TRY(hipStreamSynchronize(s));
        total.x = h_counters[0]; total.y = h_counters[1];
        const uint64_t trees1 = trees + 8ull * total.x, twigs1 = twigs + total.y;
        if (trees1 >= (1ull << 30) || twigs1 >= (1ull << 30)) { set_error("device builder: chunk exceeds the 30-bit node offset"); return SVO_ERR_UNSUPPORTED; }
        if ((rc = tree.reserve(trees1, true, s)) != SVO_OK || (rc = twig.reserve(std::max<uint64_t>(twigs1, 1) * TWIG_WORDS, true, s)) != SVO_OK ||
            (rc = ops.reserve(std::max<uint32_t>(brick_edits, 1u), false, s)) != SVO_OK) return rc;
        if (total.y) BUILD_TRY(hipMemsetAsync(twig.p + twigs * TWIG_WORDS, 0, (size_t)total.y * TWIG_WORDS * sizeof(uint16_t), s));
        BUILD_TRY(hipMemsetAsync(lv[0].cnt.p, 0, sizeof(uint2), s));
        // sweep C
        for (uint32_t level = 0; level <= last; ++level) {
            Level &L = lv[level], &N = lv[level + 1];
            F.level = level; F.edge = edges[level];
            hipLaunchKernelGGL(k_fill_number, dim3(blocks_for(L.n, FILL_BLOCK)), dim3(FILL_BLOCK), 0, s, L.cells.p, L.n, F, L.act.p, L.cnt.p, N.cells.p, N.cnt.p,
                               (uint32_t)trees, (uint32_t)twigs, tree.p, ops.p, counters.p + 64);
        }
        if (brick_edits)
            hipLaunchKernelGGL(k_brick_edit, dim3(blocks_for((uint64_t)brick_edits * 64, 256)), dim3(256), 0, s, twig.p, ops.p, brick_edits, edit,
                               F.rlo[0], F.rlo[1], F.rlo[2], F.rhi[0], F.rhi[1], F.rhi[2], material);
        BUILD_TRY(hipGetLastError());
        // capacity bookkeeping of the reference's appends (src/Octree.cpp:349-351,365-368; terrain.cpp's Filler)
        if (total.x) while (trees1 >= c.tree_capacity) c.tree_capacity *= 2;
        while (twigs1 > c.twig_capacity) c.twig_capacity *= 2;
        trees = trees1; twigs = twigs1;
        return SVO_OK;
    }
};

struct DeviceGrower {
    DevBuf<Cell> frontier, next, jobs;
    DevBuf<uint32_t> word, tree, totals;
    DevBuf<unsigned long long> flags, rank;
    DevBuf<uint16_t> twig;
    DevBuf<unsigned char> scan_tmp;
    uint32_t *h_totals = nullptr;               // pinned
    ~DeviceGrower() { if (h_totals) (void)hipHostFree(h_totals); }

    uint64_t hint_tree = 1024, hint_twig = 0;   // what the previous chunk needed: the next one starts there instead of doubling its way up

    // grow() and - if the terrain has water - Ocroot::build behind it, both on the device.  Node words and bricks stay in HBM:
    // *tree_dev / *bricks_dev receive the device arrays (caller owns them, hipFree), c.trees_on_device / c.twigs_on_device
    // their lengths; the host copies are fetched on request (device.hip: fetch_pools).
    int grow(ChunkPools &c, const float position[3], float size, uint32_t depth, const DevPyramid &P, const TerrainParams &tp, hipStream_t s,
             DeviceFiller &filler, uint32_t **tree_dev, uint16_t **bricks_dev)
    {
        c.position[0] = position[0]; c.position[1] = position[1]; c.position[2] = position[2];
        c.size = size; c.depth = depth;
        c.tree_capacity = 16; c.twig_capacity = 16;
        uint64_t trees = 1, twigs = 0;
        int rc;
        if ((rc = frontier.reserve(1, false, s)) != SVO_OK || (rc = tree.reserve(hint_tree, false, s)) != SVO_OK ||
            (hint_twig && (rc = twig.reserve(hint_twig, false, s)) != SVO_OK) || (rc = totals.reserve(64, false, s)) != SVO_OK) return rc;
        if (!h_totals) BUILD_TRY(hipHostMalloc((void **)&h_totals, 2 * sizeof(uint32_t)));
        BUILD_TRY(hipMemsetAsync(totals.p, 0, 64 * sizeof(uint32_t), s));
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
            if (level >= 32) { set_error("device builder: more than 32 levels"); return SVO_ERR_UNSUPPORTED; }
            if ((rc = word.reserve(n, false, s)) != SVO_OK || (rc = flags.reserve(n, false, s)) != SVO_OK || (rc = rank.reserve(n, false, s)) != SVO_OK) return rc;
            hipLaunchKernelGGL(k_classify, dim3(blocks_for(n, 256)), dim3(256), 0, s, frontier.p, n, G, P, word.p, flags.p);
            size_t need = 0;
            BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, need, flags.p, rank.p, (int)n, s));
            if ((rc = scan_tmp.reserve(need + 16, false, s)) != SVO_OK) return rc;
            BUILD_TRY(hipcub::DeviceScan::ExclusiveSum(scan_tmp.p, need, flags.p, rank.p, (int)n, s));
            hipLaunchKernelGGL(k_level_totals, dim3(1), dim3(1), 0, s, flags.p, rank.p, n, totals.p + 2 * level);
            BUILD_TRY(hipMemcpyAsync(h_totals, totals.p + 2 * level, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            BUILD_TRY(hipStreamSynchronize(s));
            const uint64_t nb = h_totals[0], nt = h_totals[1];
            if (trees + 8 * nb >= (1ull << 30) || twigs + nt >= (1ull << 30)) { set_error("device builder: chunk exceeds the 30-bit node offset"); return SVO_ERR_UNSUPPORTED; }
            if ((rc = tree.reserve(trees + 8 * nb, true, s)) != SVO_OK || (rc = twig.reserve((twigs + nt) * TWIG_WORDS, true, s)) != SVO_OK ||
                (rc = next.reserve(std::max<uint64_t>(8 * nb, 1), false, s)) != SVO_OK || (rc = jobs.reserve(std::max<uint64_t>(nt, 1), false, s)) != SVO_OK) return rc;
            hipLaunchKernelGGL(k_emit, dim3(blocks_for(n, 256)), dim3(256), 0, s, frontier.p, n, half, word.p, rank.p,
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
        if (tp.water) {     // World::g_chunk, src/World.cpp:316-320: everything of the chunk below the water level
            const float hi[3] = { position[0] + size, tp.water_level, position[2] + size };
            if ((rc = filler.fill(c, position, hi, tp.water_material, tree, trees, twig, twigs, s)) != SVO_OK) return rc;
        }
        c.tree.clear(); c.twig.clear();
        c.trees_on_device = trees;
        c.twigs_on_device = twigs;
        hint_tree = std::max<uint64_t>(hint_tree, tree.cap); hint_twig = std::max<uint64_t>(hint_twig, twig.cap);
        *tree_dev = tree.p; *bricks_dev = twig.p;   // hand the arrays over; the next chunk gets fresh ones
        tree.p = nullptr; tree.cap = 0;
        twig.p = nullptr; twig.cap = 0;
        return SVO_OK;
    }
};

static int positive_mod_b(int n, int m) { return (m + (n % m)) % m; }

// World::init on the device, pools left in HBM: noise, mips, grow() and the water fill (Ocroot::build) as kernels (above).
// Neither node words nor bricks visit the host (fetched on request: device.hip, fetch_pools).  The pools are packed exactly as
// svo_world_upload packs them; the world is uploaded to `device` when this returns.
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
    std::vector<uint16_t *> bricks(chunks.size(), nullptr);             // per chunk: its pools as the builder left them in HBM
    std::vector<uint32_t *> trees(chunks.size(), nullptr);
    struct Cleanup {
        std::vector<uint16_t *> &b; std::vector<uint32_t *> &t;
        ~Cleanup() { for (uint16_t *p : b) if (p) (void)hipFree(p); for (uint32_t *p : t) if (p) (void)hipFree(p); }
    } cleanup{ bricks, trees };
    const uint32_t res = tp.pyramid_resolution ? tp.pyramid_resolution : (1u << tp.depth);
    hipStream_t s = nullptr;
    {
        DevicePyramidBuilder pyr;
        DeviceGrower grower;
        DeviceFiller filler;
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
                    rc = grower.grow(c, pos, (float)chunksize, tp.depth, pyr.view, tp, s, filler, &trees[(size_t)idx], &bricks[(size_t)idx]);
                    if (rc != SVO_OK) return rc;
                }
            }
        lapt("noise + mips + grow + fill");
    }
    // pack: the layout of svo_world_upload
    int rc = plan_pools(w);
    if (rc != SVO_OK) return rc;
    if ((rc = alloc_pools(w, device)) != SVO_OK) return rc;
    lapt("alloc pools");
    for (size_t i = 0; i < chunks.size(); ++i) {
        const ChunkPools &c = chunks[i];
        const DevChunk &e = w.table[i];
        BUILD_TRY(hipMemcpyAsync(w.d_tree + e.tree_off, trees[i], c.trees_on_device * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        if (c.twigs_on_device)
            BUILD_TRY(hipMemcpyAsync(w.d_twig + e.twig_off * TWIG_WORDS, bricks[i], c.twigs_on_device * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
        if ((rc = launch_brick_masks(w, e.twig_off, c.twigs_on_device, s)) != SVO_OK) return rc;
    }
    BUILD_TRY(hipMemcpyAsync(w.d_chunks, w.table.data(), chunks.size() * sizeof(DevChunk), hipMemcpyHostToDevice, s));
    BUILD_TRY(hipStreamSynchronize(s));
    for (size_t i = 0; i < chunks.size(); ++i) {
        (void)hipFree(bricks[i]); bricks[i] = nullptr;
        (void)hipFree(trees[i]); trees[i] = nullptr;
    }
    lapt("pack + masks");
    const bool literal_only = build_wide_all(w, s) != SVO_OK;         // (a complete world either way: see svo_world_upload)
    lapt("wide trees");
    BUILD_TRY(hipDeviceSynchronize());
    return literal_only ? SVO_OK_LITERAL_ONLY : SVO_OK;
}

// What svo_world_shift and svo_world_edit_box keep between calls on an uploaded world (created by the first one, freed with the
// device copy): the builders' working buffers - a few hundred MB for depth-12 chunks - so that an interactive caller's edits and
// slides do not pay some forty hipMalloc / hipFree each.
struct BuilderContext {
    DevicePyramidBuilder pyr;
    DeviceGrower grower;
    DeviceFiller filler;
    DevBuf<uint32_t> edit_tree;
    DevBuf<uint16_t> edit_twig;
};
static BuilderContext &builder_context(svo_world &w)
{
    if (!w.builder_ctx) w.builder_ctx = new BuilderContext();
    return *static_cast<BuilderContext *>(w.builder_ctx);
}
void free_builder_context(svo_world &w)
{
    delete static_cast<BuilderContext *>(w.builder_ctx);
    w.builder_ctx = nullptr;
}

// World::shift (src/World.cpp:334-378) on an uploaded world: the plane of chunks entering the grid is generated on the device
// the pools live on (g_pyramid + g_chunk as in generate_world_resident_impl) and takes the slots of the plane that leaves -
// the toroidal index of a chunk coordinate does not depend on chunkcoordmin -, then chunkcoordmin moves.
static int shift_world_resident_impl(svo_world &w, int axis, int sign)
{
    const TerrainParams &tp = w.terrain;
    const int dims[3] = { w.width, w.height, w.depth };
    const int u = sign < 0 ? w.chunkcoordmin[axis] - 1 : w.chunkcoordmin[axis] + dims[axis];
    const uint32_t res = tp.pyramid_resolution ? tp.pyramid_resolution : (1u << tp.depth);
    int lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = w.chunkcoordmin[a]; hi[a] = w.chunkcoordmin[a] + dims[a]; }
    lo[axis] = u; hi[axis] = u + 1;
    BUILD_TRY(hipSetDevice(w.device));
    hipStream_t s = nullptr;
    BuilderContext &ctx = builder_context(w);
    DevicePyramidBuilder &pyr = ctx.pyr;
    DeviceGrower &grower = ctx.grower;
    DeviceFiller &filler = ctx.filler;
    // The whole entering plane is generated before any of it is installed: a failure on the way (device memory, mostly) leaves
    // the world as it was.  Once the installs have begun they all happen and chunkcoordmin moves - no launch sees a grid whose
    // slots hold chunks of two positions of the window ("none sees a mixture", svo.h); a wide tree that could not be rebuilt on
    // the way only takes the stack kernel away (SVO_OK_LITERAL_ONLY).
    struct Entering { ChunkPools c; uint32_t *tree_dev = nullptr; uint16_t *twig_dev = nullptr; int index = 0; };
    std::vector<Entering> plane;
    auto release = [&plane]() { for (Entering &e : plane) { (void)hipFree(e.tree_dev); (void)hipFree(e.twig_dev); } plane.clear(); };
    for (int cz = lo[2]; cz < hi[2]; ++cz)
        for (int cx = lo[0]; cx < hi[0]; ++cx) {
            int rc = pyr.build(res, tp.amplitude, 1.0f / (float)res, (float)cx * (float)res + (float)tp.seed, tp.yshift,
                               (float)cz * (float)res + (float)tp.seed, s);
            if (rc != SVO_OK) { release(); return rc; }
            for (int cy = lo[1]; cy < hi[1]; ++cy) {
                plane.emplace_back();
                Entering &e = plane.back();
                const float pos[3] = { (float)cx * (float)w.chunksize, (float)cy * (float)w.chunksize, (float)cz * (float)w.chunksize };
                e.index = svo_world_index(&w, cx, cy, cz);
                rc = grower.grow(e.c, pos, (float)w.chunksize, tp.depth, pyr.view, tp, s, filler, &e.tree_dev, &e.twig_dev);
                if (rc != SVO_OK) { release(); return rc; }
            }
        }
    int status = SVO_OK;
    for (Entering &e : plane) {
        const int rc = install_resident_chunk(w, e.index, e.c, e.tree_dev, e.twig_dev);
        if (rc < 0) { release(); return rc; }       // a HIP failure in the middle of the copies: the device is in no state to go on with
        if (rc != SVO_OK) status = rc;
    }
    release();
    w.chunkcoordmin[axis] += sign;
    return status;
}

int shift_world_resident(svo_world &w, int axis, int sign)
{
    try { return shift_world_resident_impl(w, axis, sign); }
    catch (const std::bad_alloc &) { set_error("svo_world_shift: out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
}

// Ocroot::build / destroy / replace + World::modify (src/Octree.cpp:203-443, src/World.cpp:268-274; the caller's pattern is
// src/Main.cpp:340-367) on an uploaded world, without the host: the chunk's pools are copied out of the packed pools, edited by
// the three sweeps above and installed again with svo_world_update's slot logic.  A host copy of the chunk, if there was one,
// is dropped (svo_world_chunk fetches the edited pools on request).
static int edit_box_resident_impl(svo_world &w, int chunk, int op, const float lo[3], const float hi[3], uint32_t material)
{
    BUILD_TRY(hipSetDevice(w.device));
    BUILD_TRY(hipDeviceSynchronize());              // ordered behind every launch issued before it, like svo_world_update
    hipStream_t s = nullptr;
    const ChunkPools &c = w.chunks[(size_t)chunk];
    const DevChunk &e = w.table[(size_t)chunk];
    uint64_t trees = c.tree_count(), twigs = c.twig_count();
    BuilderContext &ctx = builder_context(w);
    DevBuf<uint32_t> &tree = ctx.edit_tree;
    DevBuf<uint16_t> &twig = ctx.edit_twig;
    int rc;
    if ((rc = tree.reserve(trees + 1024, false, s)) != SVO_OK || (rc = twig.reserve((twigs + 16) * TWIG_WORDS, false, s)) != SVO_OK) return rc;
    BUILD_TRY(hipMemcpyAsync(tree.p, w.d_tree + e.tree_off, trees * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (twigs) BUILD_TRY(hipMemcpyAsync(twig.p, w.d_twig + e.twig_off * TWIG_WORDS, twigs * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToDevice, s));
    ChunkPools meta;
    std::memcpy(meta.position, c.position, sizeof meta.position);
    meta.size = c.size; meta.depth = c.depth;
    meta.tree_capacity = c.tree_capacity; meta.twig_capacity = c.twig_capacity;
    DeviceFiller &filler = ctx.filler;
    if (op == SVO_EDIT_DESTROY || op == SVO_EDIT_REPLACE)
        if ((rc = filler.fill(meta, lo, hi, material, tree, trees, twig, twigs, s, EDIT_DESTROY)) != SVO_OK) return rc;
    if (op == SVO_EDIT_BUILD || op == SVO_EDIT_REPLACE)
        if ((rc = filler.fill(meta, lo, hi, material, tree, trees, twig, twigs, s, EDIT_BUILD)) != SVO_OK) return rc;
    meta.trees_on_device = trees; meta.twigs_on_device = twigs;
    return install_resident_chunk(w, chunk, meta, tree.p, twig.p);
}

int edit_box_resident(svo_world &w, int chunk, int op, const float lo[3], const float hi[3], uint32_t material)
{
    try { return edit_box_resident_impl(w, chunk, op, lo, hi, material); }
    catch (const std::bad_alloc &) { set_error("svo_world_edit_box: out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
}

int generate_world_resident(svo_world &w, int device)
{
    try {
        const int rc = generate_world_resident_impl(w, device);
        if (rc < 0) release_device(w);
        return rc;
    }
    catch (const std::bad_alloc &) { release_device(w); set_error("svo_world_generate (device builder): out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
}

} // namespace svo
