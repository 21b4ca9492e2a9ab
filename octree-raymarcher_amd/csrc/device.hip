// device.hip — device half of the C ABI: HBM residency of a world and the trace launches.
//
//   svo_world_upload  <- World::load_gpu + RootAllocator::alloc   src/World.cpp:57-94, src/Allocator.cpp:28-35
//   svo_world_update  <- World::modify + RootAllocator::subst     src/World.cpp:268-274, src/Allocator.cpp:37-55
//   svo_trace*        <- World::draw / draw_shadowmap             src/World.cpp:162-266
//
// The reference's first-fit free-list allocator over GL buffers is not reproduced: a world is
// packed into one flat pool per kind with per-chunk slots sized by the chunk's host capacity
// (the reference also sizes GPU slots by capacity, src/Allocator.cpp:30-33), and a chunk that
// outgrows its slot moves to the pool tail.
//
// No CPU fallback: every entry point fails with SVO_ERR_NO_DEVICE / SVO_ERR_HIP when HIP does.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#ifndef SVO_STACK_REFILL
#define SVO_STACK_REFILL 8       // retired lanes per wave that trigger a refill (cheap: rays are staged in LDS)
#endif
#ifndef SVO_STACK_WAVES
#define SVO_STACK_WAVES 6        // waves per SIMD the stack kernel is register-budgeted for: 80 VGPRs (the asm step holds 63; spills sit in the rare blocks)
#endif

#include "kernel_literal.hip.h"
#include "kernel_stack.hip.h"
#include "wide_tree.hip.h"
#include "world.h"

using namespace svo;

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error(std::string(#expr) + ": " + hipGetErrorString(e_));                          \
            return (e_ == hipErrorOutOfMemory) ? SVO_ERR_OUT_OF_MEMORY                             \
                 : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice) ? SVO_ERR_NO_DEVICE     \
                 : SVO_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

namespace svo {

// bit w of mask[b] = (brick b cell w != 0).  One thread per 16-byte eighth of a brick (8 cells -> one mask byte):
// 16 B per lane, fully coalesced reads of the twig pool, byte stores into the little-endian uint64 masks.  The eight lanes of
// a brick also agree on bmat[b]: the brick's material if all its non-empty cells hold the same one (what grow() produces), 0 for an
// empty brick, 0xFFFF if it holds several (the hit block then reads the cell itself).
__global__ __launch_bounds__(256) void k_brick_masks(const uint16_t *twig, uint64_t *mask, uint16_t *bmat, uint64_t first, uint64_t count)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;      // eighth-of-brick index
    const bool live = i < count * 8;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (live) v = reinterpret_cast<const uint4 *>(twig + first * TWIG_WORDS)[i];
    const uint32_t w[4] = { v.x, v.y, v.z, v.w };
    uint32_t bits = 0, lo = 0xFFFFu, hi = 0u;                       // smallest / largest non-zero material of this eighth
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t a = w[k] & 0xFFFFu, b = w[k] >> 16;
        bits |= (a ? 1u : 0u) << (2 * k);
        bits |= (b ? 1u : 0u) << (2 * k + 1);
        if (a) { lo = a < lo ? a : lo; hi = a > hi ? a : hi; }
        if (b) { lo = b < lo ? b : lo; hi = b > hi ? b : hi; }
    }
#pragma unroll
    for (int d = 1; d < 8; d <<= 1) {                                   // the brick's eight lanes are neighbours (256 % 8 == 0)
        const uint32_t lo2 = __shfl_xor(lo, d, 64), hi2 = __shfl_xor(hi, d, 64);
        lo = lo2 < lo ? lo2 : lo; hi = hi2 > hi ? hi2 : hi;
    }
    if (!live) return;
    reinterpret_cast<uint8_t *>(mask + first)[i] = (uint8_t)bits;
    if ((i & 7u) == 0u) bmat[first + (i >> 3)] = (uint16_t)(hi == 0u ? 0u : (lo == hi ? lo : 0xFFFFu));
}

// svo_tile_order: key = primary + shadow step maxima of the tile (saturating), value = the tile's index
__global__ __launch_bounds__(256) void k_tile_keys(const uint32_t *cost, uint32_t *keys, uint32_t *idx, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t a = cost[2 * i], b = cost[2 * i + 1];
    keys[i] = a + b < a ? 0xFFFFFFFFu : a + b;
    idx[i] = (uint32_t)i;
}

static int launch_masks(svo_world &w, uint64_t first, uint64_t count, hipStream_t s)
{
    if (!count) return SVO_OK;
    const uint64_t blocks = (count * 8 + 255) / 256;
    if (blocks > 0x7FFFFFFFull) { set_error("brick pool too large for one mask launch"); return SVO_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(k_brick_masks, dim3((unsigned)blocks), dim3(256), 0, s, w.d_twig, w.d_mask, w.d_bmat, first, count);
    HIP_TRY(hipGetLastError());
    return SVO_OK;
}

// ---- the large device buffers of a world (tree, brick, mask, material, wide pools, the wide builder's scratch) -------------
// A caller that replaces its world - destroy + generate, a re-pack after an edit outgrew the pools - asks for the sizes it has
// just given back.  hipFree + hipMalloc of multi-GB buffers is not free on every runtime (on the development pool a hipMalloc
// behind a large hipFree stalled for ~4 s about once in ten 12-GB cycles, scripts/alloc_probe.py), so buffers of 1 MiB and more
// go to a small per-process cache instead of back to the driver and are handed out again to requests they fit (best fit, at
// most 25 % + 1 MiB larger than asked for).  At most POOL_CACHE_SLOTS buffers are held; svo_device_cache_trim() returns them.
constexpr size_t POOL_CACHE_SLOTS = 8, POOL_CACHE_MIN = 1u << 20;
struct PoolBuf { void *p; size_t bytes; int device; };
static std::mutex g_pool_mutex;
static std::vector<PoolBuf> g_pool_live, g_pool_cache;      // what pool_malloc has handed out / what pool_free has kept

static hipError_t pool_malloc(void **out, size_t bytes, int device)
{
    if (bytes == 0) bytes = 1;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        size_t best = g_pool_cache.size();
        for (size_t i = 0; i < g_pool_cache.size(); ++i) {
            const PoolBuf &b = g_pool_cache[i];
            if (b.device == device && b.bytes >= bytes && b.bytes <= bytes + bytes / 4 + POOL_CACHE_MIN && (best == g_pool_cache.size() || b.bytes < g_pool_cache[best].bytes)) best = i;
        }
        if (best != g_pool_cache.size()) {
            *out = g_pool_cache[best].p;
            g_pool_live.push_back(g_pool_cache[best]);
            g_pool_cache.erase(g_pool_cache.begin() + (long)best);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess) {                                  // the cache may be what stands in the way: give it back and try once more
        svo_device_cache_trim();
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) { std::lock_guard<std::mutex> lock(g_pool_mutex); g_pool_live.push_back({ *out, bytes, device }); }
    return e;
}
static void pool_free(void *p)
{
    if (!p) return;
    PoolBuf b{ p, 0, -1 };
    void *evict = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_pool_mutex);
        for (size_t i = 0; i < g_pool_live.size(); ++i)
            if (g_pool_live[i].p == p) { b = g_pool_live[i]; g_pool_live.erase(g_pool_live.begin() + (long)i); break; }
        if (b.bytes >= POOL_CACHE_MIN) {
            if (g_pool_cache.size() >= POOL_CACHE_SLOTS) {  // full: the smallest buffer makes room (the large ones are the expensive ones)
                size_t small = 0;
                for (size_t i = 1; i < g_pool_cache.size(); ++i) if (g_pool_cache[i].bytes < g_pool_cache[small].bytes) small = i;
                if (g_pool_cache[small].bytes < b.bytes) { evict = g_pool_cache[small].p; g_pool_cache[small] = b; }
                else evict = p;
            } else g_pool_cache.push_back(b);
        } else evict = p;
    }
    if (evict) (void)hipFree(evict);
}

int release_device(svo_world &w)
{
    if (w.device >= 0) {
        (void)hipSetDevice(w.device);
        free_builder_context(w);
        (void)hipDeviceSynchronize();                       // the large buffers may be handed to another world at once: nothing may still use them
        (void)hipFree(w.d_chunks); pool_free(w.d_tree); pool_free(w.d_twig);
        pool_free(w.d_mask); pool_free(w.d_bmat); (void)hipFree(w.d_work);
        pool_free(w.d_wide); pool_free(w.d_wbase); (void)hipFree(w.d_wchunks); pool_free(w.d_wscratch); (void)hipFree(w.d_sort);
        for (void *e : w.work_event) if (e) (void)hipEventDestroy((hipEvent_t)e);
        if (w.sort_event) (void)hipEventDestroy((hipEvent_t)w.sort_event);
    }
    w.work_event.clear();
    w.d_chunks = nullptr; w.d_tree = nullptr; w.d_twig = nullptr; w.d_mask = nullptr; w.d_bmat = nullptr; w.d_work = nullptr;
    w.d_wide = nullptr; w.d_wbase = nullptr; w.d_wchunks = nullptr; w.d_wscratch = nullptr; w.wscratch_words = 0; w.wscan_words = 0;
    w.d_sort = nullptr; w.sort_bytes = 0; w.sort_event = nullptr;
    if (w.h_wide_tail) { (void)hipHostFree(w.h_wide_tail); w.h_wide_tail = nullptr; }
    w.device = -1;
    w.table.clear(); w.tree_slot.clear(); w.twig_slot.clear(); w.wtable.clear(); w.wide_slot.clear();
    w.tree_pool_len = w.twig_pool_len = w.tree_pool_cap = w.twig_pool_cap = 0;
    w.wide_pool_len = w.wide_pool_cap = 0;
    return SVO_OK;
}

// Persistent grid = the waves the kernel can keep resident (occupancy query), never more than tiles / tiles_per_wave.
template <int MAXLV, bool BIG, bool GLSL>
static int launch_stack_as(svo_world *w, const TraceArgs &A, int tiles_per_wave, int in_flight, hipStream_t s)
{
    auto kernel = k_trace_stack<MAXLV, SVO_STACK_REFILL, SVO_STACK_WAVES, BIG, GLSL>;
    if (w->occupancy_blocks <= 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, w->device) != hipSuccess) return SVO_ERR_HIP;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 64, 0) != hipSuccess || per_cu <= 0) per_cu = 16;
        w->occupancy_blocks = prop.multiProcessorCount * per_cu;
#ifdef SVO_TEST_HOOKS
        if (const char *cap = std::getenv("SVO_GRID_WAVES_PER_CU")) { const int c = std::atoi(cap); if (c > 0) w->occupancy_blocks = prop.multiProcessorCount * std::min(c, per_cu); }     // experiments (scripts/sweep_grid.sh)
#endif
    }
    const int64_t per_wave = tiles_per_wave > 1 ? tiles_per_wave : 1;
    const int64_t tiles = (int64_t)A.ntiles * (A.nframes > 0 ? A.nframes : 1);
    // launches the caller keeps in flight share the wave slots: 2/n each (include/svo.h, svo_trace_params.launches_in_flight)
    const int64_t slots = in_flight >= 2 ? std::max<int64_t>(1, (int64_t)w->occupancy_blocks * 2 / in_flight) : w->occupancy_blocks;
    const int blocks = (int)std::min<int64_t>((tiles + per_wave - 1) / per_wave, slots);
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64), 0, s, A);
    return SVO_OK;
}

template <int MAXLV, bool BIG>
static int launch_stack(svo_world *w, const TraceArgs &A, int tiles_per_wave, int in_flight, hipStream_t s)
{
    return A.glsl ? launch_stack_as<MAXLV, BIG, true>(w, A, tiles_per_wave, in_flight, s) : launch_stack_as<MAXLV, BIG, false>(w, A, tiles_per_wave, in_flight, s);
}


} // namespace svo

extern "C" {

int svo_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void *svo_device_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { set_error("svo_device_alloc: hipMalloc failed"); return nullptr; }
    return p;
}
void svo_device_free(void *p) { if (p) (void)hipFree(p); }
void svo_device_cache_trim(void)
{
    std::vector<PoolBuf> out;
    { std::lock_guard<std::mutex> lock(g_pool_mutex); out.swap(g_pool_cache); }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (const PoolBuf &b : out) { (void)hipSetDevice(b.device); (void)hipFree(b.p); }
    if (have) (void)hipSetDevice(cur);
}
int svo_memcpy_h2d(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); return SVO_OK; }
int svo_memcpy_d2h(void *dst, const void *src, size_t bytes) { HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); return SVO_OK; }
int svo_stream_synchronize(void *stream) { HIP_TRY(hipStreamSynchronize((hipStream_t)stream)); return SVO_OK; }

} // extern "C"

namespace svo {

int launch_brick_masks(svo_world &w, uint64_t first, uint64_t count, void *stream) { return launch_masks(w, first, count, (hipStream_t)stream); }

// slots: capacity-sized like the reference (src/Allocator.cpp:30-33), 8-node / 1-brick granular, plus tail slack so that a
// chunk that outgrows its slot can be re-packed without a full re-upload
int plan_pools(svo_world &w)
{
    for (const ChunkPools &c : w.chunks)
        if (c.size != (float)w.chunksize) { set_error("svo_world_upload: every chunk's size must equal chunksize"); return SVO_ERR_UNSUPPORTED; }
    const size_t n = w.chunks.size();
    w.table.assign(n, DevChunk());
    w.wtable.assign(n, DevWide());
    w.tree_slot.assign(n, 0); w.twig_slot.assign(n, 0); w.wide_slot.assign(n, 0);
    uint64_t tcur = 0, bcur = 0;
    for (size_t i = 0; i < n; ++i) {
        const ChunkPools &c = w.chunks[i];
        const uint64_t tcap = std::max<uint64_t>(c.tree_capacity, c.tree_count());
        const uint64_t bcap = std::max<uint64_t>(c.twig_capacity, c.twig_count());
        const uint64_t base = ((tcur + 8) & ~(uint64_t)7) - 1;          // base % 8 == 7, base >= tcur
        DevChunk &e = w.table[i];
        e.bmin[0] = c.position[0]; e.bmin[1] = c.position[1]; e.bmin[2] = c.position[2];
        e.levels = c.depth - TWIG_LEVELS;
        e.tree_off = base;
        e.twig_off = bcur;
        w.tree_slot[i] = tcap; w.twig_slot[i] = bcap;
        tcur = base + tcap;
        bcur += bcap;
    }
    w.tree_pool_len = tcur; w.twig_pool_len = bcur;
    w.tree_pool_cap = tcur + tcur / 4 + 64;
    w.twig_pool_cap = bcur + bcur / 4 + 16;
    return SVO_OK;
}

int alloc_pools(svo_world &w, int device)
{
    if (hipSetDevice(device) != hipSuccess) { set_error("svo_world_upload: hipSetDevice failed"); return SVO_ERR_NO_DEVICE; }
    w.device = device;
    const size_t n = w.chunks.size();
    if (pool_malloc((void **)&w.d_tree, w.tree_pool_cap * sizeof(uint32_t), device) != hipSuccess ||
        pool_malloc((void **)&w.d_twig, w.twig_pool_cap * TWIG_WORDS * sizeof(uint16_t), device) != hipSuccess ||
        pool_malloc((void **)&w.d_mask, w.twig_pool_cap * sizeof(uint64_t), device) != hipSuccess ||
        pool_malloc((void **)&w.d_bmat, w.twig_pool_cap * sizeof(uint16_t), device) != hipSuccess ||
        hipMalloc((void **)&w.d_chunks, n * sizeof(DevChunk)) != hipSuccess ||
        hipMalloc((void **)&w.d_wchunks, n * sizeof(DevWide)) != hipSuccess ||
        hipMalloc((void **)&w.d_work, WORK_SLOTS * WORK_SLOT_WORDS * sizeof(unsigned long long)) != hipSuccess) {
        set_error("svo_world_upload: hipMalloc failed"); return SVO_ERR_OUT_OF_MEMORY;
    }
    if (hipMemset(w.d_tree, 0, w.tree_pool_cap * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(w.d_mask, 0, w.twig_pool_cap * sizeof(uint64_t)) != hipSuccess ||
        hipMemset(w.d_bmat, 0, w.twig_pool_cap * sizeof(uint16_t)) != hipSuccess ||
        hipMemset(w.d_work, 0, WORK_SLOTS * WORK_SLOT_WORDS * sizeof(unsigned long long)) != hipSuccess) { set_error("svo_world_upload: hipMemset failed"); return SVO_ERR_HIP; }
    w.occupancy_blocks = 0;
    w.wide_ok = false;                                                  // until build_wide_all has run
    return SVO_OK;
}

// A chunk built on the device keeps its node words and bricks there until somebody asks for the host copy.
int fetch_pools(svo_world &w, int chunk)
{
    ChunkPools &c = w.chunks[(size_t)chunk];
    if (!c.twigs_on_device && !c.trees_on_device) return SVO_OK;
    if (w.device < 0 || !w.d_twig || !w.d_tree) { set_error("fetch_pools: the device copy is gone"); return SVO_ERR_NOT_UPLOADED; }
    HIP_TRY(hipSetDevice(w.device));
    if (c.trees_on_device) {
        const uint64_t n = c.trees_on_device;
        c.tree.resize(n);
        HIP_TRY(hipMemcpy(c.tree.data(), w.d_tree + w.table[(size_t)chunk].tree_off, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
        c.trees_on_device = 0;
    }
    if (c.twigs_on_device) {
        const uint64_t n = c.twigs_on_device;
        c.twig.resize(n * TWIG_WORDS);
        HIP_TRY(hipMemcpy(c.twig.data(), w.d_twig + w.table[(size_t)chunk].twig_off * TWIG_WORDS, n * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToHost));
        c.twigs_on_device = 0;
    }
    return SVO_OK;
}

// The wide tree of chunk `chunk` (wide_tree.hip.h) from its node words in the tree pool, level by level, into
// wide_dst / wbase_dst (room for slot_cap wide nodes); *count = wide nodes written.  The per-entry reference indices that link
// one level to the next live in the builder's scratch.
constexpr int WIDE_SLOT_FULL = 100;     // expand_wide_chunk: the chunk's wide tree does not fit slot_cap (not an svo_status: never leaves this file)
static int expand_wide_chunk(svo_world &w, int chunk, hipStream_t s, uint32_t *wide_dst, uint32_t *wbase_dst, uint64_t slot_cap, uint64_t *count_out)
{
    const ChunkPools &c = w.chunks[(size_t)chunk];
    const DevChunk &e = w.table[(size_t)chunk];
    const uint32_t levels = c.depth - TWIG_LEVELS;
    const uint32_t nw = levels == 0 ? 1u : (levels + 1u) / 2u;
    const int pad = (int)(2u * nw - levels);
    const uint64_t B = c.tree_count() / 8 + 1;                          // most BRANCH nodes (= wide nodes) a level can have
    uint32_t *front = w.d_wscratch, *next = front + B, *flag = next + B, *rank = flag + 64 * B, *wref_dst = rank + 64 * B;
    const uint32_t *tree = w.d_tree + e.tree_off;
    HIP_TRY(hipMemsetAsync(front, 0, sizeof(uint32_t), s));             // the top wide node expands reference node 0
    if (!w.h_wide_tail && hipHostMalloc((void **)&w.h_wide_tail, 2 * sizeof(uint32_t)) != hipSuccess) { set_error("wide tree: hipHostMalloc failed"); return SVO_ERR_OUT_OF_MEMORY; }
    uint32_t count = 1, first = 0;
    // the scan's own scratch lies behind the builder's (reserve_wide_scratch sized it for the largest level): no hipMalloc / hipFree per chunk
    void *tmp = w.d_wscratch + (w.wscratch_words - w.wscan_words);
    const size_t tmp_bytes = w.wscan_words * sizeof(uint32_t);
    int rc = SVO_OK;
    for (uint32_t k = 0; k < nw && count > 0; ++k) {
        if ((uint64_t)first + count > slot_cap) { rc = WIDE_SLOT_FULL; break; }      // (the callers turn this into a larger slot, or into an error)
        const uint32_t n = count * 64u;
        hipLaunchKernelGGL(k_wide_expand, dim3((n + 255) / 256), dim3(256), 0, s, tree, front, count, first,
                           k == 0 ? pad : 0, 2u * k + 1u - (uint32_t)pad, wide_dst, wref_dst, wbase_dst, flag);
        if (hipGetLastError() != hipSuccess) { set_error("wide tree: launch failed"); rc = SVO_ERR_HIP; break; }
        if (k + 1 == nw) { first += count; count = 0; break; }         // grandchildren of the last wide level are never BRANCH
        size_t bytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, flag, rank, (int)n, s) != hipSuccess) { rc = SVO_ERR_HIP; break; }
        if (bytes > tmp_bytes) { set_error("wide tree: the scan asks for more scratch than was reserved"); rc = SVO_ERR_HIP; break; }
        uint32_t *tail = w.h_wide_tail;                                 // pinned: a pageable destination stages every 4-byte copy
        if (hipcub::DeviceScan::ExclusiveSum(tmp, bytes, flag, rank, (int)n, s) != hipSuccess ||
            hipMemcpyAsync(&tail[0], rank + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipMemcpyAsync(&tail[1], flag + (n - 1), 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { set_error("wide tree: scan failed"); rc = SVO_ERR_HIP; break; }
        const uint32_t total = tail[0] + tail[1];
        if ((uint64_t)total > B) { set_error("wide tree: more BRANCH nodes than the tree can hold"); rc = SVO_ERR_MALFORMED_TREE; break; }
        if (total) {
            hipLaunchKernelGGL(k_wide_link, dim3((n + 255) / 256), dim3(256), 0, s, count, first, first + count, flag, rank, wide_dst, wref_dst, next);
            if (hipGetLastError() != hipSuccess) { rc = SVO_ERR_HIP; break; }
        }
        first += count; count = total;
        std::swap(front, next);
    }
    (void)hipStreamSynchronize(s);
    if (rc == SVO_OK && count_out) *count_out = first;
    return rc;
}

// scratch for the builder: fronts, flags, ranks of the largest chunk, and (count pass) a throw-away wide tree of its bound
static int reserve_wide_scratch(svo_world &w, uint64_t largest_tree)
{
    const uint64_t B = largest_tree / 8 + 1;
    size_t scan_bytes = 0;
    uint32_t *nul = nullptr;
    if (hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, nul, nul, (int)(64 * B), (hipStream_t)nullptr) != hipSuccess) { set_error("wide tree: scan size query failed"); return SVO_ERR_HIP; }
    const uint64_t scan_words = (scan_bytes + 3) / 4 + 64;
    // fronts, flags + ranks, entry references + a throw-away tree, its bases; the scan's scratch (kept 256-byte aligned: everything before it is a multiple of 64 words)
    const uint64_t body = ((2 * B + 2 * 64 * B + 2 * 64 * B + WIDE_BASE_WORDS * B + 1024 + 63) / 64) * 64;
    const uint64_t need = body + scan_words;
    if (scan_words <= w.wscan_words && body <= w.wscratch_words - w.wscan_words) return SVO_OK;
    if (w.d_wscratch) { (void)hipDeviceSynchronize(); pool_free(w.d_wscratch); w.d_wscratch = nullptr; w.wscratch_words = 0; w.wscan_words = 0; }
    if (pool_malloc((void **)&w.d_wscratch, need * sizeof(uint32_t), w.device) != hipSuccess) { set_error("wide tree: hipMalloc of the builder scratch failed"); return SVO_ERR_OUT_OF_MEMORY; }
    w.wscratch_words = need; w.wscan_words = scan_words;
    return SVO_OK;
}
static bool wide_fits(const svo_world &w, int chunk)
{
    const ChunkPools &c = w.chunks[(size_t)chunk];
    // (a wide entry keeps its reference node's level in 5 bits; WIDE_MAX_LEVELS branch levels, i.e. chunk depth <= 24, are marched)
    // and the builder scans one level's entries (64 per wide node) with 32-bit counts: < 2^31 entries per chunk
    return c.depth - TWIG_LEVELS <= WIDE_MAX_LEVELS && c.twig_count() <= (uint64_t)WIDE_PAYLOAD_MASK && (c.tree_count() / 8 + 1) * 64 < (1ull << 31);
}

// Wide trees of every chunk: a count pass into scratch sizes the pool (each chunk's slot = its wide nodes + 25 % + 16),
// the build pass writes them in place.  The node words must already be in the tree pool.
static void drop_wide(svo_world &w)
{
    if (w.d_wide || w.d_wbase) (void)hipDeviceSynchronize();
    pool_free(w.d_wide); pool_free(w.d_wbase); w.d_wide = w.d_wbase = nullptr;
    w.wide_ok = false;
    w.wide_pool_len = w.wide_pool_cap = w.wide_nodes_used = 0;
}
static void drop_wide_scratch(svo_world &w, bool failed = true)
{
    // after a successful rebuild an interactive caller (one that has edited or slid the world: builder_ctx) keeps the scratch for
    // the next one; everybody else gets the ~1 GB back
    if (!failed && w.builder_ctx) return;
    if (w.d_wscratch) { (void)hipDeviceSynchronize(); pool_free(w.d_wscratch); w.d_wscratch = nullptr; }
    w.wscratch_words = 0; w.wscan_words = 0;
}
// test hook (the `hooks` variant of the Makefile only; the shipped library reads no such variable): SVO_TEST_FAIL_WIDE=1 makes
// the next wide-tree build fail as an allocation failure would
static bool wide_fault_injected()
{
#ifdef SVO_TEST_HOOKS
    const char *e = std::getenv("SVO_TEST_FAIL_WIDE");
    return e && e[0] == '1';
#else
    return false;
#endif
}

int build_wide_all(svo_world &w, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    const size_t n = w.chunks.size();
    // wide_ok says "the wide pool is complete": false from here until the build pass has succeeded, so that a failure
    // on the way (scratch or pool allocation, a malformed tree) leaves a world the literal kernel marches, never a
    // stack kernel reading a null or half-written pool
    drop_wide(w);
    bool fits = true;
    uint64_t largest = 0;
    for (size_t i = 0; i < n; ++i) { largest = std::max<uint64_t>(largest, w.chunks[i].tree_count()); if (!wide_fits(w, (int)i)) fits = false; }
    w.wtable.assign(n, DevWide()); w.wide_slot.assign(n, 0);
    if (!fits) return SVO_OK;                                           // the literal kernel marches such a world
    int rc = reserve_wide_scratch(w, largest);
    if (rc == SVO_OK && wide_fault_injected()) { set_error("wide tree: injected allocation failure"); rc = SVO_ERR_OUT_OF_MEMORY; }
    if (rc != SVO_OK) { drop_wide_scratch(w); return rc; }
    const uint64_t Bmax = largest / 8 + 1;
    uint32_t *tmp_wide = w.d_wscratch + 2 * Bmax + 3 * 64 * Bmax, *tmp_wbase = tmp_wide + 64 * Bmax;
    // One pass (until round 4 every chunk was expanded twice: a count pass into scratch sized the pool, a build pass filled it): the
    // largest chunk is counted, the pool is sized from its wide nodes per BRANCH node (+ 35 %) for all chunks, and every chunk is built in
    // place behind the previous one's slot (= its wide nodes + 1/8 + 16); a pool that turns out too small is grown (x 1.5, copied):
    // entries hold wide-node indices relative to their chunk's top node, so a built chunk can move.
    uint64_t count0 = 0;
    size_t sample = 0;                                                          // the chunk with the most nodes stands for all of them
    for (size_t i = 0; i < n; ++i) if (w.chunks[i].tree_count() > w.chunks[sample].tree_count()) sample = i;
    if ((rc = expand_wide_chunk(w, (int)sample, s, tmp_wide, tmp_wbase, Bmax, &count0)) != SVO_OK) {
        if (rc == WIDE_SLOT_FULL) { set_error("wide tree: more wide nodes than BRANCH nodes"); rc = SVO_ERR_MALFORMED_TREE; }
        drop_wide_scratch(w); return rc;
    }
    uint64_t branches = 0;
    for (size_t i = 0; i < n; ++i) branches += w.chunks[i].tree_count() / 8 + 1;
    const double per_branch = (double)count0 / (double)(w.chunks[sample].tree_count() / 8 + 1);
    uint64_t cap = (uint64_t)((double)branches * per_branch * 1.35) + 32 * n + 64;
    cap = std::max<uint64_t>(cap, count0 + count0 / 8 + 16 + 64);
#ifdef SVO_TEST_HOOKS
    // (the `hooks` variant only) SVO_TEST_WIDE_ESTIMATE=<factor> scales the estimate, so that the tests reach the growth path
    if (const char *e = std::getenv("SVO_TEST_WIDE_ESTIMATE")) cap = std::max<uint64_t>(64, (uint64_t)((double)cap * std::atof(e)));
#endif
    auto alloc_pool = [&](uint64_t nodes, uint32_t **wide, uint32_t **wbase) {
        *wide = *wbase = nullptr;
        if (nodes >= (1ull << 32)) return false;
        if (pool_malloc((void **)wide, nodes * 64 * sizeof(uint32_t), w.device) != hipSuccess) return false;
        if (pool_malloc((void **)wbase, nodes * WIDE_BASE_WORDS * sizeof(uint32_t), w.device) != hipSuccess) { pool_free(*wide); *wide = nullptr; return false; }
        return true;
    };
    if (cap >= (1ull << 32)) { drop_wide_scratch(w); return SVO_OK; }         // wide node indices are 32-bit (1 TiB of wide nodes): literal kernel
    if (!alloc_pool(cap, &w.d_wide, &w.d_wbase)) {
        drop_wide(w); drop_wide_scratch(w);
        set_error("wide tree: hipMalloc of the pool failed"); return SVO_ERR_OUT_OF_MEMORY;
    }
    uint64_t cur = 0, used = 0;
    auto grow_pool = [&](uint64_t at_least) -> int {
        uint64_t bigger = std::max<uint64_t>(cap + cap / 2, at_least + at_least / 16 + 64);
        if (bigger >= (1ull << 32)) return 1;                                   // literal kernel
        uint32_t *nw = nullptr, *nb = nullptr;
        if (!alloc_pool(bigger, &nw, &nb)) return SVO_ERR_OUT_OF_MEMORY;
        if (hipMemcpyAsync(nw, w.d_wide, cur * 64 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(nb, w.d_wbase, cur * WIDE_BASE_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) { pool_free(nw); pool_free(nb); return SVO_ERR_HIP; }
        pool_free(w.d_wide); pool_free(w.d_wbase);
        w.d_wide = nw; w.d_wbase = nb; cap = bigger;
        return SVO_OK;
    };
    for (size_t i = 0; i < n; ) {
        uint64_t count = 0;
        rc = expand_wide_chunk(w, (int)i, s, w.d_wide + cur * 64, w.d_wbase + cur * WIDE_BASE_WORDS, cap - cur, &count);
        const uint64_t slot = count + count / 8 + 16;
        if (rc == WIDE_SLOT_FULL || (rc == SVO_OK && cur + slot > cap)) {        // (grow, then this chunk again)
            const uint64_t bound = w.chunks[i].tree_count() / 8 + 1;
            const int g = grow_pool(cur + (rc == SVO_OK ? slot : std::min<uint64_t>(bound, 2 * (cap - cur) + 1024)));
            if (g == 1) { drop_wide(w); drop_wide_scratch(w); return SVO_OK; }
            if (g != SVO_OK) { drop_wide(w); drop_wide_scratch(w); set_error("wide tree: growing the pool failed"); return g; }
            continue;
        }
        if (rc != SVO_OK) { drop_wide(w); drop_wide_scratch(w); return rc; }
        const DevChunk &e = w.table[i];
        DevWide &v = w.wtable[i];
        v.bmin[0] = e.bmin[0]; v.bmin[1] = e.bmin[1]; v.bmin[2] = e.bmin[2];
        v.levels = e.levels; v.wide_off = (uint32_t)cur; v._pad = 0; v.twig_off = e.twig_off;
        w.wide_slot[i] = slot;
        cur += slot;
        used += count;
        ++i;
    }
    if (hipMemcpy(w.d_wchunks, w.wtable.data(), n * sizeof(DevWide), hipMemcpyHostToDevice) != hipSuccess) {
        drop_wide(w); drop_wide_scratch(w); set_error("wide tree: chunk table copy failed"); return SVO_ERR_HIP;
    }
    w.wide_pool_len = cur; w.wide_pool_cap = cap; w.wide_nodes_used = used;
    w.wide_ok = true;
    // the builder's scratch (fronts, flags, ranks and a throw-away tree of the largest chunk: ~1 GB at C3) is only needed
    // here and by svo_world_update, which re-reserves what the edited chunk needs
    (void)hipStreamSynchronize(s);
    drop_wide_scratch(w, false);
    return SVO_OK;
}

// One chunk again after an edit: in place if its wide tree still fits the slot, at the pool's tail if that has room,
// otherwise everything is rebuilt.
int rebuild_wide_chunk(svo_world &w, int chunk, void *stream)
{
    hipStream_t s = (hipStream_t)stream;
    if (!w.wide_ok || !wide_fits(w, chunk)) return build_wide_all(w, stream);
    const ChunkPools &c = w.chunks[(size_t)chunk];
    // any failure below leaves the chunk's old wide tree in the pool while tree[] has changed: the pool is dropped
    // (wide_ok = false) and the literal kernel takes over until a full rebuild succeeds
    int rc = reserve_wide_scratch(w, c.tree_count());
    if (rc == SVO_OK && wide_fault_injected()) { set_error("wide tree: injected allocation failure"); rc = SVO_ERR_OUT_OF_MEMORY; }
    if (rc != SVO_OK) { drop_wide(w); drop_wide_scratch(w); return rc; }
    const uint64_t Bmax = c.tree_count() / 8 + 1;
    uint32_t *tmp_wide = w.d_wscratch + 2 * Bmax + 3 * 64 * Bmax, *tmp_wbase = tmp_wide + 64 * Bmax;
    uint64_t count = 0;
    if ((rc = expand_wide_chunk(w, chunk, s, tmp_wide, tmp_wbase, Bmax, &count)) != SVO_OK) {
        if (rc == WIDE_SLOT_FULL) { set_error("wide tree: more wide nodes than BRANCH nodes"); rc = SVO_ERR_MALFORMED_TREE; }
        drop_wide(w); drop_wide_scratch(w); return rc;
    }
    DevWide &v = w.wtable[(size_t)chunk];
    const DevChunk &e = w.table[(size_t)chunk];
    if (count > w.wide_slot[(size_t)chunk]) {
        const uint64_t want = count + count / 4 + 16;
        if (w.wide_pool_len + want > w.wide_pool_cap) { drop_wide_scratch(w); return build_wide_all(w, stream); }
        v.wide_off = (uint32_t)w.wide_pool_len;
        w.wide_slot[(size_t)chunk] = want;
        w.wide_pool_len += want;
    }
    v.bmin[0] = e.bmin[0]; v.bmin[1] = e.bmin[1]; v.bmin[2] = e.bmin[2];
    v.levels = e.levels; v.twig_off = e.twig_off;
    if (hipMemcpyAsync(w.d_wide + (uint64_t)v.wide_off * 64, tmp_wide, count * 64 * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(w.d_wbase + (uint64_t)v.wide_off * WIDE_BASE_WORDS, tmp_wbase, count * WIDE_BASE_WORDS * sizeof(uint32_t), hipMemcpyDeviceToDevice, s) != hipSuccess ||
        hipMemcpyAsync(w.d_wchunks + chunk, &v, sizeof(DevWide), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        drop_wide(w); drop_wide_scratch(w); set_error("wide tree: copy of the rebuilt chunk failed"); return SVO_ERR_HIP;
    }
    drop_wide_scratch(w, false);
    return SVO_OK;
}

} // namespace svo

extern "C" {

static int world_upload_impl(svo_world *w, int device, bool force = false)
{
    if (!w) return SVO_ERR_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("svo_world_upload: no HIP device"); return SVO_ERR_NO_DEVICE; }
    if (device < 0 || device >= ndev) { set_error("svo_world_upload: device index out of range"); return SVO_ERR_INVALID_ARG; }
    // a world generated on this device is already resident there, pools packed exactly as below
    bool resident_only = false;
    for (const ChunkPools &c : w->chunks) resident_only |= c.twigs_on_device != 0 || c.trees_on_device != 0;
    if (resident_only && w->device == device && !force) return SVO_OK;
    for (size_t i = 0; i < w->chunks.size(); ++i) {                     // moving elsewhere: the host copy must be complete first
        const int rc = fetch_pools(*w, (int)i);
        if (rc != SVO_OK) return rc;
    }
    int rc = plan_pools(*w);
    if (rc != SVO_OK) return rc;
    const std::vector<DevChunk> table = w->table;
    const std::vector<uint64_t> tslot = w->tree_slot, bslot = w->twig_slot;
    const uint64_t tl = w->tree_pool_len, bl = w->twig_pool_len, tc = w->tree_pool_cap, bc = w->twig_pool_cap;
    void *ctx = w->device == device ? w->builder_ctx : nullptr;          // a re-pack on the same device keeps the builders' buffers: the caller may be one of them
    if (ctx) w->builder_ctx = nullptr;
    release_device(*w);                                                 // (clears the plan too)
    w->builder_ctx = ctx;
    w->table = table; w->tree_slot = tslot; w->twig_slot = bslot;
    w->tree_pool_len = tl; w->twig_pool_len = bl; w->tree_pool_cap = tc; w->twig_pool_cap = bc;

    const size_t n = w->chunks.size();
    bool literal_only = false;
    do {
        if ((rc = alloc_pools(*w, device)) != SVO_OK) break;
        for (size_t i = 0; i < n && rc == SVO_OK; ++i) {
            const ChunkPools &c = w->chunks[i];
            const DevChunk &e = w->table[i];
            if (hipMemcpy(w->d_tree + e.tree_off, c.tree.data(), c.tree.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) rc = SVO_ERR_HIP;
            if (rc == SVO_OK && !c.twig.empty() &&
                hipMemcpy(w->d_twig + e.twig_off * TWIG_WORDS, c.twig.data(), c.twig.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) rc = SVO_ERR_HIP;
            if (rc == SVO_OK) rc = launch_masks(*w, e.twig_off, c.twig_count(), nullptr);
        }
        if (rc != SVO_OK) { if (rc == SVO_ERR_HIP) set_error("svo_world_upload: copy failed"); break; }
        // the stack kernel's wide trees: a failure here (device memory, mostly) leaves a complete world for the literal kernel
        // (build_wide_all has dropped whatever it had begun), and the caller is told so
        literal_only = build_wide_all(*w, nullptr) != SVO_OK;
        if (hipMemcpy(w->d_chunks, w->table.data(), n * sizeof(DevChunk), hipMemcpyHostToDevice) != hipSuccess ||
            hipDeviceSynchronize() != hipSuccess) { set_error("svo_world_upload: chunk table copy failed"); rc = SVO_ERR_HIP; break; }
    } while (0);
    if (rc != SVO_OK) { release_device(*w); return rc; }
    return literal_only ? SVO_OK_LITERAL_ONLY : SVO_OK;
}

static int world_update_impl(svo_world *w, int chunk, const svo_chunk_desc *desc,
                             uint64_t tree_left, uint64_t tree_right, uint64_t twig_left, uint64_t twig_right, int realloc_)
{
    if (!w || !desc || chunk < 0 || chunk >= (int)w->chunks.size() || !desc->tree || !desc->trees) return SVO_ERR_INVALID_ARG;
    // 1. adopt the edited pools on the host (validated like svo_world_create).
    //    The reference re-sends only the ranges Ocroot::build / destroy report as dirty unless the pools were reallocated
    //    (src/World.cpp:268-274, glBufferSubData); the host copy kept here follows suit when it can - it is current, the chunk's
    //    frame is unchanged, the pools have not shrunk and the caller does not ask for `realloc` - and is patched over the dirty ranges
    //    (and whatever was appended) instead of being replaced: copying a depth-12 chunk's 280 MB took 46 of an update's 64 ms.
    //    Either way the result is validated as a whole before HBM sees any of it, and a patch that fails is taken back.
    ChunkPools &cur = w->chunks[(size_t)chunk];
    const uint64_t old_trees = cur.tree.size(), old_twigs = cur.twig_count();
    const bool patch = !realloc_ && cur.trees_on_device == 0 && cur.twigs_on_device == 0 && old_trees != 0 &&
        desc->trees >= old_trees && desc->twigs >= old_twigs && (desc->twigs == 0 || desc->twig) &&
        std::memcmp(cur.position, desc->position, sizeof cur.position) == 0 && cur.size == desc->size && cur.depth == desc->depth;
    std::string why;
    int rc;
    if (patch) {
        // the ranges that change on the host: the caller's, clamped, widened over what was appended
        uint64_t tl = std::min<uint64_t>(tree_left, desc->trees), tr = std::min<uint64_t>(std::max(tree_right, tree_left), desc->trees);
        uint64_t bl = std::min<uint64_t>(twig_left, desc->twigs), br = std::min<uint64_t>(std::max(twig_right, twig_left), desc->twigs);
        if (desc->trees > old_trees) { tl = std::min(tl, old_trees); tr = desc->trees; }
        if (desc->twigs > old_twigs) { bl = std::min(bl, old_twigs); br = desc->twigs; }
        const uint64_t keep_t = std::min(tr, old_trees), keep_b = std::min(br, old_twigs);          // what a failed patch has to restore
        std::vector<uint32_t> undo_t(cur.tree.begin() + (ptrdiff_t)std::min(tl, keep_t), cur.tree.begin() + (ptrdiff_t)keep_t);
        std::vector<uint16_t> undo_b(cur.twig.begin() + (ptrdiff_t)(std::min(bl, keep_b) * TWIG_WORDS), cur.twig.begin() + (ptrdiff_t)(keep_b * TWIG_WORDS));
        cur.tree.resize(desc->trees);
        cur.twig.resize(desc->twigs * TWIG_WORDS);
        if (tl < tr) std::memcpy(cur.tree.data() + tl, desc->tree + tl, (tr - tl) * sizeof(uint32_t));
        if (bl < br) std::memcpy(cur.twig.data() + bl * TWIG_WORDS, desc->twig + bl * TWIG_WORDS, (br - bl) * TWIG_WORDS * sizeof(uint16_t));
        const uint64_t cap_t0 = cur.tree_capacity, cap_b0 = cur.twig_capacity;
        while (cur.tree_capacity <= cur.tree.size() + 8) cur.tree_capacity *= 2;
        while (cur.twig_capacity < cur.twig_count()) cur.twig_capacity *= 2;
        rc = validate_chunk(cur, why);
        if (rc != SVO_OK) {
            cur.tree.resize(old_trees); cur.twig.resize(old_twigs * TWIG_WORDS);
            std::copy(undo_t.begin(), undo_t.end(), cur.tree.begin() + (ptrdiff_t)std::min(tl, keep_t));
            std::copy(undo_b.begin(), undo_b.end(), cur.twig.begin() + (ptrdiff_t)(std::min(bl, keep_b) * TWIG_WORDS));
            cur.tree_capacity = cap_t0; cur.twig_capacity = cap_b0;
            set_error("svo_world_update: " + why); return rc;
        }
        tree_left = tl; tree_right = tr; twig_left = bl; twig_right = br;       // what HBM receives below
    } else {
        ChunkPools next;
        std::memcpy(next.position, desc->position, sizeof next.position);
        next.size = desc->size; next.depth = desc->depth;
        next.tree.assign(desc->tree, desc->tree + desc->trees);
        if (desc->twigs) next.twig.assign(desc->twig, desc->twig + desc->twigs * TWIG_WORDS);
        next.tree_capacity = cur.tree_capacity;
        next.twig_capacity = cur.twig_capacity;
        while (next.tree_capacity <= next.tree.size() + 8) next.tree_capacity *= 2;
        while (next.twig_capacity < next.twig_count()) next.twig_capacity *= 2;
        rc = validate_chunk(next, why);
        if (rc != SVO_OK) { set_error("svo_world_update: " + why); return rc; }
        if (next.size != (float)w->chunksize) { set_error("svo_world_update: chunk size must equal chunksize"); return SVO_ERR_UNSUPPORTED; }
        cur.tree.swap(next.tree);
        cur.twig.swap(next.twig);
        cur.twigs_on_device = 0;       // the caller's pools replace whatever lived only on the device
        cur.trees_on_device = 0;
        std::memcpy(cur.position, next.position, sizeof cur.position);
        cur.size = next.size; cur.depth = next.depth;
        cur.tree_capacity = next.tree_capacity; cur.twig_capacity = next.twig_capacity;
        if (!realloc_ && (desc->trees != old_trees || desc->twigs != old_twigs)) { tree_left = 0; tree_right = desc->trees; twig_left = 0; twig_right = desc->twigs; }
    }
    ChunkPools &c = cur;
    classify_world(*w);
    if (w->device < 0) return SVO_OK;

    // 2. refresh HBM.  Launches of this world may still be in flight on the caller's streams (non-blocking streams are
    //    not ordered against the copies below, and a march that reads a half-rewritten tree could follow a stale BRANCH
    //    chain): like World::modify on the GL queue (src/World.cpp:268-274), the update is ordered behind everything
    //    issued before it - the device is drained first.
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipDeviceSynchronize());
    DevChunk &e = w->table[(size_t)chunk];
    const bool tree_fits = c.tree.size() <= w->tree_slot[(size_t)chunk];
    const bool twig_fits = c.twig_count() <= w->twig_slot[(size_t)chunk];
    bool table_dirty = e.levels != c.depth - TWIG_LEVELS || e.bmin[0] != c.position[0] || e.bmin[1] != c.position[1] || e.bmin[2] != c.position[2];
    e.levels = c.depth - TWIG_LEVELS;
    e.bmin[0] = c.position[0]; e.bmin[1] = c.position[1]; e.bmin[2] = c.position[2];
    if (!tree_fits || !twig_fits) {
        // move the outgrown pool(s) to the tail; no room there -> full re-upload
        const uint64_t tbase = ((w->tree_pool_len + 8) & ~(uint64_t)7) - 1;
        const uint64_t need_t = tree_fits ? 0 : c.tree_capacity, need_b = twig_fits ? 0 : c.twig_capacity;
        if ((!tree_fits && tbase + need_t > w->tree_pool_cap) || (!twig_fits && w->twig_pool_len + need_b > w->twig_pool_cap))
            return world_upload_impl(w, w->device, true);
        if (!tree_fits) { e.tree_off = tbase; w->tree_slot[(size_t)chunk] = need_t; w->tree_pool_len = tbase + need_t; }
        if (!twig_fits) { e.twig_off = w->twig_pool_len; w->twig_slot[(size_t)chunk] = need_b; w->twig_pool_len += need_b; }
        table_dirty = true;
        realloc_ = 1;
        if (tree_fits) { /* tree stays, only its dirty range is re-sent below */ }
    }
    uint64_t tl = tree_left, tr = tree_right, bl = twig_left, br = twig_right;
    if (realloc_ || !tree_fits) { tl = 0; tr = c.tree.size(); }
    if (realloc_ || !twig_fits) { bl = 0; br = c.twig_count(); }
    tr = std::min<uint64_t>(tr, c.tree.size()); br = std::min<uint64_t>(br, c.twig_count());
    if (tl < tr) HIP_TRY(hipMemcpy(w->d_tree + e.tree_off + tl, c.tree.data() + tl, (tr - tl) * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (bl < br) {
        HIP_TRY(hipMemcpy(w->d_twig + (e.twig_off + bl) * TWIG_WORDS, c.twig.data() + bl * TWIG_WORDS, (br - bl) * TWIG_WORDS * sizeof(uint16_t), hipMemcpyHostToDevice));
        rc = launch_masks(*w, e.twig_off + bl, br - bl, nullptr);
        if (rc != SVO_OK) return rc;
    }
    if (table_dirty) HIP_TRY(hipMemcpy(w->d_chunks + chunk, &e, sizeof(DevChunk), hipMemcpyHostToDevice));
    // the stack kernel's view of the chunk: rebuilt from the node words now in the pool
    // (the pools and the chunk table already hold the new chunk: a wide tree that cannot be rebuilt - rebuild_wide_chunk has dropped
    // the wide pool then - leaves a world the literal kernel marches, and the caller is told so instead of being told "error"
    // about a change that took effect)
    rc = rebuild_wide_chunk(*w, chunk, nullptr);
    HIP_TRY(hipDeviceSynchronize());
    return rc == SVO_OK ? SVO_OK : SVO_OK_LITERAL_ONLY;
}

// nothing throws across the C ABI: host-side allocations of the two entry points above are fenced here
int svo_world_upload(svo_world *w, int device)
{
    try { return world_upload_impl(w, device); }
    catch (const std::bad_alloc &) { set_error("svo_world_upload: out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
    catch (...) { set_error("svo_world_upload: unexpected exception"); return SVO_ERR_HIP; }
}

int svo_world_update(svo_world *w, int chunk, const svo_chunk_desc *desc,
                     uint64_t tree_left, uint64_t tree_right, uint64_t twig_left, uint64_t twig_right, int realloc_)
{
    try { return world_update_impl(w, chunk, desc, tree_left, tree_right, twig_left, twig_right, realloc_); }
    catch (const std::bad_alloc &) { set_error("svo_world_update: out of host memory"); return SVO_ERR_OUT_OF_MEMORY; }
    catch (...) { set_error("svo_world_update: unexpected exception"); return SVO_ERR_HIP; }
}

} // extern "C"

namespace svo {

// svo_world_update's device half for a chunk whose new pools already lie in HBM (World::shift on an uploaded world,
// builder.hip): same ordering rule (the device is drained first), same slot logic - in place if it fits, at the pools' tails
// if they have room; otherwise the chunk comes to the host and the whole world is packed again.
int install_resident_chunk(svo_world &w, int chunk, const ChunkPools &meta, const uint32_t *tree_dev, const uint16_t *twig_dev)
{
    if (w.device < 0 || chunk < 0 || chunk >= (int)w.chunks.size()) return SVO_ERR_INVALID_ARG;
    HIP_TRY(hipSetDevice(w.device));
    HIP_TRY(hipDeviceSynchronize());
    ChunkPools &c = w.chunks[(size_t)chunk];
    const uint64_t trees = meta.trees_on_device, twigs = meta.twigs_on_device;
    std::memcpy(c.position, meta.position, sizeof c.position);
    c.size = meta.size; c.depth = meta.depth;
    c.tree_capacity = std::max(c.tree_capacity, meta.tree_capacity);    // (svo_world_update keeps the slot's capacity as the floor too)
    c.twig_capacity = std::max(c.twig_capacity, meta.twig_capacity);
    std::vector<uint32_t>().swap(c.tree);
    std::vector<uint16_t>().swap(c.twig);
    c.trees_on_device = trees; c.twigs_on_device = twigs;
    classify_world(w);
    DevChunk &e = w.table[(size_t)chunk];
    const bool tree_fits = trees <= w.tree_slot[(size_t)chunk], twig_fits = twigs <= w.twig_slot[(size_t)chunk];
    const bool trace = std::getenv("SVO_BUILD_TIMING") != nullptr;
    if (trace) std::fprintf(stderr, "[svo install] chunk %d: %s\n", chunk, tree_fits && twig_fits ? "in place" : "outgrew its slot");
    if (!tree_fits || !twig_fits) {
        const uint64_t tbase = ((w.tree_pool_len + 8) & ~(uint64_t)7) - 1;
        const uint64_t need_t = tree_fits ? 0 : c.tree_capacity, need_b = twig_fits ? 0 : c.twig_capacity;
        if ((!tree_fits && tbase + need_t > w.tree_pool_cap) || (!twig_fits && w.twig_pool_len + need_b > w.twig_pool_cap)) {
            // no room: this chunk's pools come to the host, then everything is fetched and packed afresh
            if (trace) std::fprintf(stderr, "[svo install] chunk %d: no room at the tails, packing the world again\n", chunk);
            c.tree.resize(trees); c.twig.resize(twigs * TWIG_WORDS);
            c.trees_on_device = c.twigs_on_device = 0;
            HIP_TRY(hipMemcpy(c.tree.data(), tree_dev, trees * sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (twigs) HIP_TRY(hipMemcpy(c.twig.data(), twig_dev, twigs * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToHost));
            return world_upload_impl(&w, w.device, true);
        }
        if (!tree_fits) { e.tree_off = tbase; w.tree_slot[(size_t)chunk] = need_t; w.tree_pool_len = tbase + need_t; }
        if (!twig_fits) { e.twig_off = w.twig_pool_len; w.twig_slot[(size_t)chunk] = need_b; w.twig_pool_len += need_b; }
    }
    e.levels = c.depth - TWIG_LEVELS;
    e.bmin[0] = c.position[0]; e.bmin[1] = c.position[1]; e.bmin[2] = c.position[2];
    HIP_TRY(hipMemcpy(w.d_tree + e.tree_off, tree_dev, trees * sizeof(uint32_t), hipMemcpyDeviceToDevice));
    if (twigs) {
        HIP_TRY(hipMemcpy(w.d_twig + e.twig_off * TWIG_WORDS, twig_dev, twigs * TWIG_WORDS * sizeof(uint16_t), hipMemcpyDeviceToDevice));
        const int rc = launch_masks(w, e.twig_off, twigs, nullptr);
        if (rc != SVO_OK) return rc;
    }
    HIP_TRY(hipMemcpy(w.d_chunks + chunk, &e, sizeof(DevChunk), hipMemcpyHostToDevice));
    const int rc = rebuild_wide_chunk(w, chunk, nullptr);      // (failure: the wide pool is dropped, the installed chunk stays - see svo_world_update)
    HIP_TRY(hipDeviceSynchronize());
    return rc == SVO_OK ? SVO_OK : SVO_OK_LITERAL_ONLY;
}

} // namespace svo

extern "C" {

// ---------------------------------------------------------------------------------------------
static int fill_common(svo_world *w, const svo_trace_params *prm, TraceArgs &A)
{
    if (!w) return SVO_ERR_INVALID_ARG;
    if (w->device < 0) { set_error("svo_trace: world is not uploaded"); return SVO_ERR_NOT_UPLOADED; }
    std::memset(&A, 0, sizeof A);
    // src/Traverse.cpp:129-133
    const float cs = (float)w->chunksize;
    const int ics = (int)cs;
    const int dims[3] = { w->width, w->height, w->depth };
    for (int a = 0; a < 3; ++a) {
        A.worldmin[a] = (float)(w->chunkcoordmin[a] * ics);
        A.worldmax[a] = (float)(w->chunkcoordmin[a] + dims[a]) * cs;
    }
    A.chunksize = cs;
    A.inv_chunksize = 1.0f / cs;
    A.dimw = w->width; A.dimh = w->height; A.dimd = w->depth;
    for (int a = 0; a < 3; ++a) {
        A.ccm[a] = w->chunkcoordmin[a];
        A.cbase[a] = (dims[a] + (w->chunkcoordmin[a] % dims[a])) % dims[a];
    }
    A.chunks = w->d_chunks; A.tree = w->d_tree; A.twig = w->d_twig; A.mask = w->d_mask;
    A.wchunks = w->d_wchunks; A.wide = w->d_wide; A.wbase = w->d_wbase; A.bmat = w->d_bmat;
    if (prm && prm->semantics != SVO_SEMANTICS_CPU && prm->semantics != SVO_SEMANTICS_GLSL) { set_error("svo_trace: unknown semantics"); return SVO_ERR_INVALID_ARG; }
    const bool glsl = prm && prm->semantics == SVO_SEMANTICS_GLSL;      // defaults: src/Traverse.cpp:8,54,79,142 / shaders/Chunkmarch.glsl:1-3,17
    A.glsl = glsl ? 1 : 0;
    A.eps = (prm && prm->eps != 0.0f) ? prm->eps : (glsl ? 1.0f / 4096.0f : 1.0f / 8192.0f);
    A.cap_chunk = (prm && prm->max_chunk_steps > 0) ? prm->max_chunk_steps : (glsl ? 256 : 1000);
    A.cap_tree = (prm && prm->max_tree_steps > 0) ? prm->max_tree_steps : (glsl ? 512 : 1000);
    A.cap_twig = (prm && prm->max_twig_steps > 0) ? prm->max_twig_steps : (glsl ? 64 : 1000);
    A.guard_eps = glsl ? A.eps : -INFINITY;
    A.leaf_back = glsl ? 0.0f : A.eps;
    A.shadow = (prm && prm->shadow) ? 1 : 0;
    A.normal_mode = (prm && prm->normal_mode == SVO_NORMAL_FACE) ? SVO_NORMAL_FACE : SVO_NORMAL_CUBE;
    float l[3] = { 1.0f, -1.0f, 0.0f };                                  // src/Main.cpp:116 (normalised below)
    if (prm && (prm->light_dir[0] != 0.0f || prm->light_dir[1] != 0.0f || prm->light_dir[2] != 0.0f))
        std::memcpy(l, prm->light_dir, sizeof l);
    // shadow direction = normalize(-light_dir), glm::normalize semantics
    const float nx = -l[0], ny = -l[1], nz = -l[2];
    const float inv = 1.0f / std::sqrt(nx * nx + ny * ny + nz * nz);
    A.sdir[0] = nx * inv; A.sdir[1] = ny * inv; A.sdir[2] = nz * inv;
    A.counters = prm ? prm->counters_dev : nullptr;
    A.exact_geometry = w->exact_geometry ? 1 : 0;
    A.tile_cost = prm ? prm->tile_cost_dev : nullptr;
    A.tile_order = prm ? prm->tile_order_dev : nullptr;
    A.work = w->d_work;                 // the launch picks its slot
    return SVO_OK;
}

// The stack kernel's default instantiation addresses wide-tree entries by a 32-bit byte offset into the wide pool and brick masks by
// a 32-bit byte offset into the mask pool (8 B per brick): pools of 2^30 entries (4 GiB; the benchmark world has 0.2 G) / 2^29 bricks
// and more are marched by the large-world instantiation (64-bit addresses; until round 4 by the literal kernel, 5-17 times slower).
static bool stack_needs_big(const svo_world *w)
{
#ifdef SVO_FORCE_WIDE64
    return true;                    // (the `wide64` variant of the Makefile: the large-world kernel on every world, for the tests)
#else
    return w->wide_pool_cap * 64 >= (1ull << 30) || w->twig_pool_cap >= (1ull << 29);
#endif
}

static int pick_kernel(const svo_world *w, const svo_trace_params *prm, const TraceArgs &A)
{
    const int want = prm ? prm->kernel : SVO_KERNEL_AUTO;
    // (brick indices and wide node indices are 32-bit in the kernel at any size: fewer than 2^32 bricks / wide nodes per world)
    const bool stack_ok = w->exact_geometry && w->max_levels <= (int)WIDE_MAX_LEVELS && w->wide_ok && w->twig_pool_cap < (1ull << 32);
    if (want == SVO_KERNEL_LITERAL) return SVO_KERNEL_LITERAL;
    if (want == SVO_KERNEL_STACK) {
        if (!stack_ok) { set_error("svo_trace: SVO_KERNEL_STACK needs exact geometry, chunk depth <= 24 and the world's wide trees (svo_world_info.wide_nodes)"); return SVO_ERR_UNSUPPORTED; }
        return SVO_KERNEL_STACK;
    }
    if (want != SVO_KERNEL_AUTO) { set_error("svo_trace: unknown kernel id"); return SVO_ERR_INVALID_ARG; }
    return (stack_ok && !A.counters) ? SVO_KERNEL_STACK : SVO_KERNEL_LITERAL;
}

static int launch(svo_world *w, const svo_trace_params *prm, TraceArgs &A, hipStream_t s)
{
    const int kernel = pick_kernel(w, prm, A);
    if (kernel < 0) return kernel;
    HIP_TRY(hipSetDevice(w->device));
    // every launch gets its own {tile cursor, ray count} slot so that launches on different streams may overlap
    w->work_last = w->work_next;
    w->work_next = (w->work_next + 1) % WORK_SLOTS;
    A.work = w->d_work + WORK_SLOT_WORDS * w->work_last;
    // a slot coming round again must not be reset under a launch that still reads it: order behind that launch
    if (w->work_event.size() != WORK_SLOTS) w->work_event.assign(WORK_SLOTS, nullptr);
    hipEvent_t &ev = reinterpret_cast<hipEvent_t &>(w->work_event[w->work_last]);
    if (ev) HIP_TRY(hipStreamWaitEvent(s, ev, 0));
    else HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipMemsetAsync(A.work, 0, WORK_SLOT_WORDS * sizeof(unsigned long long), s));
    if (A.n <= 0) return hipEventRecord(ev, s) == hipSuccess ? SVO_OK : SVO_ERR_HIP;
    if (kernel == SVO_KERNEL_LITERAL) {
        const int64_t blocks = (A.n + 255) / 256;
        if (blocks > 0x7FFFFFFF) { set_error("svo_trace: too many rays for one launch"); return SVO_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(k_trace_literal, dim3((unsigned)blocks), dim3(256), 0, s, A);
    } else {
        if (A.ntiles > (1 << 25)) { set_error("svo_trace: more than 2^31 rays in one stack-kernel launch"); return SVO_ERR_UNSUPPORTED; }
        if (A.tile_cost) HIP_TRY(hipMemsetAsync(A.tile_cost, 0, (size_t)A.ntiles * (size_t)(A.from_camera ? A.nframes : 1) * 2 * sizeof(uint32_t), s));
        int rc;
        const int tpw = prm ? prm->tiles_per_wave : 0, nfl = prm ? prm->launches_in_flight : 0;
        // the large-world instantiation (64-bit wide-tree and mask addresses) where 32-bit offsets do not reach: two depth classes
        // of it are compiled (a world that large is built of deep chunks)
        if (stack_needs_big(w)) rc = w->max_levels <= 10 ? launch_stack<10, true>(w, A, tpw, nfl, s) : launch_stack<22, true>(w, A, tpw, nfl, s);
        else if (w->max_levels <= 6) rc = launch_stack<6, false>(w, A, tpw, nfl, s);
        else if (w->max_levels <= 10) rc = launch_stack<10, false>(w, A, tpw, nfl, s);
        else if (w->max_levels <= 16) rc = launch_stack<16, false>(w, A, tpw, nfl, s);
        else rc = launch_stack<22, false>(w, A, tpw, nfl, s);
        if (rc != SVO_OK) { set_error("svo_trace: device query failed"); return rc; }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ev, s));
    return SVO_OK;
}

static int fill_cameras(const svo_camera *cams, int nframes, TraceArgs &A)
{
    if (!cams || nframes < 1 || nframes > MAX_FRAMES) { set_error("svo_trace: between 1 and 16 cameras per launch"); return SVO_ERR_INVALID_ARG; }
    for (int f = 0; f < nframes; ++f) {
        const svo_camera &c = cams[f];
        if (c.width <= 0 || c.height <= 0) { set_error("svo_trace: bad camera"); return SVO_ERR_INVALID_ARG; }
        if (c.width != cams[0].width || c.height != cams[0].height) { set_error("svo_trace_frames: the cameras of one launch share one image size"); return SVO_ERR_INVALID_ARG; }
        FrameCam &d = A.cams[f];
        std::memcpy(d.eye, c.eye, sizeof d.eye); std::memcpy(d.fwd, c.forward, sizeof d.fwd);
        std::memcpy(d.right, c.right, sizeof d.right); std::memcpy(d.up, c.up, sizeof d.up);
        d.tanx = c.tan_half_x; d.tany = c.tan_half_y;
    }
    A.from_camera = 1;
    A.nframes = nframes;
    A.imgw = cams[0].width; A.imgh = cams[0].height;
    return SVO_OK;
}

// Camera-mode launch of A.nframes frames.  The stack kernel marches them behind one set of cursors (its persistent
// waves drain once per launch, not once per frame); any other kernel gets one launch per frame.
static int launch_frames(svo_world *w, const svo_trace_params *prm, TraceArgs &A, const char *who, hipStream_t s)
{
    A.n = (int64_t)A.w * A.h;
    A.tiles_per_row = (A.w + TILE_W - 1) / TILE_W;
    const int64_t tiles = (int64_t)A.tiles_per_row * ((A.h + TILE_H - 1) / TILE_H);
    if (tiles * A.nframes > 0x3FFFFFFF || A.n * A.nframes > 0x7FFFFFFF) { set_error(std::string(who) + ": image too large"); return SVO_ERR_UNSUPPORTED; }
    A.ntiles = (int32_t)tiles;
    if (A.nframes == 1) return launch(w, prm, A, s);
    const int kernel = pick_kernel(w, prm, A);
    if (kernel < 0) return kernel;
    if (kernel == SVO_KERNEL_STACK) return launch(w, prm, A, s);
    if (prm && prm->counters_dev) { set_error(std::string(who) + ": work counters are per launch of one frame"); return SVO_ERR_UNSUPPORTED; }
    const int nframes = A.nframes;
    char *out = static_cast<char *>(A.out);
    for (int f = 0; f < nframes; ++f) {
        TraceArgs F = A;
        F.nframes = 1; F.cams[0] = A.cams[f];
        F.out = out + (size_t)f * (size_t)A.n * sizeof(svo_hit);
        const int rc = launch(w, prm, F, s);
        if (rc != SVO_OK) return rc;
    }
    return SVO_OK;
}

int svo_trace_frames(svo_world *w, const svo_camera *cams, int nframes, const svo_trace_params *prm,
                     int x0, int y0, int rw, int rh, svo_hit *out_dev, void *stream)
{
    TraceArgs A;
    int rc = fill_common(w, prm, A);
    if (rc != SVO_OK) return rc;
    if ((rc = fill_cameras(cams, nframes, A)) != SVO_OK) return rc;
    if (!out_dev || rw < 0 || rh < 0 || x0 < 0 || y0 < 0) { set_error("svo_trace: bad rectangle or output"); return SVO_ERR_INVALID_ARG; }
    A.x0 = x0; A.y0 = y0; A.w = rw; A.h = rh;
    A.bh = rh > 0 ? rh : 1; A.ystep = 0;
    A.out = out_dev;
    return launch_frames(w, prm, A, "svo_trace", (hipStream_t)stream);
}

int svo_trace(svo_world *w, const svo_camera *cam, const svo_trace_params *prm,
              int x0, int y0, int rw, int rh, svo_hit *out_dev, void *stream)
{
    return svo_trace_frames(w, cam, 1, prm, x0, y0, rw, rh, out_dev, stream);
}

int svo_trace_rows_frames(svo_world *w, const svo_camera *cams, int nframes, const svo_trace_params *prm,
                          int band0, int band_stride, int nbands, int band_height, svo_hit *out_dev, void *stream)
{
    TraceArgs A;
    int rc = fill_common(w, prm, A);
    if (rc != SVO_OK) return rc;
    if ((rc = fill_cameras(cams, nframes, A)) != SVO_OK) return rc;
    if (!out_dev || band0 < 0 || band_stride <= 0 || nbands < 0 || band_height <= 0) { set_error("svo_trace_rows: bad band partition"); return SVO_ERR_INVALID_ARG; }
    A.x0 = 0; A.y0 = band0 * band_height; A.w = cams[0].width; A.h = nbands * band_height;
    A.bh = band_height; A.ystep = band_stride * band_height;
    A.out = out_dev;
    return launch_frames(w, prm, A, "svo_trace_rows", (hipStream_t)stream);
}

int svo_trace_rows(svo_world *w, const svo_camera *cam, const svo_trace_params *prm,
                   int band0, int band_stride, int nbands, int band_height, svo_hit *out_dev, void *stream)
{
    return svo_trace_rows_frames(w, cam, 1, prm, band0, band_stride, nbands, band_height, out_dev, stream);
}

int svo_trace_rays(svo_world *w, const float *origins_dev, const float *dirs_dev, int64_t n,
                   const svo_trace_params *prm, svo_hit *out_dev, void *stream)
{
    TraceArgs A;
    int rc = fill_common(w, prm, A);
    if (rc != SVO_OK) return rc;
    if (n < 0 || (n > 0 && (!origins_dev || !dirs_dev || !out_dev))) { set_error("svo_trace_rays: bad ray list"); return SVO_ERR_INVALID_ARG; }
    A.from_camera = 0; A.nframes = 1;
    A.origins = origins_dev; A.dirs = dirs_dev;
    A.n = n;
    A.w = 64; A.h = 1; A.bh = 1; A.tiles_per_row = 1;
    const int64_t tiles = (n + 63) / 64;
    if (tiles > 0x3FFFFFFF) { set_error("svo_trace_rays: too many rays"); return SVO_ERR_UNSUPPORTED; }
    A.ntiles = (int32_t)tiles;
    A.out = out_dev;
    return launch(w, prm, A, (hipStream_t)stream);
}

int svo_tile_order(svo_world *w, const uint32_t *cost_dev, uint32_t *order_dev, int ntiles, void *stream)
{
    if (!w || !cost_dev || !order_dev || ntiles < 0) return SVO_ERR_INVALID_ARG;
    if (w->device < 0) return SVO_ERR_NOT_UPLOADED;
    if (ntiles == 0) return SVO_OK;
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(hipSetDevice(w->device));
    // scratch of the sort: keys in / out, indices in, and hipcub's own (cached in the handle, grown on demand)
    size_t cub_bytes = 0;
    uint32_t *nul = nullptr;
    if (hipcub::DeviceRadixSort::SortPairsDescending(nullptr, cub_bytes, nul, nul, nul, nul, ntiles, 0, 32, s) != hipSuccess) return SVO_ERR_HIP;
    const size_t need = (size_t)ntiles * 3 * sizeof(uint32_t) + cub_bytes + 256;
    if (need > w->sort_bytes) {
        if (w->d_sort) { HIP_TRY(hipDeviceSynchronize()); (void)hipFree(w->d_sort); w->d_sort = nullptr; w->sort_bytes = 0; }
        if (hipMalloc(&w->d_sort, need) != hipSuccess) { set_error("svo_tile_order: hipMalloc failed"); return SVO_ERR_OUT_OF_MEMORY; }
        w->sort_bytes = need;
    }
    // One scratch per world: calls on different streams (one cost / order pair per launch in flight is the intended use) are
    // ordered behind one another here, like the work slots of the launches - a sort that shared its keys with another need not
    // even yield a permutation.
    hipEvent_t &sorted = reinterpret_cast<hipEvent_t &>(w->sort_event);
    if (sorted) HIP_TRY(hipStreamWaitEvent(s, sorted, 0));
    else HIP_TRY(hipEventCreateWithFlags(&sorted, hipEventDisableTiming));
    uint32_t *keys = static_cast<uint32_t *>(w->d_sort), *keys_out = keys + ntiles, *idx = keys_out + ntiles;
    void *tmp = reinterpret_cast<char *>(idx + ntiles) + ((256 - ((size_t)ntiles * 12) % 256) % 256);
    hipLaunchKernelGGL(k_tile_keys, dim3((unsigned)((ntiles + 255) / 256)), dim3(256), 0, s, cost_dev, keys, idx, ntiles);
    HIP_TRY(hipGetLastError());
    if (hipcub::DeviceRadixSort::SortPairsDescending(tmp, cub_bytes, keys, keys_out, idx, order_dev, ntiles, 0, 32, s) != hipSuccess) { set_error("svo_tile_order: sort failed"); return SVO_ERR_HIP; }
    HIP_TRY(hipEventRecord(sorted, s));
    return SVO_OK;
}

int svo_trace_last_ray_count(svo_world *w, void *stream, uint64_t *rays)
{
    if (!w || !rays) return SVO_ERR_INVALID_ARG;
    if (w->device < 0) return SVO_ERR_NOT_UPLOADED;
    HIP_TRY(hipSetDevice(w->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    unsigned long long v[2] = { 0, 0 };
    HIP_TRY(hipMemcpy(v, w->d_work + WORK_SLOT_WORDS * w->work_last, sizeof v, hipMemcpyDeviceToHost));
    *rays = v[1];
    return SVO_OK;
}

} // extern "C"
