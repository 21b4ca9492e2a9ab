// kernel_stack.hip.h — SVO_KERNEL_STACK: the CDNA4 fast path of the SVO march.
//
// Same results as kernel_literal.hip.h / src/Traverse.cpp, bit for bit, on worlds whose voxel
// corners are exact floats (svo_world_info.exact_geometry: power-of-two chunk edge, positions on
// the voxel lattice), reached differently:
//
//  * Persistent single-wave workgroups pull 8x8-pixel tiles (64 rays) from atomic cursors, one per screen region;
//    a wave starts in the region of its XCD (HW_REG_XCC_ID), so each XCD's L2 serves one part of the world, and
//    moves on to the other regions when its own is used up.  One launch may carry several frames (cameras): a
//    region's cursor then runs through its tiles frame after frame, and the waves drain once per launch.
//    When >= REFILL lanes of the wave have retired (64-bit __ballot), the dead lanes are handed
//    the next ray ids by prefix rank (mbcnt) — the wave re-compacts instead of idling on its
//    slowest ray.  A primary hit turns its lane into the shadow ray in place.
//  * The three nested loops of chunkmarch/treemarch/twigmarch are flattened into one loop whose
//    body every lane runs in one of three modes, sharing one escape-distance evaluation.
//  * Descent: the position inside the chunk is reduced to integer cell coordinates at level
//    depth-2 (one exact compare-and-fix per axis reproduces the reference's float `p >= mid`
//    tests) and looked up in the chunk's WIDE tree (wide_tree.hip.h: built on upload from tree[], 64 entries per node
//    = two reference levels per dependent load).  The wide node of every wide level of the current path is cached in
//    a per-lane LDS column ([level][lane]: bank = lane, conflict-free); the next step restarts at the deepest level
//    whose coordinate prefix is unchanged - and keeps the cache across rays of the same chunk (a shadow ray starts
//    where its primary ray ended).  ONE dependent load per lane and step: a lane whose entry is a BRANCH pushes the child
//    onto its cache and sits the step out (step_asm.hip.h) - the wave never waits for a second level, unless a quarter of
//    its tree lanes need one.
//  * Bricks are tested against a 64-bit occupancy mask held in registers (one 8-byte load per
//    brick visit); the 128-byte brick line is touched only to fetch the hit material.
//  * Ray generation (camera ray, 1/dir, world-entry distance: ~10 IEEE divisions) is done once per
//    tile by all 64 lanes together and parked in LDS; a refill is then 11 LDS reads per lane, so
//    waves can refill after only a few retirements.
//  * The rare, expensive blocks (chunk step, G-buffer resolve of a primary hit) run only when a
//    ballot shows enough lanes waiting for them (or nothing else is runnable).
//  * Creeping rays - pinned on a lattice plane, advancing by EPS alone for thousands of steps, alone or nested over a
//    brick whose marches are bound to miss - are taken in closed form by a vote-gated block (same t, same counters).
//  * The step runs four times per pass of the outer loop while >= 8 lanes are marching (decided before the loop: nothing
//    is voted on between the steps, the loop control is scalar), and keeps repeating once the tile cursors are dry and
//    every live lane is marching: the refill / vote / block checks around it are a quarter of an iteration, and a
//    launch's last waves are alone on their SIMD, bound by their own instruction stream.  Scalar instructions and
//    branches cost a SIMD what vector instructions do (scripts/microbench/valu_issue.hip): the step avoids them
//    where a select or an unconditional LDS read does the same.
//  * The step itself is one hand-scheduled asm statement that runs the steps of a pass (step_asm.hip.h; the C++ step below
//    is the readable statement of the same algorithm, built with -DSVO_CXX_STEP): what a wave-step costs is its chain of
//    dependent loads and the instructions between them, and that chain is what the statement is arranged around.
//  * Lane state is kept small (re-basing points pw/pt and the node box are recomputed with the
//    reference's own expressions instead of being held): 80 VGPRs, 6 waves per SIMD, no spill inside the pass loop.
//  * The chunk table of worlds of up to 64 chunks sits in LDS: the chunk step waits for no global load.
//  * Instantiations (round 4): BIG - 64-bit wide-tree and mask addresses for worlds beyond 4 GiB of wide nodes / 2^29 bricks, a
//    128-chunk LDS table; GLSL - the march of shaders/Chunkmarch.glsl (svo_trace_params.semantics): guarded escape distance, LEAF
//    hits at t, tnear > 0 at the world entry, no containment re-check.  The asm statement (step_asm_body.inc) and the creep block
//    (creep_block.inc) are shared text.
//
// All float arithmetic that decides t is evaluated exactly as in the reference; only loads and
// integer bookkeeping differ.  Divisions by powers of two (chunk edge, node edge) are written as
// multiplications by the exact reciprocal, which is bit-identical.
#pragma once
#include <type_traits>

#include "march.hip.h"
#include "step_asm.hip.h"

namespace svo {

constexpr int CW_ESCAPE_PENDING = (int)0x80000000;    // bit of the lane's chunk-step counter: tw still lacks the escape out of the chunk just left
enum : int { M_DONE = 0, M_WORLD = 1, M_HIT = 2, M_TREE = 3, M_TWIG = 4 };     // (step_asm.hip.h: marching = mode > 2)

#ifndef SVO_VOTE_WORLD
#define SVO_VOTE_WORLD 8         // lanes waiting for a chunk step that make the wave run it
#endif
#ifndef SVO_VOTE_HIT
#define SVO_VOTE_HIT 16          // primary hits waiting for their G-buffer record
#endif
#ifndef SVO_VOTE_BUSY
#define SVO_VOTE_BUSY 24         // fewer marching lanes than this: serve the waiting ones regardless
#endif
// Creeping rays (see the creep block in the kernel): a lane whose last SVO_CREEP_SERIOUS steps all advanced by ~EPS makes
// the wave run the creep block; SVO_CREEP_LANES creeping lanes do so at once.
#ifndef SVO_CREEP_SERIOUS
#define SVO_CREEP_SERIOUS 6
#endif
#ifndef SVO_CREEP_LANES
#define SVO_CREEP_LANES 8
#endif
// the step runs up to 1 + SVO_STEP_EXTRA times per pass of the outer loop while at least SVO_STEP_LANES lanes are marching
// Block placement: a taken branch costs a shared SIMD about three vector instructions' time, a not-taken one about one
// (scripts/microbench/valu_issue.hip) - rare blocks go out of line so that the common path falls through.
#ifdef SVO_NO_EXPECT
#define SVO_LIKELY(x) (x)
#define SVO_UNLIKELY(x) (x)
#else
#define SVO_LIKELY(x) __builtin_expect(!!(x), 1)
#define SVO_UNLIKELY(x) __builtin_expect(!!(x), 0)
#endif
#ifndef SVO_STEP_EXTRA
#define SVO_STEP_EXTRA 4
#endif
// steps per statement while the inner repeat lasts (the drain of a launch: the wave is alone on its SIMD, every instruction of the
// loop control around the statement costs it ~5 cycles)
#ifndef SVO_DRAIN_STEPS
#define SVO_DRAIN_STEPS 8        // (round 4: 8 instead of 4 - one frame 1.62 - 1.63 ms against 1.655 - 1.659, the serialized 16-frame launch 8.79 against 8.91 - 8.99 ms; 16: the same as 8, 2: as 4)
#endif
// when a wave skips the bricks whose march is bound to miss (step_asm_body.inc, "sure"): the test is ~95 instructions for every wave-step
// in which some lane enters a brick, so a wave of the bulk would pay more than its lanes win; a draining wave's instructions are its
// critical path and the test runs while the brick's mask is on its way
#ifndef SVO_SURE_MISS_WHEN
#define SVO_SURE_MISS_WHEN __builtin_amdgcn_readfirstlane(more ? 0 : 1)
#endif
#ifndef SVO_STEP_LANES
#define SVO_STEP_LANES 8
#endif

// 1/x for x an exact power of two (normal range): exponent negation, no division sequence.
__device__ __forceinline__ float recip_pow2(float x) { return __uint_as_float(0x7F000000u - __float_as_uint(x)); }

// World::index(World::index_float(p)) for a power-of-two chunk edge (src/World.cpp:288-293,323-332):
// p / chunksize == p * inv (exact), and positive_mod() of a coordinate that lies within one grid
// period of the grid's first chunk is a conditional add/subtract.  Only for p inside the world box.
__device__ __forceinline__ int chunk_index_pow2(const TraceArgs &A, V3 p)
{
    float qx = p.x * A.inv_chunksize, qy = p.y * A.inv_chunksize, qz = p.z * A.inv_chunksize;
    if (qx < 0.0f) qx -= 1.0f;
    if (qy < 0.0f) qy -= 1.0f;
    if (qz < 0.0f) qz -= 1.0f;
    const int ix = (int)qx, iy = (int)qy, iz = (int)qz;
    // p lies inside the world box (the caller's isInsideCube test, src/Traverse.cpp:145): wlo <= p <= whi with
    // wlo = ccm * chunksize and whi = (ccm + dims) * chunksize exactly, so ccm - 1 <= i <= ccm + dims per axis (the - 1:
    // index_float sends an exact negative multiple one chunk down) - within one grid period, no general modulo needed
    const int rx = ix - A.ccm[0], ry = iy - A.ccm[1], rz = iz - A.ccm[2];
    int mx = A.cbase[0] + rx, my = A.cbase[1] + ry, mz = A.cbase[2] + rz;
    mx -= (mx >= A.dimw) ? A.dimw : 0; mx += (mx < 0) ? A.dimw : 0;
    my -= (my >= A.dimh) ? A.dimh : 0; my += (my < 0) ? A.dimh : 0;
    mz -= (mz >= A.dimd) ? A.dimd : 0; mz += (mz < 0) ? A.dimd : 0;
    return my * A.dimw * A.dimd + mz * A.dimw + mx;
}

// cubeNormal (shaders/Chunkmarch.glsl:128-136) for a cube whose half edge is a power of two:
// p / d == p * (1/d) exactly, and normalize() of a vector of small integers multiplies by
// 1/sqrt(1|2|3), whose correctly rounded values are constants.
__device__ __forceinline__ V3 cube_normal_pow2(V3 s, V3 lo, float size, float eps)
{
    const V3 hi = lo + size;
    const V3 c = (lo + hi) * 0.5f;
    const V3 p = s - c;
    const float d = fabsf(lo.x - hi.x) * 0.5f;                 // same on every axis: a cube
    const float invd = recip_pow2(d);                           // power of two: exact
    const float b = 1.0f + eps;
    const float ix = (float)(int)((p.x * invd) * b), iy = (float)(int)((p.y * invd) * b), iz = (float)(int)((p.z * invd) * b);
    const float dot = ix * ix + iy * iy + iz * iz;
    float inv;
    if (dot == 1.0f) inv = 1.0f;
    else if (dot == 2.0f) inv = __uint_as_float(0x3F3504F3u);  // 1.0f / sqrtf(2.0f)
    else if (dot == 3.0f) inv = __uint_as_float(0x3F13CD3Au);  // 1.0f / sqrtf(3.0f)
    else inv = 1.0f / sqrtf(dot);                               // 0 -> inf -> NaN normal, as in the reference
    return mk(ix * inv, iy * inv, iz * inv);
}

// Octree::branch (src/Octree.cpp:55-58) of the cell (ux, uy, uz) at the level whose selector bit is `sh`.
__device__ __forceinline__ uint32_t child_slot(int ux, int uy, int uz, int sh)
{
    const uint32_t bx = __builtin_amdgcn_ubfe((uint32_t)ux, (uint32_t)sh, 1u);
    const uint32_t by = __builtin_amdgcn_ubfe((uint32_t)uy, (uint32_t)sh, 1u);
    const uint32_t bz = __builtin_amdgcn_ubfe((uint32_t)uz, (uint32_t)sh, 1u);
    return ((bz << 1) + by) * 2u + bx;
}
// node word at byte offset `boff` of the tree pool: scalar base + zero-extended 32-bit vector offset (one VGPR of address
// instead of a 64-bit pointer per lane; pools of 4 GiB and more are marched by the BIG instantiation, which addresses with 64 bits)
__device__ __forceinline__ uint32_t ld_node(const uint32_t *pool, uint32_t boff)
{
    return *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pool) + (size_t)boff);
}
// entry of a wide node (wide_tree.hip.h) for the cell (ux, uy, uz): its two coordinate bits at `sh` per axis, x lowest
constexpr uint32_t WIDE_PAYLOAD = (1u << 25) - 1u;
__device__ __forceinline__ uint32_t wide_slot(int ux, int uy, int uz, int sh)
{
    const uint32_t bx = __builtin_amdgcn_ubfe((uint32_t)ux, (uint32_t)sh, 2u);
    const uint32_t by = __builtin_amdgcn_ubfe((uint32_t)uy, (uint32_t)sh, 2u);
    const uint32_t bz = __builtin_amdgcn_ubfe((uint32_t)uz, (uint32_t)sh, 2u);
    return (bz << 4) | (by << 2) | bx;
}
// type == BRANCH (binary 10 in the top bits) as one signed comparison
__device__ __forceinline__ bool is_branch(uint32_t word) { return (int32_t)word < (int32_t)0xC0000000; }

// The launch arguments as they lie in the kernarg segment, behind a barrier the optimiser cannot see through: a rare
// block that reads its arguments through this re-loads them with scalar loads where it runs, instead of pinning
// ~60 SGPRs (camera, world grid) across the march loop, which spilled into VGPR lanes (v_readlane in the hot path).
__device__ __forceinline__ TraceArgs args_reloaded()
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    TraceArgs T;
    __builtin_memcpy(&T, p, sizeof T);              // only the fields the caller uses survive (scalar loads)
    return T;
}
// camera f of the launch, read where it is needed with scalar loads at a uniform offset
__device__ __forceinline__ FrameCam camera_reloaded(int f)
{
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    FrameCam c;
    __builtin_memcpy(&c, (decltype(p))((const char __attribute__((address_space(4))) *)p + __builtin_offsetof(TraceArgs, cams) + (size_t)f * sizeof(FrameCam)), sizeof c);
    return c;
}

// How far along the ray a + b*s (s >= 0) can brick marches go before one of them might see an occupied cell?
// Asked for a run of tree steps over one TWIG node whose start points p_j = a + b*s_j all lie ON the brick's lower
// face of axis k (p_j.k == bmin.k) with b.k < 0; every twigmarch(p_j, b, brick) (src/Traverse.cpp:50-72) then
//  * only sees cells with coordinate 0 on axis k: p.k = p_j.k + b.k*t <= bmin.k for t >= 0, so p is in that layer or
//    outside the brick (the march returns false);
//  * takes pinned steps only, t += EPS (see the creep block), and ends once |b.k|*t exceeds half an ulp of bmin.k;
//  * moves forward along the ray: on the other two axes the brick's corner is at least a brick edge away from 0
//    (checked), so p - bmin is exact (Sterbenz), the truncated cell index is the cell p really lies in and no escape
//    distance is negative.
// Returns the smallest s at which the ray, widened by r/16 (>> the rounding of p, checked), touches an occupied cell of
// that layer: +inf if it never does, -1 if a precondition fails.  Marches that stay below that parameter are misses,
// however they end (leaving the brick, step cap, NaN).  r = voxel edge.
#ifndef SVO_NEST_INLINE
#define SVO_NEST_INLINE __noinline__
#endif
__device__ SVO_NEST_INLINE float brick_layer_first_touch(int k, V3 a, V3 b, V3 g, V3 bmin, float r, unsigned long long mask)
{
    // axis k is the pinned one; u, v = the other two (component selection by k, no indexed arrays)
    const float ak = k == 0 ? a.x : k == 1 ? a.y : a.z, bk = k == 0 ? b.x : k == 1 ? b.y : b.z, mk_ = k == 0 ? bmin.x : k == 1 ? bmin.y : bmin.z;
    const float au = k == 0 ? a.y : k == 1 ? a.z : a.x, av = k == 0 ? a.z : k == 1 ? a.x : a.y;
    const float gu = k == 0 ? g.y : k == 1 ? g.z : g.x, gv = k == 0 ? g.z : k == 1 ? g.x : g.y;
    const float mu = k == 0 ? bmin.y : k == 1 ? bmin.z : bmin.x, mv = k == 0 ? bmin.z : k == 1 ? bmin.x : bmin.y;
    const int su = k == 0 ? 4 : k == 1 ? 16 : 1, sv = k == 0 ? 16 : k == 1 ? 1 : 4;      // bit strides of u and v in the brick mask
    const float delta = r * 0.0625f;
    bool ok = (ak == mk_) & (bk < 0.0f);
    ok &= (mu >= 4.0f * r) & (mv >= 4.0f * r) & (delta >= (fabsf(mu) + fabsf(mv) + 8.0f * r) * 0x1p-21f);
    ok = ok && fabsf(gu) < 1.0e30f && fabsf(gv) < 1.0e30f;                // 1/b finite on both axes (no 0 * inf below)
    if (!ok) return -1.0f;
    const float u0 = au - mu, v0 = av - mv;
    float first = __uint_as_float(0x7F800000u);
    bool clean = true;
    for (int cv = 0; cv < 4; ++cv) {
        const float va = ((float)cv * r - delta - v0) * gv, vb = ((float)(cv + 1) * r + delta - v0) * gv;
        const float v_in = fminf(va, vb), v_out = fmaxf(va, vb);
        for (int cu = 0; cu < 4; ++cu) {
            const bool occupied = (mask >> (cu * su + cv * sv)) & 1ull;
            const float ua = ((float)cu * r - delta - u0) * gu, ub = ((float)(cu + 1) * r + delta - u0) * gu;
            const float s_in = fmaxf(fmaxf(fminf(ua, ub), v_in), 0.0f), s_out = fminf(fmaxf(ua, ub), v_out);
            clean &= (s_in == s_in) & (s_out == s_out);               // NaN cannot happen with finite operands; stay on the safe side
            first = (occupied & (s_in <= s_out)) ? fminf(first, s_in) : first;
        }
    }
    return clean ? first * (1.0f - 0x1p-18f) : -1.0f;                  // the slab parameters carry a few ulps of their own
}

// svo_trace_params.tile_cost_dev: the step count of a ray that ends (or turns into its shadow ray) is folded into its tile's
// record - [frame][tile][primary | shadow] maxima, what svo_tile_order sorts the next frame's tiles by.  Rare blocks only.
__device__ __forceinline__ void note_tile_cost(int outk, uint32_t steps)
{
    const TraceArgs T = args_reloaded();
    if (!T.tile_cost) return;
    const int k = outk & 0x7FFFFFFF;
    int tile, frame = 0;
    if (T.from_camera) {
        const int per = T.w * T.h;
        frame = k / per;
        const int r = k - frame * per;
        const int ly = r / T.w, lx = r - ly * T.w;
        tile = (ly / TILE_H) * T.tiles_per_row + lx / TILE_W;
    } else tile = k >> 6;
    atomicMax(&T.tile_cost[((size_t)frame * (size_t)T.ntiles + (size_t)tile) * 2 + (outk < 0 ? 1 : 0)], steps);
}

// BIG: the large-world instantiation - the chunk's wide tree is a 64-bit address per lane and the brick masks are addressed with
// 64 bits (step_asm.hip.h: march_steps_asm_big), for wide pools of 4 GiB and more and mask pools of 2^29 bricks and more; the same
// kernel otherwise, the same results.
// GLSL: the shader twin's march (svo_trace_params.semantics = SVO_SEMANTICS_GLSL; shaders/Chunkmarch.glsl): the guarded escape
// distance and the LEAF hit at t inside the step (step_asm.hip.h), the entry condition and the missing containment re-check here.
template <int MAXLV, int REFILL, int WAVES_PER_SIMD, bool BIG, bool GLSL>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_trace_stack(TraceArgs A)
{
    __shared__ uint32_t stk[MAXLV / 2 + 1][64];     // wide node index per wide level of the current path (level 0 is node 0)
    __shared__ float tile_ray[11][64];
    // the chunk table (bmin, levels, wide offset, brick offset) of worlds of up to 64 chunks - the reference's default is 4x4x4 -
    // staged in LDS: the chunk step then reads it there instead of waiting for a 32-byte global load per lane (the block runs
    // in six of ten passes and a wave waits out the slowest lane's load each time)
    // (the large-world instantiation holds 128 chunks - a 10x1x10 grid of depth-12 chunks is 68 GB of pools - for 1.5 KB more LDS
    // per wave: 22 instead of 24 waves fit a CU's 160 KB)
    constexpr int CHUNK_TAB = (BIG && MAXLV <= 10) ? 128 : 64;      // (the deep instantiation's descent column already takes the room)
    __shared__ uint32_t chunk_tab[6][CHUNK_TAB];
    stk[0][threadIdx.x] = 0u;              // o, d, 1/d, world-entry t, output index (as int; -1 = no ray)
    const int lane = threadIdx.x;
    const bool want_cost = A.tile_cost != nullptr;      // (one SGPR held; the blocks below must not re-read the kernel arguments for a feature that is off)
    const int n_chunks = A.dimw * A.dimh * A.dimd;
#ifdef SVO_NO_LDS_CHUNK_TAB      // experiment (scripts/build_variants.sh): what the chunk table costs once it has left LDS (worlds of more than CHUNK_TAB chunks)
    const bool chunks_in_lds = false;
#else
    const bool chunks_in_lds = n_chunks <= CHUNK_TAB;
#endif
    if (chunks_in_lds)
        for (int i = lane; i < n_chunks; i += 64) {
            const DevWide ch = A.wchunks[i];
            chunk_tab[0][i] = __float_as_uint(ch.bmin[0]); chunk_tab[1][i] = __float_as_uint(ch.bmin[1]); chunk_tab[2][i] = __float_as_uint(ch.bmin[2]);
            chunk_tab[3][i] = ch.levels; chunk_tab[4][i] = ch.wide_off; chunk_tab[5][i] = (uint32_t)ch.twig_off;
        }
    __syncthreads();
#ifdef SVO_STACK_TIMING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned n_iters = 0, n_tree_lanes = 0, n_twig_lanes = 0, n_world_lanes = 0;
    unsigned n_world_runs = 0, n_hit_runs = 0, n_refill = 0, n_tilegen = 0, n_fix = 0;
    unsigned n_lane_twig = 0;
    unsigned n_lane_busy = 0;       // per lane: asm steps it entered a statement for as a marching lane; the wave's maximum is its critical path
    unsigned n_creep_runs = 0, n_creep_steps = 0, n_creep_rounds = 0, n_dbg = 0;   // block runs, lane-steps taken in it (this lane), rounds
    StepStats step_stats;
    unsigned n_wsteps = 0, n_lsteps = 0, n_hit_wait = 0, n_dead_wait = 0, n_wsteps_b = 0, n_lsteps_b = 0, n_world_wait = 0, n_twig_b = 0;   // step bodies executed, marching lanes summed over them; M_HIT / M_DONE lanes summed over them
#endif

    const V3 wlo = ld3(A.worldmin), whi = ld3(A.worldmax);
    const V3 sdir = ld3(A.sdir);
    const V3 sg = recip(sdir);                      // 1/sdir: the same quotient for every shadow ray
    const float eps = A.eps;
    const bool eps_pow2 = (__float_as_uint(eps) & 0x807FFFFFu) == 0u && __float_as_uint(eps) >= 0x00800000u && __float_as_uint(eps) < 0x7F800000u;
    const float csize = A.chunksize;

    // ---- wave state (uniform) ------------------------------------------------------------
    // The image is dealt out as TILE_REGIONS screen regions (column strips of 8x8 tiles; row bands when the raster has
    // fewer than 8 tile columns, e.g. ray lists), each behind its own cursor.  A wave starts in the region of its XCD:
    // an XCD's waves then march rays that traverse the same part of the world, and that XCD's private L2 holds it.
    // A wave whose region is used up moves on to the next one (work stealing).
    const int tr_cols = A.tiles_per_row;
    const int tr_rows = A.ntiles / (tr_cols > 0 ? tr_cols : 1);
    const bool by_cols = tr_cols >= TILE_REGIONS;
    const int reg_q = (by_cols ? tr_cols : tr_rows) / TILE_REGIONS, reg_rem = (by_cols ? tr_cols : tr_rows) % TILE_REGIONS;
    int region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (TILE_REGIONS - 1));    // HW_REG_XCC_ID[3:0]
    int regions_left = TILE_REGIONS;
    int tile_first = 0;             // first ray id of the tile being handed out
    int tile_next = 64;             // next unassigned slot of that tile (64 = exhausted)
    bool more = true;               // tiles left in the global cursor
    unsigned rays_marched = 0;      // per lane, summed at exit

    // ---- lane state ------------------------------------------------------------------------
    int mode = M_DONE;
    int outk = 0;                   // output record of the ray; bit 31 set: the lane marches that pixel's shadow ray
    V3 alpha = mk(0, 0, 0), beta = mk(0, 0, 1), g = mk(0, 0, 0);
    float tw = 0.0f;                // chunkmarch's t (src/Traverse.cpp:135)
    int cw = 0;
    uint32_t guard = 0;
    // The level being marched (tree: src/Traverse.cpp:74-113, brick: :50-72) as one "frame": p = O + beta*t is
    // tested against the box [Blo, Blo+Bsize] and located on a lattice of pitch `res` anchored at Blo.
    // Entering a brick swaps the frame (and parks the tree level's t / counter); leaving swaps it back.
    V3 O = mk(0, 0, 0), Blo = mk(0, 0, 0);
    // (the box edge, 1/res and the step cap follow from mode and res: chunk edge / 4 voxels, A.cap_tree / A.cap_twig)
    float res = 1.0f, t = 0.0f;
    int cnt = 0;                    // steps LEFT of the level's cap (counts down: cap - the reference's loop counter); in M_HIT: the brick cell that was hit (or SVO_CELL_NONE: a LEAF)
    float tt_saved = 0.0f, t_miss = 0.0f;  // tree-level t at the brick's entry; ... after the brick march has missed
    int it_saved = 0;
    // chunk
    V3 clo = mk(0, 0, 0);
    typename std::conditional<BIG, unsigned long long, uint32_t>::type wide_b = 0;   // the chunk's top wide node: byte offset into the wide pool (BIG: its address)
    // entry `e` (64 per wide node, counted from the chunk's top wide node) of the current chunk's wide tree
    auto ld_wide = [&](uint32_t e) -> uint32_t {
        if constexpr (BIG) return *reinterpret_cast<const uint32_t *>((size_t)wide_b + ((size_t)e << 2));
        else return ld_node(A.wide, wide_b + (e << 2));
    };
    uint32_t twig_off = 0;
    int levels = 0, ci = -1;
    // descent cache: cell coordinates of the last tree step and the level of the node it ended at
    // (valid: deepest wide level whose node index is cached; plev: reference level of the node the last tree step ended at)
    int pux = 0, puy = 0, puz = 0, valid = 0, plev = 0;
    // brick
    unsigned long long bmask = 0;
#ifndef SVO_CXX_STEP
    // step_asm.hip.h keeps the level's box edge and the chunk's wide-level count / tree-level pitch instead of deriving them per step
    float bsize = 0.0f, res_tree = 1.0f;
    int nw_chunk = 1;
    const uint32_t lds_lane = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)&stk[0][lane];
    StepUniform SU;
    SU.csize = csize; SU.eps = eps; SU.eps2 = 2.0f * eps; SU.cap_twig = A.cap_twig; SU.wide = A.wide; SU.mask = A.mask; SU.descend_shift = MAXLV <= 10 ? SVO_DESCEND_SHIFT_SHALLOW : SVO_DESCEND_SHIFT_DEEP;
#endif
    // creeping rays: |creepn| = consecutive advances of this ray by less than 2 EPS (kept across level changes: a ray pinned
    // on a chunk face creeps at every level); > 0 only while the cell located last is known to be empty (creep block armed)
    int creepn = 0;

    for (;;) {
        // ==== refill retired lanes =============================================================
        unsigned long long dead = __ballot(mode == M_DONE);
        while (more && __popcll(dead) >= REFILL) {
            if (tile_next >= 64) {
                int t32 = -1, tcol = 0, trow = 0, tframe = 0;   // raster index of the tile, its column and row, its frame
                const uint32_t *order = args_reloaded().tile_order;
                if (order) {                                    // caller's order (longest tiles first): one cursor, frame after frame
                    regions_left = 0;
                    unsigned long long tix = 0;
                    if (lane == 0) tix = atomicAdd(&A.work[WORK_CURSOR0], 1ull);
                    int t = __builtin_amdgcn_readfirstlane((int)tix);
                    if (t < A.ntiles * A.nframes) {
                        tframe = t / A.ntiles; t -= tframe * A.ntiles;
                        t32 = (int)order[t];
                        // the caller's array: anything that is not a tile index reads as "this entry has no tile" (an empty tile:
                        // its 64 slots hold no ray) instead of indexing the raster or the ray list out of range
                        if ((uint32_t)t32 >= (uint32_t)A.ntiles) { t32 = 0; tframe = -1; }
                        trow = t32 / tr_cols; tcol = t32 - trow * tr_cols;
                    }
                }
                while (regions_left > 0) {
                    const int span = reg_q + (region < reg_rem ? 1 : 0);                    // columns (rows) of this region
                    const int first = region * reg_q + (region < reg_rem ? region : reg_rem);
                    const int count = span * (by_cols ? tr_rows : tr_cols);          // tiles of this region in one frame
                    unsigned long long tix = 0;
                    if (lane == 0) tix = atomicAdd(&A.work[WORK_CURSOR0 + region], 1ull);
                    int t = __builtin_amdgcn_readfirstlane((int)tix);
                    if (t < count * A.nframes) {                                    // frame after frame within the region
                        if (A.nframes > 1) { tframe = t / count; t -= tframe * count; }
                        if (by_cols) { trow = t / span; tcol = first + (t - trow * span); }
                        else { trow = first + t / tr_cols; tcol = t % tr_cols; }
                        t32 = trow * tr_cols + tcol;
                        break;
                    }
                    region = (region + 1) & (TILE_REGIONS - 1);
                    --regions_left;
                }
                if (t32 < 0) { more = false; break; }
                tile_first = t32 * 64;
                tile_next = 0;
#ifdef SVO_STACK_TIMING
                ++n_tilegen;
#endif
                // all 64 lanes generate the tile's rays (src/Traverse.cpp:135-140 included) and park them in LDS
                {
                    const TraceArgs T = args_reloaded();
                    const int id = tile_first + lane;
                    bool ok;
                    int k = -1;
                    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
                    if (T.from_camera) {
                        const int lx = tcol * TILE_W + (lane & (TILE_W - 1));
                        const int ly = trow * TILE_H + lane / TILE_W;
                        ok = (lx < T.w) & (ly < T.h) & (tframe >= 0);      // (tframe < 0: an entry of the caller's tile order that names no tile)
                        k = (tframe * T.h + ly) * T.w + lx;
                        int px = 0, py = 0;
                        if (ok) local_to_pixel(T, lx, ly, px, py);
                        if (ok && (py >= T.imgh || px >= T.imgw)) { store_miss(A.out, k, 0); ok = false; }
                        if (ok) camera_ray(camera_reloaded(tframe), T.imgw, T.imgh, px, py, o, d);
                    } else {
                        ok = id < T.n && tframe >= 0;
                        k = id;
                        if (ok) { o = ld3(T.origins + 3 * (long long)id); d = ld3(T.dirs + 3 * (long long)id); }
                    }
                    const V3 gg = recip(d);
                    float t0 = 0.0f;
                    if (ok) {
                        bool hit = true;
                        if (!inside(o, wlo, whi)) t0 = (GLSL ? enter_glsl(o, gg, wlo, whi, hit) : enter(o, d, wlo, whi, hit)) + eps;
                        rays_marched++;
                        if (!hit) { store_miss(A.out, k, 0); ok = false; }
                    }
                    tile_ray[0][lane] = o.x; tile_ray[1][lane] = o.y; tile_ray[2][lane] = o.z;
                    tile_ray[3][lane] = d.x; tile_ray[4][lane] = d.y; tile_ray[5][lane] = d.z;
                    tile_ray[6][lane] = gg.x; tile_ray[7][lane] = gg.y; tile_ray[8][lane] = gg.z;
                    tile_ray[9][lane] = t0;
                    tile_ray[10][lane] = __int_as_float(ok ? k : -1);
                    __syncthreads();                        // one wave per block: orders the LDS writes before the reads
                }
            }
#ifdef SVO_STACK_TIMING
            ++n_refill;
#endif
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dead, 0u));
            const int avail = 64 - tile_next;
            if (mode == M_DONE && rank < avail) {
                const int slot = tile_next + rank;
                const int k = __float_as_int(tile_ray[10][slot]);
                if (k >= 0) {
                    outk = k;                               // (bit 31 clear: primary ray)
                    alpha = mk(tile_ray[0][slot], tile_ray[1][slot], tile_ray[2][slot]);
                    beta = mk(tile_ray[3][slot], tile_ray[4][slot], tile_ray[5][slot]);
                    g = mk(tile_ray[6][slot], tile_ray[7][slot], tile_ray[8][slot]);
                    tw = tile_ray[9][slot];
                    cw = 0; guard = 0; creepn = 0;
                    mode = M_WORLD;
                }
            }
            const int ndead = __popcll(dead);
            tile_next += (ndead < avail) ? ndead : avail;
            __syncthreads();                                // reads done before a later tile overwrites the buffer
            dead = __ballot(mode == M_DONE);
        }
        if (dead == ~0ull) break;                       // nothing alive and nothing left to fetch
#ifdef SVO_STACK_TIMING
        ++n_iters;
        n_tree_lanes += __popcll(__ballot(mode == M_TREE)); n_twig_lanes += __popcll(__ballot(mode == M_TWIG));
        n_world_lanes += __popcll(__ballot(mode == M_WORLD));
#endif

        // ==== votes: which of the rare blocks run this iteration ================================
        const int n_busy = __popcll(__ballot(mode == M_TREE || mode == M_TWIG));
        const int n_world = __popcll(__ballot(mode == M_WORLD));
        const int n_hit = __popcll(__ballot(mode == M_HIT));
        const bool run_world = n_world > 0 && (n_world >= SVO_VOTE_WORLD || n_busy < SVO_VOTE_BUSY ||
                                               __ballot(mode == M_WORLD && creepn <= -SVO_CREEP_SERIOUS) != 0ull);
        const bool run_hit = n_hit > 0 && (n_hit >= SVO_VOTE_HIT || n_busy < SVO_VOTE_BUSY);

        if (SVO_UNLIKELY(mode != M_DONE && mode != M_HIT && ++guard > STEP_GUARD)) {     // runaway ray: give up, flag it
            if (outk < 0) store_flags(A.out, outk & 0x7FFFFFFF, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG | (A.normal_mode == SVO_NORMAL_FACE ? (uint32_t)SVO_FACE_NORMAL : 0u));
            else store_miss(A.out, outk, SVO_ERR_FLAG);
            mode = M_DONE;
        }

        // ---- chunk step: src/Traverse.cpp:142-156 -------------------------------------------
#ifdef SVO_STACK_TIMING
        n_world_runs += run_world; n_hit_runs += run_hit;
#endif
        if (run_world && mode == M_WORLD) {
            if (cw < 0) {                               // left its chunk in the step: t += escape(chunk box) + EPS, src/Traverse.cpp:164-168
                cw &= ~CW_ESCAPE_PENDING;
                if constexpr (GLSL) tw += guarded(escape(O, g, clo, clo + csize), eps) + eps;       // (the guard's threshold IS the march's EPS, Chunkmarch.glsl:113)
                else tw += escape(O, g, clo, clo + csize) + eps;
            }
            bool miss = cw >= A.cap_chunk;
            if (!miss) {
                cw++;
                const V3 p = alpha + beta * tw;
                miss = !inside(p, wlo, whi);
                if (!miss) {
                    const int ci_new = chunk_index_pow2(A, p);
                    // The descent cache (stk column, pux/puy/puz, valid) is keyed by chunk and cell, not by ray: a ray that
                    // enters the chunk this lane marched last - a shadow ray leaving its primary hit, the next pixel of the
                    // tile, a ray pinned on a chunk face - restarts below the deepest common level instead of at the root.
                    // Another chunk: nothing of the cache holds.  The cached cell goes too: chunks may differ in depth
                    // (src/Octree.h:56-76: an Ocroot carries its own), and a cell of a deeper chunk would make the next
                    // descent compute a common prefix above this chunk's top level (keep < 0).
                    if (ci_new != ci) { valid = 0; pux = 0; puy = 0; puz = 0; }
                    ci = ci_new;
                    uint32_t ch_levels, ch_wide, ch_twig;
                    if (chunks_in_lds) {
                        clo = mk(__uint_as_float(chunk_tab[0][ci]), __uint_as_float(chunk_tab[1][ci]), __uint_as_float(chunk_tab[2][ci]));
                        ch_levels = chunk_tab[3][ci]; ch_wide = chunk_tab[4][ci]; ch_twig = chunk_tab[5][ci];
                    } else {
                        const DevWide ch = A.wchunks[ci];
                        clo = ld3(ch.bmin);
                        ch_levels = ch.levels; ch_wide = ch.wide_off; ch_twig = (uint32_t)ch.twig_off;
                    }
                    bool contained = inside(p, clo, clo + csize);
                    // src/Traverse.cpp:154-155: a position outside the box of the chunk found for it ends the ray.  The shader has no
                    // such check (shaders/Chunkmarch.glsl:297-330): its treemarch fails at once, and rootmarch steps on out of that box
                    if (GLSL && !contained) tw += guarded(escape(p, g, clo, clo + csize), eps) + eps;     // (stays M_WORLD; cw counts it)
                    else miss = !contained;
                    if (contained) {                    // treemarch(p, beta, chunk): a = p, t = 0 (src/Traverse.cpp:158,78)
                        if constexpr (BIG) wide_b = (unsigned long long)(size_t)A.wide + ((unsigned long long)ch_wide << 8);
                        else wide_b = ch_wide << 8;                  // 64 entries of 4 bytes per wide node
                        twig_off = ch_twig;
                        levels = (int)ch_levels;
                        O = p; t = 0.0f; cnt = A.cap_tree;
                        Blo = clo;
                        res = csize * __uint_as_float((uint32_t)(127 - levels) << 23);     // csize / 2^levels, exact
#ifndef SVO_CXX_STEP
                        bsize = csize; res_tree = res; nw_chunk = levels ? (levels + 1) >> 1 : 1;
#endif
                        mode = M_TREE;
                    }
                }
            }
            if (miss) {
                if (outk >= 0) store_miss(A.out, outk, 0);
                if (want_cost) note_tile_cost(outk, guard);
                mode = M_DONE;                          // shadow miss: record already says "traced, lit"
            }
        }

        // ---- one step of the current level: tree (src/Traverse.cpp:79-111) or brick (:54-70) -----
        //      First decide what the step does (locate the cell, read its node / mask bit), then apply exactly one of
        //      the outcomes below as a flat sequence of predicated updates of the lane state.
        //      The step repeats at once, without the refill / vote / block checks around it, while every live lane of the
        //      wave is marching and nothing else can be due: that is the state of the waves that carry a launch's longest
        //      rays after the tile cursors ran dry, alone on their SIMD and bound by their own instruction stream.
        int pass = 0;
        // in the bulk: SVO_STEP_EXTRA more steps, decided once per pass of the outer loop (n_busy: before the chunk step)
        const int fixed_steps = (n_busy >= SVO_STEP_LANES && __ballot(creepn > 0 || creepn <= -4 * SVO_CREEP_SERIOUS) == 0ull) ? SVO_STEP_EXTRA : 0;
        for (;;) {
#ifdef SVO_STACK_TIMING
        n_lane_twig += mode == M_TWIG ? (pass == 0 ? 1u + (unsigned)fixed_steps : (unsigned)SVO_DRAIN_STEPS) : 0u;        // ... of them started inside a brick
        n_lane_busy += (mode == M_TREE || mode == M_TWIG) ? (pass == 0 ? 1u + (unsigned)fixed_steps : (unsigned)SVO_DRAIN_STEPS) : 0u;     // steps of the coming statement this lane starts as a marching lane (an upper bound on its own steps)
        { const int nm = __popcll(__ballot(mode == M_TREE || mode == M_TWIG)); n_wsteps += nm > 0; n_lsteps += nm; if (more) { n_wsteps_b += nm > 0; n_lsteps_b += nm; n_hit_wait += __popcll(__ballot(mode == M_HIT)); n_dead_wait += __popcll(__ballot(mode == M_DONE)); n_world_wait += __popcll(__ballot(mode == M_WORLD)); n_twig_b += __popcll(__ballot(mode == M_TWIG)); } }
#endif
#ifndef SVO_CXX_STEP
        // (step_asm.hip.h) the steps of this pass in one statement: 1 + fixed_steps at first, single steps while the inner repeat lasts
        const int nsteps = pass == 0 ? 1 + fixed_steps : SVO_DRAIN_STEPS;
        const int sure_miss = SVO_SURE_MISS_WHEN;      // (step_asm_body.inc: bricks whose march is bound to miss are not entered)
#ifdef SVO_STACK_TIMING
#define SVO_STEP_STATS_ARG , step_stats
#else
#define SVO_STEP_STATS_ARG
#endif
        if constexpr (BIG && GLSL)
            march_steps_asm_big_glsl(mode, O, Blo, bsize, res, t, cnt, tt_saved, t_miss, it_saved, tw, cw, pux, puy, puz, valid, plev, bmask, creepn,
                                     beta, g, clo, alpha, levels, nw_chunk, res_tree, wide_b, twig_off, lds_lane, SU, nsteps | (sure_miss << 16) SVO_STEP_STATS_ARG);
        else if constexpr (GLSL)
            march_steps_asm_glsl(mode, O, Blo, bsize, res, t, cnt, tt_saved, t_miss, it_saved, tw, cw, pux, puy, puz, valid, plev, bmask, creepn,
                                 beta, g, clo, alpha, levels, nw_chunk, res_tree, wide_b, twig_off, lds_lane, SU, nsteps | (sure_miss << 16) SVO_STEP_STATS_ARG);
        else if constexpr (BIG)
            march_steps_asm_big(mode, O, Blo, bsize, res, t, cnt, tt_saved, t_miss, it_saved, tw, cw, pux, puy, puz, valid, plev, bmask, creepn,
                                beta, g, clo, alpha, levels, nw_chunk, res_tree, wide_b, twig_off, lds_lane, SU, nsteps | (sure_miss << 16) SVO_STEP_STATS_ARG);
        else
            march_steps_asm(mode, O, Blo, bsize, res, t, cnt, tt_saved, t_miss, it_saved, tw, cw, pux, puy, puz, valid, plev, bmask, creepn,
                            beta, g, clo, alpha, levels, nw_chunk, res_tree, wide_b, twig_off, lds_lane, SU, nsteps | (sure_miss << 16) SVO_STEP_STATS_ARG);
#undef SVO_STEP_STATS_ARG
        pass += nsteps - 1;
#else
        if (mode == M_TREE || mode == M_TWIG) {
            enum : int { S_LEAVE = 0, S_ADVANCE = 1, S_ENTER = 2, S_HIT_LEAF = 3, S_HIT_CELL = 4, S_BAD = 5 };
            const bool twig = mode == M_TWIG;
            const float Bsize = twig ? res * 4.0f : csize, inv_res = recip_pow2(res);
            const int crept = creepn < 0 ? -creepn : creepn;
            creepn = -crept;                                        // disarmed unless this step advances (see the creep block)
            bool leave = cnt <= 0;                                  // the level's step cap (src/Traverse.cpp:54,79)
            cnt -= 1;                                               // (a lane that leaves restores or resets cnt below)
            const V3 p = O + beta * t;
            leave |= !inside(p, Blo, Blo + Bsize);
            // lattice coordinates of p inside the box.  Brick: truncation, as the reference (:58).  Tree: the number
            // of cell boundaries <= p; truncation gives exactly that unless the quotient is integral (p on a lattice
            // plane, or rounded onto one), which the rare branch below settles with the reference's own comparison.
            const float fx = (p.x - Blo.x) * inv_res, fy = (p.y - Blo.y) * inv_res, fz = (p.z - Blo.z) * inv_res;
            int ux = (int)fx, uy = (int)fy, uz = (int)fz;
#ifdef SVO_STACK_TIMING
            n_fix += __ballot(!twig && ((fx == (float)ux) | (fy == (float)uy) | (fz == (float)uz))) != 0;
#endif
            if (SVO_UNLIKELY(!twig && ((fx == (float)ux) | (fy == (float)uy) | (fz == (float)uz)))) {
                const int nmax = (1 << levels) - 1;
                ux = ux > nmax ? nmax : ux; uy = uy > nmax ? nmax : uy; uz = uz > nmax ? nmax : uz;
                ux -= (Blo.x + (float)ux * res > p.x) ? 1 : 0;
                uy -= (Blo.y + (float)uy * res > p.y) ? 1 : 0;
                uz -= (Blo.z + (float)uz * res > p.z) ? 1 : 0;
            }
            if (twig) leave |= (ux > 3) | (uy > 3) | (uz > 3);      // isInsideCube(off, 0, 3), :59 (off >= 0 always: p >= Blo)

            int what = S_LEAVE;
            int low = 0;                                            // the located cell spans (low+1) lattice steps
            uint32_t payload = 0;                                   // node word (tree) / cell index (brick)
            if (SVO_LIKELY(!leave)) {
                if (!twig) {
                    // descend through the chunk's wide tree (wide_tree.hip.h: two reference levels per node) from the
                    // deepest cached wide level whose node is unchanged: wide level k is selected by the coordinate bits
                    // above 2 (nw - k), so it survives while the highest differing bit lies below that
                    const uint32_t diff = (uint32_t)((ux ^ pux) | (uy ^ puy) | (uz ^ puz));
                    const int nw = levels ? (levels + 1) >> 1 : 1;
                    const int hb = 32 - __clz((int)(diff | 1u));    // (diff == 0 reads as "bit 0 differs": keep = nw - 1 >= valid either way)
                    const int keep = nw - ((hb + 1) >> 1);
                    int k;                                          // min(keep, valid), never below the top level (one v_med3_i32)
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(k) : "v"(keep), "v"(valid));
                    uint32_t wnode = stk[k][lane];                  // (row 0 holds node 0: no branch for k == 0)
                    int sh = 2 * (nw - 1 - k);                      // the two coordinate bits that select the entry
                    uint32_t word = ld_wide((wnode << 6) + wide_slot(ux, uy, uz, sh));
                    while (is_branch(word)) {
                        wnode = word & WIDE_PAYLOAD;
                        ++k;
                        stk[k][lane] = wnode;
                        sh -= 2;
                        word = ld_wide((wnode << 6) + wide_slot(ux, uy, uz, sh));
                    }
                    valid = k; pux = ux; puy = uy; puz = uz;
                    plev = (int)((word >> 25) & 31u);               // the reference node's level: it spans 2^(levels - level) cells
                    low = (1 << (levels - plev)) - 1;
                    const uint32_t type = node_type(word);
                    what = type == EMPTY ? S_ADVANCE : type == LEAF ? S_HIT_LEAF : S_ENTER;
                    payload = word;
                } else {
                    payload = (uint32_t)(uz * 16 + uy * 4 + ux);
                    what = ((bmask >> payload) & 1ull) ? S_HIT_CELL : S_ADVANCE;
                }
            }

            // ---- the one escape evaluation of the step, out of the located cell from p (src/Traverse.cpp:89,104-105,67):
            //      an EMPTY node or an empty brick cell advances by it; a TWIG node about to be entered remembers where the
            //      tree level goes on if the brick march misses - t + (escape(p, node box) + EPS), :104-105, the same
            //      expression with the same operands - so that leaving the brick later costs no evaluation of its own.
            //      A lane that leaves its level evaluates nothing here: out of a brick it resumes the tree level at the
            //      remembered parameter; out of the chunk (:164-168) it only raises the "escape pending" bit of cw, and the
            //      chunk step, for which it has to wait anyway, advances tw from the tree frame's origin, which stays put.
            if (what == S_ADVANCE || what == S_ENTER) {
                const V3 E_lo = mk(Blo.x + (float)(ux & ~low) * res, Blo.y + (float)(uy & ~low) * res, Blo.z + (float)(uz & ~low) * res);
                const float E_size = res * (float)(low + 1);
                const float e = (GLSL ? guarded(escape(p, g, E_lo, E_lo + E_size), eps) : escape(p, g, E_lo, E_lo + E_size)) + eps;
                if (what == S_ADVANCE) {
                    t += e;
                    creepn = e < 2.0f * eps ? crept + 1 : 0;        // pinned on a lattice plane: see the creep block
                } else {                                            // twigmarch(p, b, node box, ...): a = p, t = 0 (:99,53)
                    bmask = A.mask[twig_off + (payload & WIDE_PAYLOAD)];
                    tt_saved = t; t_miss = t + e; it_saved = cnt;
                    O = p; t = 0.0f; cnt = A.cap_twig;
                    Blo = E_lo;
                    res = E_size * 0.25f;                           // leafsize = node size / 4, exact
                    mode = M_TWIG;
                }
            }
            if (leave) {
                if (twig) {                                         // back to the tree level that entered the brick
                    t = t_miss;
                    cnt = it_saved;
                    O = alpha + beta * tw;                          // the chunk march's p (src/Traverse.cpp:144,158)
                    Blo = clo;
                    res = csize * __uint_as_float((uint32_t)(127 - levels) << 23);
                    mode = M_TREE;
                } else {                                            // out of the chunk
                    cw |= CW_ESCAPE_PENDING;
                    mode = M_WORLD;
                }
            }
            if (what == S_HIT_LEAF) {
                tw = tw + (GLSL ? t : t - eps);                     // src/Traverse.cpp:93,160 (t - EPS); shaders/Chunkmarch.glsl:266 (t)
                cnt = (int)SVO_CELL_NONE;
                mode = M_HIT;
            }
            if (what == S_HIT_CELL) {
                float sdist = t;                                    // src/Traverse.cpp:63
                sdist += tt_saved;                                  // :101
                tw = tw + sdist;                                    // :160
                cnt = (int)payload;
                mode = M_HIT;
            }
        }
#endif

        // Loop control is wave-uniform (scalar compare and branch).  In the bulk the number of extra steps was fixed before
        // the loop: nothing is looked at between them - the checks around the step (refill, three votes, guard, creep vote,
        // hit blocks) cost a quarter of an iteration, even one ballot per step 2 %; the lanes that wait a step or two
        // longer for a refill or a vote cost less.  The per-lane runaway guard is settled after the loop, which ends
        // after 256 steps at the latest so that the outer loop's check sees a runaway ray.
        ++pass;
        if (pass <= fixed_steps) continue;
        if (more || pass >= 256) break;
        const unsigned long long marching = __ballot(mode == M_TREE || mode == M_TWIG);
        if (marching == 0ull || marching != __ballot(mode != M_DONE) || __ballot(creepn > 0 || creepn <= -4 * SVO_CREEP_SERIOUS) != 0ull) break;
        }
        guard += pass - 1;

#include "creep_block.inc"

        // ---- hits.  A shadow ray only sets a flag; a primary hit waits (M_HIT) until the wave votes to
        //      resolve: G-buffer record, then the lane becomes its own shadow ray -------------------
        if (mode == M_HIT && outk < 0) {
            store_flags(A.out, outk & 0x7FFFFFFF, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_SHADOWED | (A.normal_mode == SVO_NORMAL_FACE ? (uint32_t)SVO_FACE_NORMAL : 0u));
            if (want_cost) note_tile_cost(outk, guard);
            mode = M_DONE;
        }
        if (run_hit && mode == M_HIT) {
            // which voxel: the node comes from the descent cache; the frame still describes the level that hit
            const int nw = levels ? (levels + 1) >> 1 : 1;
            const uint32_t wn = valid > 0 ? stk[valid][lane] : 0u, slot = wide_slot(pux, puy, puz, 2 * (nw - 1 - valid));
            const uint32_t word = ld_wide((wn << 6) + slot);        // the terminal entry again (material / brick index) ...
            // ... and the reference node it stands for (svo_hit.node): the entry's wide node keeps the index of the child block it
            // expands and of its eight grandchild blocks (wide_tree.hip.h: wbase); level 0 is the chunk's root, node 0
            uint32_t node = 0u;
            if (plev != 0) {
                const uint32_t ci = ((slot >> 1) & 1u) | ((slot >> 2) & 2u) | ((slot >> 3) & 4u);
                const uint32_t gi = (slot & 1u) | ((slot >> 1) & 2u) | ((slot >> 2) & 4u);
                size_t wnode_in_pool;                               // the wide node's index in the pool: the chunk's top wide node + wn
                if constexpr (BIG) wnode_in_pool = (((size_t)wide_b - (size_t)A.wide) >> 8) + wn;
                else wnode_in_pool = (wide_b >> 8) + wn;
                const uint32_t *wb = A.wbase + wnode_in_pool * WIDE_BASE_WORDS;
                const int level_child = 2 * valid + 1 - (2 * nw - levels);
                node = plev == level_child ? wb[0] + ci : wb[1 + ci] + gi;
            }
            const uint32_t hitc = (uint32_t)cnt;                    // which brick cell (or SVO_CELL_NONE: a LEAF)
            V3 vlo;
            float vsize;
            uint32_t material;
            if (hitc == SVO_CELL_NONE) {                            // LEAF node: frame = tree level (Blo = clo, res = cell)
                const int low = (1 << (levels - plev)) - 1;
                vlo = mk(Blo.x + (float)(pux & ~low) * res, Blo.y + (float)(puy & ~low) * res, Blo.z + (float)(puz & ~low) * res);
                vsize = res * (float)(low + 1);
                material = word & 0xFFFFu;
            } else {                                                // brick cell: frame = brick (Blo = node box, res = voxel)
                vlo = mk(Blo.x + (float)(hitc & 3u) * res, Blo.y + (float)((hitc >> 2) & 3u) * res, Blo.z + (float)(hitc >> 4) * res);
                vsize = res;
                // most bricks hold one material (grow() fills a brick with its node's): it is kept per brick next to the masks, 64
                // bricks to a cache line; only a brick of several materials (0xFFFF) is read itself, 128 B for one cell
                const unsigned long long brick = (unsigned long long)twig_off + (word & WIDE_PAYLOAD);
                material = A.bmat[brick];
                if (material == 0xFFFFu) material = A.twig[brick * TWIG_WORDS + hitc];
            }
            const V3 point = alpha + beta * (tw - eps);
            const bool face = A.normal_mode == SVO_NORMAL_FACE;
            const V3 n = face ? face_normal(point, vlo, vlo + vsize, beta) : cube_normal_pow2(point, vlo, vsize, eps);
            const uint32_t flags = SVO_HIT_FLAG | (A.shadow ? SVO_SHADOW_TRACED : 0u) | (face ? (uint32_t)SVO_FACE_NORMAL : 0u);
            store_hit(A.out, outk, tw, n, material, flags, (uint32_t)ci, node, hitc);
            if (want_cost) note_tile_cost(outk, guard);
            mode = M_DONE;
            if (A.shadow) {                             // the lane becomes its own shadow ray
                alpha = point; beta = sdir; g = sg;
                outk |= (int)0x80000000;
                tw = 0.0f; cw = 0; guard = 0; creepn = 0;
                bool hit = true;
                if (!inside(alpha, wlo, whi)) tw = (GLSL ? enter_glsl(alpha, g, wlo, whi, hit) : enter(alpha, beta, wlo, whi, hit)) + eps;
                mode = hit ? M_WORLD : M_DONE;          // a shadow ray that misses the world box stays "lit"
                rays_marched++;
            }
        }
    }

    unsigned total = rays_marched;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if (lane == 0 && total) atomicAdd(&A.work[1], (unsigned long long)total);
#ifdef SVO_STACK_TIMING
    (void)n_creep_steps;
    unsigned busiest = n_lane_busy;
    for (int off = 32; off > 0; off >>= 1) busiest = max(busiest, (unsigned)__shfl_xor((int)busiest, off, 64));
    unsigned busiest_twig = n_lane_busy == busiest ? n_lane_twig : 0u;          // the busiest lane's share inside bricks
    for (int off = 32; off > 0; off >>= 1) busiest_twig = max(busiest_twig, (unsigned)__shfl_xor((int)busiest_twig, off, 64));
    if (lane == 0 && A.counters) {       // diagnostic build only: per-wave [start, end] in 10 ns ticks, iterations, rays
        uint4 c; c.x = (uint32_t)t_begin; c.y = (uint32_t)__builtin_amdgcn_s_memrealtime(); c.z = n_iters; c.w = total;
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x] = c;
        uint4 e; e.x = n_world_runs | (n_hit_runs << 16); e.y = n_refill | (n_tilegen << 12) | (n_fix << 20); e.z = n_tree_lanes; e.w = n_twig_lanes | (n_world_lanes << 20);
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x + 1] = e;
        uint4 f; f.x = n_creep_runs | (busiest_twig << 16); f.y = busiest; f.z = n_creep_rounds; f.w = n_dbg;
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x + 2] = f;
        uint4 h; h.x = n_wsteps; h.y = n_lsteps; h.z = n_hit_wait; h.w = n_dead_wait;
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x + 3] = h;
        uint4 h2; h2.x = n_wsteps_b; h2.y = n_lsteps_b; h2.z = n_world_wait; h2.w = n_twig_b;
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x + 4] = h2;
        uint4 h3; h3.x = step_stats.steps; h3.y = step_stats.lanes; h3.z = step_stats.stalls; h3.w = step_stats.chased;
#ifdef SVO_SURE_STAT_WORD  // (measurement: the last word counts the brick entries skipped as sure misses instead)
        h3.w = step_stats.sure;
#endif
        reinterpret_cast<uint4 *>(A.counters)[6 * blockIdx.x + 5] = h3;
    }
#endif
}

} // namespace svo
