// kernel_pool.hip.h — the OVER-SUBSCRIBED variant of the stack kernel (round 4, experimental; SVO_KERNEL_POOL).
//
// What it is for (DESIGN.md §6, round 4's counters): k_trace_stack is bound by instruction issue with 44.8 of 64 lanes marching
// in its average step, and 22 % of its instructions are the blocks around the step (chunk step, hit resolve, refill), which run
// for two to sixteen lanes at a time.  Serving waiting lanes more often costs what it wins.  Here the waiting rays leave the
// lanes instead:
//
//   * A workgroup is FOUR waves that share three ray queues in LDS: QW (rays that need a chunk step: fresh rays, rays that left
//     their chunk, shadow rays just born), QR (rays whose chunk step is done: ready to march a tree from t = 0) and QH (primary
//     hits that need their G-buffer record).  A parked ray is 14 - 16 words; the descent cache stays with the lane.
//   * After every statement a wave EXCHANGES under the workgroup's queue lock: lanes whose ray left its chunk or hit push it to
//     QW / QH and are free at once; free lanes pop ready rays from QR.  Then - one service per pass at most - a wave takes up to 64
//     entries of QW or QH (or a fresh tile) and serves them with all its lanes at once, out of temporaries, while its own marching
//     rays rest in their registers: the chunk step of 64 rays costs what it cost for five, the hit resolve likewise.
//   * The march itself - the asm statement, the creep block - is the stack kernel's, untouched; so are the results: a ray's state
//     is carried through the queues bit for bit, and rays are independent.
//   * Nothing waits for anybody: a wave that finds a queue full serves its own lanes in place (the stack kernel's blocks), a wave
//     without work and without live lanes ends; the lock is held for one exchange (~60 LDS instructions) at a time.
//
// Limits of this first version (the launcher falls back to k_trace_stack otherwise): CPU semantics, chunk table in LDS (<= 64
// chunks), at most 10 branch levels (the four descent columns), no tile-cost recording / caller's tile order, the default
// addressing (wide pool < 4 GiB).
#pragma once
#include "kernel_stack.hip.h"

namespace svo {

constexpr int POOL_WAVES = 4;
constexpr int QW_CAP = 128, QR_CAP = 128, QH_CAP = 64;          // entries (powers of two: rings)
constexpr int QW_WORDS = 14, QR_WORDS = 15, QH_WORDS = 16;
enum : int { QC_LOCK = 0, QC_W_HEAD, QC_W_COUNT, QC_W_RESERVED, QC_W_BUSY, QC_R_HEAD, QC_R_COUNT, QC_R_RESERVED, QC_H_HEAD, QC_H_COUNT, QC_H_BUSY, QC_WORDS };
enum : int { SERVE_NONE = 0, SERVE_WORLD = 1, SERVE_HIT = 2, SERVE_TILE = 3 };
#ifndef SVO_POOL_SERVE_BUSY
#define SVO_POOL_SERVE_BUSY 40          // fewer marching lanes than this: serve whatever waits, however little
#endif
#ifndef SVO_POOL_WAVES_PER_SIMD
#define SVO_POOL_WAVES_PER_SIMD 5       // register budget: 96 VGPRs (the exchange and the services need room beside the marching state; at 80 they spill into the pass loop)
#endif
// the pool must run ahead of the lanes: a tile is fetched while fewer rays than this wait in QW + QR (a free lane finds a ready ray
// only if some are kept in stock), and the chunk steps are served early when QR runs low
#ifndef SVO_POOL_TILE_BELOW
#define SVO_POOL_TILE_BELOW 128
#endif
#ifndef SVO_POOL_R_LOW
#define SVO_POOL_R_LOW 32
#endif
#ifndef SVO_POOL_STEPS
#define SVO_POOL_STEPS 4                // extra steps per statement in the bulk (as SVO_STEP_EXTRA)
#endif

typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef const char __attribute__((address_space(4))) *kernarg_ptr;
struct PoolShared { lds_u32 *qw, *qr, *qh, *qc, *ctab; };      // the workgroup's queues, their control words and the chunk table, in LDS (ds_ instructions)

__device__ __forceinline__ TraceArgs args_from(kernarg_ptr kp)
{
    // (an out-of-line function receives its arguments in VGPRs: the kernarg pointer is wave-uniform, and scalar loads want it in SGPRs)
    const unsigned long long v = (unsigned long long)(size_t)kp;
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    kp = (kernarg_ptr)(size_t)(((unsigned long long)hi << 32) | lo);
    asm volatile("" : "+s"(kp));
    TraceArgs T;
    __builtin_memcpy(&T, kp, sizeof T);              // only the fields the caller uses survive (scalar loads)
    return T;
}
__device__ __forceinline__ void pool_lock(lds_u32 *qc, int lane)
{
    if (lane == 0) while (__hip_atomic_exchange(&qc[QC_LOCK], 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void pool_unlock(lds_u32 *qc, int lane)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(&qc[QC_LOCK], 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int pool_rd(lds_u32 *qc, int i) { return __builtin_amdgcn_readfirstlane((int)*(volatile lds_u32 *)(qc + i)); }
__device__ __forceinline__ void pool_wr(lds_u32 *qc, int i, int v) { *(volatile lds_u32 *)(qc + i) = (uint32_t)v; }
__device__ __forceinline__ int pool_rank(unsigned long long m) { return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); }

// chunkmarch's loop head for one ray (src/Traverse.cpp:142-156): the chunk entered, or -1 = the ray ends here (cap, left the
// world, not contained); cw is counted up as the reference's loop counter
__device__ __forceinline__ int pool_chunk_step(const TraceArgs &T, lds_u32 *ctab, V3 a, V3 b, float tw, int &cw)
{
    if (cw >= T.cap_chunk) return -1;
    cw++;
    const V3 p = a + b * tw;
    if (!inside(p, ld3(T.worldmin), ld3(T.worldmax))) return -1;
    const int c = chunk_index_pow2(T, p);
    const V3 lo = mk(__uint_as_float(ctab[c]), __uint_as_float(ctab[64 + c]), __uint_as_float(ctab[128 + c]));
    if (!inside(p, lo, lo + T.chunksize)) return -1;
    return c;
}
// the G-buffer record of a primary hit (kernel_stack.hip.h's hit block) from what the hit lane knew; returns the sample point
__device__ __forceinline__ V3 pool_resolve_hit(const TraceArgs &T, lds_u32 *ctab, int k, V3 a, V3 b, float tw, int c, V3 vlo, float vsize, uint32_t eidx, uint32_t misc)
{
    const uint32_t hitc = misc & 0xFFu;
    const uint32_t wide_off = ctab[256 + c], tof = ctab[320 + c];
    const uint32_t word = ld_node(T.wide, (wide_off << 8) + (eidx << 2));
    uint32_t node = 0u;
    if (misc & 0x200u) {                                    // not the chunk's root
        const uint32_t slot = eidx & 63u;
        const uint32_t cidx = ((slot >> 1) & 1u) | ((slot >> 2) & 2u) | ((slot >> 3) & 4u);
        const uint32_t gidx = (slot & 1u) | ((slot >> 1) & 2u) | ((slot >> 2) & 4u);
        const uint32_t *wb = T.wbase + ((size_t)wide_off + (eidx >> 6)) * WIDE_BASE_WORDS;
        node = (misc & 0x100u) ? wb[0] + cidx : wb[1 + cidx] + gidx;
    }
    uint32_t material;
    if (hitc == SVO_CELL_NONE) material = word & 0xFFFFu;
    else {
        const unsigned long long brick = (unsigned long long)tof + (word & WIDE_PAYLOAD);
        material = T.bmat[brick];
        if (material == 0xFFFFu) material = T.twig[brick * TWIG_WORDS + hitc];
    }
    const V3 point = a + b * (tw - T.eps);
    const bool face = T.normal_mode == SVO_NORMAL_FACE;
    const V3 n = face ? face_normal(point, vlo, vlo + vsize, b) : cube_normal_pow2(point, vlo, vsize, T.eps);
    const uint32_t flags = SVO_HIT_FLAG | (T.shadow ? SVO_SHADOW_TRACED : 0u) | (face ? (uint32_t)SVO_FACE_NORMAL : 0u);
    store_hit(T.out, k, tw, n, material, flags, (uint32_t)c, node, hitc);
    return point;
}

// ---- the lanes' own rays served in place when a queue has no room (rare): out of line, so that the marching state's registers are
//      only disturbed when it happens
struct InplaceWorld { int chunk, cw; };
__device__ __noinline__ InplaceWorld pool_inplace_world(kernarg_ptr kp, lds_u32 *ctab, int outk, V3 a, V3 b, float tw, int cw)
{
    const TraceArgs T = args_from(kp);
    InplaceWorld r;
    r.chunk = pool_chunk_step(T, ctab, a, b, tw, cw);
    r.cw = cw;
    if (r.chunk < 0 && outk >= 0) store_miss(T.out, outk, 0);
    return r;
}
struct InplaceHit { V3 point; float tw; int enters; };
__device__ __noinline__ InplaceHit pool_inplace_hit(kernarg_ptr kp, lds_u32 *ctab, int outk, V3 a, V3 b, float tw, int c, V3 vlo, float vsize, uint32_t eidx, uint32_t misc)
{
    const TraceArgs T = args_from(kp);
    InplaceHit r;
    r.point = pool_resolve_hit(T, ctab, outk, a, b, tw, c, vlo, vsize, eidx, misc);
    r.tw = 0.0f; r.enters = 0;
    if (T.shadow) {
        bool hit = true;
        const V3 wlo = ld3(T.worldmin), whi = ld3(T.worldmax);
        if (!inside(r.point, wlo, whi)) r.tw = enter(r.point, ld3(T.sdir), wlo, whi, hit) + T.eps;
        r.enters = hit ? 1 : 0;
    }
    return r;
}

// ---- a service: up to 64 entries of QW or QH - read where they lie: the claim keeps them in their queue - or the next tile's 64
//      rays, worked on by all lanes at once, results handed to QR / QW under the lock, the served entries freed.  Out of line: the
//      calling wave's marching rays rest in their (callee-saved) registers meanwhile.  Returns the tile cursor's state and the rays started.
struct TileCursor { int region, regions_left, more; unsigned started; };
__device__ __noinline__ TileCursor pool_service(kernarg_ptr kp, PoolShared P, int serve, int serve_n, int serve_head, int lane, TileCursor tc)
{
    const TraceArgs T = args_from(kp);
    const V3 wlo = ld3(T.worldmin), whi = ld3(T.worldmax);
    const float eps = T.eps;
    tc.started = 0u;
    bool have = false, to_ready = false;                // this lane holds a ray to push when the service is done (to QR / to QW)
    int s_outk = 0; V3 s_alpha = mk(0, 0, 0), s_beta = mk(0, 0, 1), s_g = mk(0, 0, 0); float s_tw = 0.0f; int s_cw = 0; uint32_t s_guard = 0; int s_creepn = 0, s_ci = 0;
    if (serve == SERVE_WORLD) {
        if (lane < serve_n) {
            lds_u32 *q = P.qw + ((serve_head + lane) & (QW_CAP - 1));
            s_outk = (int)q[0];
            s_alpha = mk(__uint_as_float(q[1 * QW_CAP]), __uint_as_float(q[2 * QW_CAP]), __uint_as_float(q[3 * QW_CAP]));
            s_beta = mk(__uint_as_float(q[4 * QW_CAP]), __uint_as_float(q[5 * QW_CAP]), __uint_as_float(q[6 * QW_CAP]));
            s_g = mk(__uint_as_float(q[7 * QW_CAP]), __uint_as_float(q[8 * QW_CAP]), __uint_as_float(q[9 * QW_CAP]));
            s_tw = __uint_as_float(q[10 * QW_CAP]); s_cw = (int)q[11 * QW_CAP]; s_guard = q[12 * QW_CAP]; s_creepn = (int)q[13 * QW_CAP];
            s_ci = pool_chunk_step(T, P.ctab, s_alpha, s_beta, s_tw, s_cw);
            if (s_ci >= 0) { have = true; to_ready = true; }
            else if (s_outk >= 0) store_miss(T.out, s_outk, 0);          // (a shadow ray that ends: its record already says "traced, lit")
        }
    } else if (serve == SERVE_HIT) {
        if (lane < serve_n) {
            lds_u32 *q = P.qh + ((serve_head + lane) & (QH_CAP - 1));
            s_outk = (int)q[0];
            s_alpha = mk(__uint_as_float(q[1 * QH_CAP]), __uint_as_float(q[2 * QH_CAP]), __uint_as_float(q[3 * QH_CAP]));
            s_beta = mk(__uint_as_float(q[4 * QH_CAP]), __uint_as_float(q[5 * QH_CAP]), __uint_as_float(q[6 * QH_CAP]));
            const V3 vlo = mk(__uint_as_float(q[10 * QH_CAP]), __uint_as_float(q[11 * QH_CAP]), __uint_as_float(q[12 * QH_CAP]));
            const V3 point = pool_resolve_hit(T, P.ctab, s_outk, s_alpha, s_beta, __uint_as_float(q[7 * QH_CAP]), (int)q[9 * QH_CAP], vlo,
                                              __uint_as_float(q[13 * QH_CAP]), q[14 * QH_CAP], q[15 * QH_CAP]);
            if (T.shadow) {                             // the shadow ray: a fresh ray for the chunk step
                s_alpha = point; s_beta = ld3(T.sdir); s_g = recip(s_beta);
                s_outk |= (int)0x80000000;
                s_tw = 0.0f; s_cw = 0; s_guard = 0; s_creepn = 0;
                bool hit = true;
                if (!inside(s_alpha, wlo, whi)) s_tw = enter(s_alpha, s_beta, wlo, whi, hit) + eps;
                have = hit;
                tc.started++;
            }
        }
    } else {                                            // SERVE_TILE: the next tile's 64 rays (kernel_stack.hip.h's tile generation)
        const int tr_cols = T.tiles_per_row;
        const int tr_rows = T.ntiles / (tr_cols > 0 ? tr_cols : 1);
        const bool by_cols = tr_cols >= TILE_REGIONS;
        const int reg_q = (by_cols ? tr_cols : tr_rows) / TILE_REGIONS, reg_rem = (by_cols ? tr_cols : tr_rows) % TILE_REGIONS;
        int t32 = -1, tcol = 0, trow = 0, tframe = 0;
        while (tc.regions_left > 0) {
            const int span = reg_q + (tc.region < reg_rem ? 1 : 0);
            const int first = tc.region * reg_q + (tc.region < reg_rem ? tc.region : reg_rem);
            const int count = span * (by_cols ? tr_rows : tr_cols);
            unsigned long long tix = 0;
            if (lane == 0) tix = atomicAdd(&T.work[WORK_CURSOR0 + tc.region], 1ull);
            int tt = __builtin_amdgcn_readfirstlane((int)tix);
            if (tt < count * T.nframes) {
                if (T.nframes > 1) { tframe = tt / count; tt -= tframe * count; }
                if (by_cols) { trow = tt / span; tcol = first + (tt - trow * span); }
                else { trow = first + tt / tr_cols; tcol = tt % tr_cols; }
                t32 = trow * tr_cols + tcol;
                break;
            }
            tc.region = (tc.region + 1) & (TILE_REGIONS - 1);
            --tc.regions_left;
        }
        if (t32 < 0) tc.more = 0;
        else {
            const int id = t32 * 64 + lane;
            bool ok;
            int k = -1;
            V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
            if (T.from_camera) {
                const int lx = tcol * TILE_W + (lane & (TILE_W - 1));
                const int ly = trow * TILE_H + lane / TILE_W;
                ok = (lx < T.w) & (ly < T.h);
                k = (tframe * T.h + ly) * T.w + lx;
                int px = 0, py = 0;
                if (ok) local_to_pixel(T, lx, ly, px, py);
                if (ok && (py >= T.imgh || px >= T.imgw)) { store_miss(T.out, k, 0); ok = false; }
                if (ok) {
                    FrameCam cam;
                    __builtin_memcpy(&cam, kp + __builtin_offsetof(TraceArgs, cams) + (size_t)tframe * sizeof(FrameCam), sizeof cam);
                    camera_ray(cam, T.imgw, T.imgh, px, py, o, d);
                }
            } else {
                ok = id < T.n;
                k = id;
                if (ok) { o = ld3(T.origins + 3 * (long long)id); d = ld3(T.dirs + 3 * (long long)id); }
            }
            const V3 gg = recip(d);
            float t0 = 0.0f;
            if (ok) {
                bool hit = true;
                if (!inside(o, wlo, whi)) t0 = enter(o, d, wlo, whi, hit) + eps;
                tc.started++;
                if (!hit) { store_miss(T.out, k, 0); ok = false; }
            }
            s_outk = k; s_alpha = o; s_beta = d; s_g = gg; s_tw = t0; s_cw = 0; s_guard = 0; s_creepn = 0;
            have = ok;
        }
    }
    // hand the results over (the room was reserved at the claim) and free the entries served
    lds_u32 *qc = P.qc;
    pool_lock(qc, lane);
    if (serve == SERVE_WORLD) {
        const int r_head = pool_rd(qc, QC_R_HEAD), r_count = pool_rd(qc, QC_R_COUNT), r_res = pool_rd(qc, QC_R_RESERVED);
        const unsigned long long m = __ballot(have && to_ready);
        if (have && to_ready) {
            lds_u32 *q = P.qr + ((r_head + r_count + pool_rank(m)) & (QR_CAP - 1));
            q[0] = (uint32_t)s_outk;
            q[1 * QR_CAP] = __float_as_uint(s_alpha.x); q[2 * QR_CAP] = __float_as_uint(s_alpha.y); q[3 * QR_CAP] = __float_as_uint(s_alpha.z);
            q[4 * QR_CAP] = __float_as_uint(s_beta.x); q[5 * QR_CAP] = __float_as_uint(s_beta.y); q[6 * QR_CAP] = __float_as_uint(s_beta.z);
            q[7 * QR_CAP] = __float_as_uint(s_g.x); q[8 * QR_CAP] = __float_as_uint(s_g.y); q[9 * QR_CAP] = __float_as_uint(s_g.z);
            q[10 * QR_CAP] = __float_as_uint(s_tw); q[11 * QR_CAP] = (uint32_t)s_cw; q[12 * QR_CAP] = s_guard; q[13 * QR_CAP] = (uint32_t)s_creepn; q[14 * QR_CAP] = (uint32_t)s_ci;
        }
        const int w_head = (pool_rd(qc, QC_W_HEAD) + serve_n) & (QW_CAP - 1), w_count = pool_rd(qc, QC_W_COUNT) - serve_n;
        if (lane == 0) {
            pool_wr(qc, QC_R_COUNT, r_count + __popcll(m)); pool_wr(qc, QC_R_RESERVED, r_res - serve_n);
            pool_wr(qc, QC_W_HEAD, w_head); pool_wr(qc, QC_W_COUNT, w_count); pool_wr(qc, QC_W_BUSY, 0);
        }
    } else {
        const int w_head = pool_rd(qc, QC_W_HEAD), w_count = pool_rd(qc, QC_W_COUNT), w_res = pool_rd(qc, QC_W_RESERVED);
        const unsigned long long m = __ballot(have);
        if (have) {
            lds_u32 *q = P.qw + ((w_head + w_count + pool_rank(m)) & (QW_CAP - 1));
            q[0] = (uint32_t)s_outk;
            q[1 * QW_CAP] = __float_as_uint(s_alpha.x); q[2 * QW_CAP] = __float_as_uint(s_alpha.y); q[3 * QW_CAP] = __float_as_uint(s_alpha.z);
            q[4 * QW_CAP] = __float_as_uint(s_beta.x); q[5 * QW_CAP] = __float_as_uint(s_beta.y); q[6 * QW_CAP] = __float_as_uint(s_beta.z);
            q[7 * QW_CAP] = __float_as_uint(s_g.x); q[8 * QW_CAP] = __float_as_uint(s_g.y); q[9 * QW_CAP] = __float_as_uint(s_g.z);
            q[10 * QW_CAP] = __float_as_uint(s_tw); q[11 * QW_CAP] = (uint32_t)s_cw; q[12 * QW_CAP] = s_guard; q[13 * QW_CAP] = (uint32_t)s_creepn;
        }
        if (lane == 0) { pool_wr(qc, QC_W_COUNT, w_count + __popcll(m)); pool_wr(qc, QC_W_RESERVED, w_res - serve_n); }
        if (serve == SERVE_HIT) {
            const int h_head = (pool_rd(qc, QC_H_HEAD) + serve_n) & (QH_CAP - 1), h_count = pool_rd(qc, QC_H_COUNT) - serve_n;
            if (lane == 0) { pool_wr(qc, QC_H_HEAD, h_head); pool_wr(qc, QC_H_COUNT, h_count); pool_wr(qc, QC_H_BUSY, 0); }
        }
    }
    pool_unlock(qc, lane);
    return tc;
}

template <int MAXLV, int WAVES_PER_SIMD>
__global__ __launch_bounds__(64 * POOL_WAVES, WAVES_PER_SIMD) void k_trace_pool(TraceArgs A)
{
    __shared__ uint32_t stk[POOL_WAVES][MAXLV / 2 + 1][64];     // per wave: wide node index per wide level of the lane's current path
    __shared__ uint32_t chunk_tab[6][64];
    __shared__ uint32_t qw[QW_WORDS][QW_CAP], qr[QR_WORDS][QR_CAP], qh[QH_WORDS][QH_CAP];      // field-major: lanes touch consecutive slots
    __shared__ uint32_t qcs[QC_WORDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    stk[wave][0][lane] = 0u;
    const int n_chunks = A.dimw * A.dimh * A.dimd;              // (<= 64: the launcher checks)
    if (threadIdx.x < (unsigned)n_chunks) {
        const DevWide ch = A.wchunks[threadIdx.x];
        chunk_tab[0][threadIdx.x] = __float_as_uint(ch.bmin[0]); chunk_tab[1][threadIdx.x] = __float_as_uint(ch.bmin[1]); chunk_tab[2][threadIdx.x] = __float_as_uint(ch.bmin[2]);
        chunk_tab[3][threadIdx.x] = ch.levels; chunk_tab[4][threadIdx.x] = ch.wide_off; chunk_tab[5][threadIdx.x] = (uint32_t)ch.twig_off;
    }
    if (threadIdx.x < QC_WORDS) qcs[threadIdx.x] = 0u;
    __syncthreads();                                            // the only workgroup barrier: from here on the waves run on their own
    PoolShared P;
    P.qw = (lds_u32 *)&qw[0][0]; P.qr = (lds_u32 *)&qr[0][0]; P.qh = (lds_u32 *)&qh[0][0]; P.qc = (lds_u32 *)&qcs[0]; P.ctab = (lds_u32 *)&chunk_tab[0][0];
    lds_u32 *const qc = P.qc;
    const kernarg_ptr kp = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();

    const V3 wlo = ld3(A.worldmin), whi = ld3(A.worldmax);
    const V3 sdir = ld3(A.sdir);
    const V3 sg = recip(sdir);
    const float eps = A.eps;
    const bool eps_pow2 = (__float_as_uint(eps) & 0x807FFFFFu) == 0u && __float_as_uint(eps) >= 0x00800000u && __float_as_uint(eps) < 0x7F800000u;
    const float csize = A.chunksize;
    (void)wlo; (void)whi;

    // ---- wave state (uniform): the tile cursor (a wave starts in the screen region of its XCD, as in k_trace_stack)
    TileCursor tc;
    tc.region = (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & (TILE_REGIONS - 1));
    tc.regions_left = TILE_REGIONS; tc.more = 1; tc.started = 0u;
    unsigned rays_started = 0;                  // per lane, summed at exit
    bool pool_idle = false;                     // the last exchange found every queue empty and no tile left

    // ---- lane state: exactly k_trace_stack's
    int mode = M_DONE;
    int outk = 0;
    V3 alpha = mk(0, 0, 0), beta = mk(0, 0, 1), g = mk(0, 0, 0);
    float tw = 0.0f;
    int cw = 0;
    uint32_t guard = 0;
    V3 O = mk(0, 0, 0), Blo = mk(0, 0, 0);
    float res = 1.0f, t = 0.0f;
    int cnt = 0;
    float tt_saved = 0.0f, t_miss = 0.0f;
    int it_saved = 0;
    V3 clo = mk(0, 0, 0);
    uint32_t wide_b = 0, twig_off = 0;
    int levels = 0, ci = -1;
    int pux = 0, puy = 0, puz = 0, valid = 0, plev = 0;
    unsigned long long bmask = 0;
    float bsize = 0.0f, res_tree = 1.0f;
    int nw_chunk = 1;
    const uint32_t lds_lane = (uint32_t)(size_t)(__attribute__((address_space(3))) uint32_t *)&stk[wave][0][lane];
    StepUniform SU;
    SU.csize = csize; SU.eps = eps; SU.eps2 = 2.0f * eps; SU.cap_twig = A.cap_twig; SU.wide = A.wide; SU.mask = A.mask; SU.descend_shift = MAXLV <= 10 ? SVO_DESCEND_SHIFT_SHALLOW : SVO_DESCEND_SHIFT_DEEP;
    int creepn = 0;
#ifdef SVO_STACK_TIMING
    unsigned n_creep_runs = 0, n_creep_steps = 0, n_creep_rounds = 0, n_dbg = 0;       // (creep_block.inc counts into these in the timing build)
#endif

    // the lane's tree frame in chunk c (src/Traverse.cpp:158,78: a = p, t = 0), from the chunk table in LDS
    auto enter_chunk = [&](int c) {
        if (c != ci) { valid = 0; pux = 0; puy = 0; puz = 0; }         // the descent cache is keyed by chunk and cell (kernel_stack.hip.h)
        ci = c;
        clo = mk(__uint_as_float(chunk_tab[0][c]), __uint_as_float(chunk_tab[1][c]), __uint_as_float(chunk_tab[2][c]));
        levels = (int)chunk_tab[3][c];
        wide_b = chunk_tab[4][c] << 8;
        twig_off = chunk_tab[5][c];
        O = alpha + beta * tw; t = 0.0f; cnt = A.cap_tree;
        Blo = clo;
        res = csize * __uint_as_float((uint32_t)(127 - levels) << 23);
        bsize = csize; res_tree = res; nw_chunk = levels ? (levels + 1) >> 1 : 1;
        mode = M_TREE;
    };
    // what a hit lane knows about its voxel: box, the terminal entry's index in the chunk's wide tree, cell / level flags
    auto hit_voxel = [&](V3 &vlo, float &vsize, uint32_t &eidx, uint32_t &misc) {
        const int nw = levels ? (levels + 1) >> 1 : 1;
        const uint32_t wn = valid > 0 ? stk[wave][valid][lane] : 0u;
        eidx = (wn << 6) + wide_slot(pux, puy, puz, 2 * (nw - 1 - valid));
        const uint32_t hitc = (uint32_t)cnt;
        const int level_child = 2 * valid + 1 - (2 * nw - levels);
        misc = (hitc & 0xFFu) | (plev == level_child ? 0x100u : 0u) | (plev != 0 ? 0x200u : 0u);
        if (hitc == SVO_CELL_NONE) {
            const int low = (1 << (levels - plev)) - 1;
            vlo = mk(Blo.x + (float)(pux & ~low) * res, Blo.y + (float)(puy & ~low) * res, Blo.z + (float)(puz & ~low) * res);
            vsize = res * (float)(low + 1);
        } else {
            vlo = mk(Blo.x + (float)(hitc & 3u) * res, Blo.y + (float)((hitc >> 2) & 3u) * res, Blo.z + (float)(hitc >> 4) * res);
            vsize = res;
        }
    };

    uint32_t dbg_iters = 0, dbg_serves[4] = { 0, 0, 0, 0 }, dbg_spins = 0, dbg_inplace = 0, dbg_statements = 0, dbg_lanes = 0;
    const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
    for (unsigned iter = 0; iter < (1u << 24); ++iter) {        // (the bound is a fuse: no launch gets near it)
        ++dbg_iters;
        // ---- rays that end or change hands without anybody's help
        if (mode == M_HIT && outk < 0) {                        // a shadow ray hit: one flag
            store_flags(A.out, outk & 0x7FFFFFFF, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_SHADOWED | (A.normal_mode == SVO_NORMAL_FACE ? (uint32_t)SVO_FACE_NORMAL : 0u));
            mode = M_DONE;
        }
        if (SVO_UNLIKELY(mode != M_DONE && mode != M_HIT && guard > STEP_GUARD)) {     // runaway ray: give up, flag it (as k_trace_stack)
            if (outk < 0) store_flags(A.out, outk & 0x7FFFFFFF, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG | (A.normal_mode == SVO_NORMAL_FACE ? (uint32_t)SVO_FACE_NORMAL : 0u));
            else store_miss(A.out, outk, SVO_ERR_FLAG);
            mode = M_DONE;
        }
        if (mode == M_WORLD && cw < 0) {                        // left its chunk in the step: t += escape(chunk box) + EPS, src/Traverse.cpp:164-168
            cw &= ~CW_ESCAPE_PENDING;
            tw += escape(O, g, clo, clo + csize) + eps;
        }

        // ---- the exchange, under the workgroup's queue lock
        int serve = SERVE_NONE, serve_n = 0, serve_head = 0;   // a service claims the OLDEST serve_n entries of its queue; they stay where they are
        {                                                       // (one service per queue at a time: the *_BUSY flags) until the server frees them
            pool_lock(qc, lane);
            const int w_head = pool_rd(qc, QC_W_HEAD);
            int w_count = pool_rd(qc, QC_W_COUNT), w_res = pool_rd(qc, QC_W_RESERVED), w_busy = pool_rd(qc, QC_W_BUSY);
            int r_head = pool_rd(qc, QC_R_HEAD), r_count = pool_rd(qc, QC_R_COUNT), r_res = pool_rd(qc, QC_R_RESERVED);
            const int h_head = pool_rd(qc, QC_H_HEAD);
            int h_count = pool_rd(qc, QC_H_COUNT), h_busy = pool_rd(qc, QC_H_BUSY);
            // push: rays that need a chunk step
            {
                const unsigned long long m = __ballot(mode == M_WORLD);
                const int room = QW_CAP - w_count - w_res, n = min(__popcll(m), room);
                const int r = pool_rank(m);
                if (mode == M_WORLD && r < n) {
                    const int s = (w_head + w_count + r) & (QW_CAP - 1);
                    qw[0][s] = (uint32_t)outk;
                    qw[1][s] = __float_as_uint(alpha.x); qw[2][s] = __float_as_uint(alpha.y); qw[3][s] = __float_as_uint(alpha.z);
                    qw[4][s] = __float_as_uint(beta.x); qw[5][s] = __float_as_uint(beta.y); qw[6][s] = __float_as_uint(beta.z);
                    qw[7][s] = __float_as_uint(g.x); qw[8][s] = __float_as_uint(g.y); qw[9][s] = __float_as_uint(g.z);
                    qw[10][s] = __float_as_uint(tw); qw[11][s] = (uint32_t)cw; qw[12][s] = guard; qw[13][s] = (uint32_t)creepn;
                    mode = M_DONE;
                }
                w_count += n;
            }
            // push: primary hits
            {
                const unsigned long long m = __ballot(mode == M_HIT);
                if (m != 0ull) {
                    const int room = QH_CAP - h_count, n = min(__popcll(m), room);
                    const int r = pool_rank(m);
                    if (mode == M_HIT && r < n) {
                        V3 vlo; float vsize; uint32_t eidx, misc;
                        hit_voxel(vlo, vsize, eidx, misc);
                        const int s = (h_head + h_count + r) & (QH_CAP - 1);
                        qh[0][s] = (uint32_t)outk;
                        qh[1][s] = __float_as_uint(alpha.x); qh[2][s] = __float_as_uint(alpha.y); qh[3][s] = __float_as_uint(alpha.z);
                        qh[4][s] = __float_as_uint(beta.x); qh[5][s] = __float_as_uint(beta.y); qh[6][s] = __float_as_uint(beta.z);
                        qh[7][s] = __float_as_uint(tw); qh[8][s] = guard; qh[9][s] = (uint32_t)ci;
                        qh[10][s] = __float_as_uint(vlo.x); qh[11][s] = __float_as_uint(vlo.y); qh[12][s] = __float_as_uint(vlo.z);
                        qh[13][s] = __float_as_uint(vsize); qh[14][s] = eidx; qh[15][s] = misc;
                        mode = M_DONE;
                    }
                    h_count += n;
                }
            }
            // pop: ready rays into the free lanes
            {
                const unsigned long long m = __ballot(mode == M_DONE);
                const int n = min(__popcll(m), r_count);
                const int r = pool_rank(m);
                if (mode == M_DONE && r < n) {
                    const int s = (r_head + r) & (QR_CAP - 1);
                    outk = (int)qr[0][s];
                    alpha = mk(__uint_as_float(qr[1][s]), __uint_as_float(qr[2][s]), __uint_as_float(qr[3][s]));
                    beta = mk(__uint_as_float(qr[4][s]), __uint_as_float(qr[5][s]), __uint_as_float(qr[6][s]));
                    g = mk(__uint_as_float(qr[7][s]), __uint_as_float(qr[8][s]), __uint_as_float(qr[9][s]));
                    tw = __uint_as_float(qr[10][s]); cw = (int)qr[11][s]; guard = qr[12][s]; creepn = (int)qr[13][s];
                    enter_chunk((int)qr[14][s]);
                }
                r_head = (r_head + n) & (QR_CAP - 1); r_count -= n;
            }
            // one service per pass: with enough waiting for a full wave, or with too few lanes marching to be worth waiting
            const int n_busy = __popcll(__ballot(mode == M_TREE || mode == M_TWIG));
            const bool starving = n_busy < SVO_POOL_SERVE_BUSY;
            if (!w_busy && w_count > 0 && (w_count >= 64 || starving || r_count + r_res < SVO_POOL_R_LOW) && QR_CAP - r_count - r_res >= min(w_count, 64)) {
                serve = SERVE_WORLD; serve_n = min(w_count, 64); serve_head = w_head;
                w_busy = 1; r_res += serve_n;
            } else if (!h_busy && h_count > 0 && (h_count >= 32 || starving) && QW_CAP - w_count - w_res >= min(h_count, 64)) {
                serve = SERVE_HIT; serve_n = min(h_count, 64); serve_head = h_head;
                h_busy = 1; w_res += serve_n;
            } else if (tc.more && w_count + w_res + r_count + r_res < SVO_POOL_TILE_BELOW && QW_CAP - w_count - w_res >= 64) {
                serve = SERVE_TILE; serve_n = 64; w_res += 64;
            }
            pool_idle = !tc.more && serve == SERVE_NONE && w_count == 0 && r_count == 0 && h_count == 0 && w_res == 0 && r_res == 0;
            if (lane == 0) {
                pool_wr(qc, QC_W_COUNT, w_count); pool_wr(qc, QC_W_RESERVED, w_res); pool_wr(qc, QC_W_BUSY, w_busy);
                pool_wr(qc, QC_R_HEAD, r_head); pool_wr(qc, QC_R_COUNT, r_count); pool_wr(qc, QC_R_RESERVED, r_res);
                pool_wr(qc, QC_H_COUNT, h_count); pool_wr(qc, QC_H_BUSY, h_busy);
            }
            pool_unlock(qc, lane);
        }

        // ---- lanes whose ray found no room in its queue are served in place, out of line: nothing ever waits for room
        if (__ballot(mode == M_WORLD) != 0ull) {
            ++dbg_inplace;
            if (mode == M_WORLD) {
                const InplaceWorld r = pool_inplace_world(kp, P.ctab, outk, alpha, beta, tw, cw);
                cw = r.cw;
                if (r.chunk >= 0) enter_chunk(r.chunk); else mode = M_DONE;
            }
        }
        if (__ballot(mode == M_HIT) != 0ull) {
            if (mode == M_HIT) {
                V3 vlo; float vsize; uint32_t eidx, misc;
                hit_voxel(vlo, vsize, eidx, misc);
                const InplaceHit r = pool_inplace_hit(kp, P.ctab, outk, alpha, beta, tw, ci, vlo, vsize, eidx, misc);
                mode = M_DONE;
                if (A.shadow) {                                 // the lane becomes its own shadow ray (queued, or stepped in place, in the next pass)
                    alpha = r.point; beta = sdir; g = sg;
                    outk |= (int)0x80000000;
                    tw = r.tw; cw = 0; guard = 0; creepn = 0;
                    mode = r.enters ? M_WORLD : M_DONE;
                    rays_started++;
                }
            }
        }

        // ---- the service: all lanes on entries that are not theirs (pool_service)
        dbg_serves[serve]++;
        if (serve != SERVE_NONE) {
            tc = pool_service(kp, P, serve, serve_n, serve_head, lane, tc);
            rays_started += tc.started;
        }

        const unsigned long long marching0 = __ballot(mode == M_TREE || mode == M_TWIG);
        if (marching0 == 0ull) {
            // nothing to march: go round again if anything may still come (a lane holds a ray for the next exchange, a service just
            // ran, the queues hold work or another wave's reservation, tiles remain), else end
            if (__ballot(mode != M_DONE) != 0ull || serve != SERVE_NONE) continue;
            if (!pool_idle) { ++dbg_spins; __builtin_amdgcn_s_sleep(8); continue; }
            break;
        }

        // ---- the march: the stack kernel's statement
        int pass = 0;
        const int n_busy = __popcll(marching0);
        dbg_lanes += (uint32_t)n_busy;
        const int fixed_steps = (n_busy >= SVO_STEP_LANES && __ballot(creepn > 0 || creepn <= -4 * SVO_CREEP_SERIOUS) == 0ull) ? SVO_POOL_STEPS : 0;
        for (;;) {
            const int nsteps = pass == 0 ? 1 + fixed_steps : SVO_DRAIN_STEPS;
            ++dbg_statements;
            guard += (mode == M_TREE || mode == M_TWIG) ? (uint32_t)nsteps : 0u;
            march_steps_asm(mode, O, Blo, bsize, res, t, cnt, tt_saved, t_miss, it_saved, tw, cw, pux, puy, puz, valid, plev, bmask, creepn,
                            beta, g, clo, alpha, levels, nw_chunk, res_tree, wide_b, twig_off, lds_lane, SU, nsteps);
            pass += nsteps;
            // once the pool has run dry and every live lane of the wave is marching, the statement repeats at once (a launch's last
            // waves are alone on their SIMD and bound by their own instruction stream)
            if (!pool_idle || pass >= 256) break;
            const unsigned long long marching = __ballot(mode == M_TREE || mode == M_TWIG);
            if (marching == 0ull || marching != __ballot(mode != M_DONE) || __ballot(creepn > 0 || creepn <= -4 * SVO_CREEP_SERIOUS) != 0ull) break;
        }

#include "creep_block.inc"
    }

#ifdef SVO_STACK_TIMING
    (void)n_creep_runs; (void)n_creep_steps; (void)n_creep_rounds; (void)n_dbg;
#endif
    if (A.counters && lane == 0) {           // diagnostics (svo_trace_params.counters_dev with this kernel): per wave, 8 words
        uint32_t *c = A.counters + 8 * (blockIdx.x * POOL_WAVES + wave);
        c[0] = dbg_iters; c[1] = dbg_serves[1]; c[2] = dbg_serves[2]; c[3] = dbg_serves[3]; c[4] = dbg_spins; c[5] = dbg_inplace | (dbg_lanes << 12); c[6] = dbg_statements;
        c[7] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - dbg_t0);
    }
    unsigned total = rays_started;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if (lane == 0 && total) atomicAdd(&A.work[1], (unsigned long long)total);
}

} // namespace svo
