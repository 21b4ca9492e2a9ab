// kernel_stack.hip.h — SVO_KERNEL_STACK: the CDNA4 fast path of the SVO march.
//
// Same results as kernel_literal.hip.h / src/Traverse.cpp, bit for bit, on worlds whose voxel
// corners are exact floats (svo_world_info.exact_geometry: power-of-two chunk edge, positions on
// the voxel lattice), reached differently:
//
//  * Persistent single-wave workgroups pull 8x8-pixel tiles (64 rays) from an atomic cursor.
//    When >= REFILL lanes of the wave have retired (64-bit __ballot), the dead lanes are handed
//    the next ray ids by prefix rank (mbcnt) — the wave re-compacts instead of idling on its
//    slowest ray.  A primary hit turns its lane into the shadow ray in place.
//  * The three nested loops of chunkmarch/treemarch/twigmarch are flattened into one loop whose
//    body every lane runs in one of three modes, sharing one escape-distance evaluation.
//  * Descent: the position inside the chunk is reduced to integer cell coordinates at level
//    depth-2 (one exact compare-and-fix per axis reproduces the reference's float `p >= mid`
//    tests).  The children-base index of every BRANCH on the current root->node path is cached
//    in a per-lane LDS column ([level][lane]: bank = lane, conflict-free).  The next step
//    restarts below the deepest level whose cell prefix is unchanged: ~1-3 dependent node loads
//    per step instead of one per level.
//  * Bricks are tested against a 64-bit occupancy mask held in registers (one 8-byte load per
//    brick visit); the 128-byte brick line is touched only to fetch the hit material.
//  * Ray generation (camera ray, 1/dir, world-entry distance: ~10 IEEE divisions) is done once per
//    tile by all 64 lanes together and parked in LDS; a refill is then 11 LDS reads per lane, so
//    waves can refill after only a few retirements.
//  * The rare, expensive blocks (chunk step, G-buffer resolve of a primary hit) run only when a
//    ballot shows enough lanes waiting for them (or nothing else is runnable).
//  * Lane state is kept small (re-basing points pw/pt and the node box are recomputed with the
//    reference's own expressions instead of being held) so that 8 waves/SIMD fit.
//
// All float arithmetic that decides t is evaluated exactly as in the reference; only loads and
// integer bookkeeping differ.  Divisions by powers of two (chunk edge, node edge) are written as
// multiplications by the exact reciprocal, which is bit-identical.
#pragma once
#include "march.hip.h"

namespace svo {

enum : int { M_DONE = 0, M_WORLD = 1, M_TREE = 2, M_TWIG = 3, M_HIT = 4 };

#ifndef SVO_DESC_ROUNDS
#define SVO_DESC_ROUNDS 0        // >0: at most this many extra node loads per lane per iteration (time-sliced descent)
#endif
#ifndef SVO_VOTE_WORLD
#define SVO_VOTE_WORLD 16        // lanes waiting for a chunk step that make the wave run it
#endif
#ifndef SVO_VOTE_HIT
#define SVO_VOTE_HIT 16          // primary hits waiting for their G-buffer record
#endif
#ifndef SVO_VOTE_BUSY
#define SVO_VOTE_BUSY 24         // fewer marching lanes than this: serve the waiting ones regardless
#endif

// 1/x for x an exact power of two (normal range): exponent negation, no division sequence.
__device__ __forceinline__ float recip_pow2(float x) { return __uint_as_float(0x7F000000u - __float_as_uint(x)); }

// World::index(World::index_float(p)) for a power-of-two chunk edge (src/World.cpp:288-293,323-332):
// p / chunksize == p * inv (exact), and positive_mod() of a coordinate that lies within one grid
// period of the grid's first chunk is a conditional add/subtract.
__device__ __forceinline__ int chunk_index_pow2(const TraceArgs &A, V3 p)
{
    float qx = p.x * A.inv_chunksize, qy = p.y * A.inv_chunksize, qz = p.z * A.inv_chunksize;
    if (qx < 0.0f) qx -= 1.0f;
    if (qy < 0.0f) qy -= 1.0f;
    if (qz < 0.0f) qz -= 1.0f;
    const int ix = (int)qx, iy = (int)qy, iz = (int)qz;
    const int rx = ix - A.ccm[0], ry = iy - A.ccm[1], rz = iz - A.ccm[2];
    const bool near = (rx >= -A.dimw) & (rx <= A.dimw) & (ry >= -A.dimh) & (ry <= A.dimh) & (rz >= -A.dimd) & (rz <= A.dimd);
    if (!near) return pmod(iy, A.dimh) * A.dimw * A.dimd + pmod(iz, A.dimd) * A.dimw + pmod(ix, A.dimw);
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

#ifndef SVO_CREEP_ROUNDS
#define SVO_CREEP_ROUNDS 0       // >0: take up to N consecutive same-cell ("creeping") steps inside one iteration; costs ~15% on ordinary frames, halves pathological ones
#endif

template <int MAXLV, int REFILL, int WAVES_PER_SIMD>
__global__ __launch_bounds__(64, WAVES_PER_SIMD) void k_trace_stack(TraceArgs A)
{
    __shared__ uint32_t stk[MAXLV > 0 ? MAXLV : 1][64];
    __shared__ float tile_ray[11][64];              // o, d, 1/d, world-entry t, output index (as int; -1 = no ray)
    const int lane = threadIdx.x;
#ifdef SVO_STACK_TIMING
    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
    unsigned n_iters = 0, n_tree_lanes = 0, n_twig_lanes = 0, n_world_lanes = 0, n_desc_rounds = 0;
    unsigned long long cyc_desc = 0, cyc_tree = 0, cyc_twig = 0, cyc_world = 0, cyc_hit = 0, cyc_refill = 0;
#endif

    const V3 wlo = ld3(A.worldmin), whi = ld3(A.worldmax);
    const V3 sdir = ld3(A.sdir);
    const V3 sg = recip(sdir);                      // 1/sdir: the same quotient for every shadow ray
    const float eps = A.eps;
    const float csize = A.chunksize;

    // ---- wave state (uniform) ------------------------------------------------------------
    int tile_first = 0;             // first ray id of the tile being handed out
    int tile_next = 64;             // next unassigned slot of that tile (64 = exhausted)
    bool more = true;               // tiles left in the global cursor
    unsigned rays_marched = 0;      // per lane, summed at exit

    // ---- lane state ------------------------------------------------------------------------
    int mode = M_DONE;
    bool is_shadow = false;
    int outk = 0;
    V3 alpha = mk(0, 0, 0), beta = mk(0, 0, 1), g = mk(0, 0, 0);
    float tw = 0.0f, tt = 0.0f, tb = 0.0f;
    int cw = 0, it = 0, ib = 0;
    uint32_t guard = 0;
    // chunk
    V3 clo = mk(0, 0, 0);
    const uint32_t *tree = A.tree;
    uint32_t twig_off = 0;
    int levels = 0, ci = 0;
    // descent cache: cell coordinates of the last tree step and the level of the node it ended at
    int pux = 0, puy = 0, puz = 0, valid = 0;
    uint32_t last_word = 0;         // node word the last tree step ended at (valid while `valid_word`)
    bool valid_word = false;
    bool descending = false;        // the last tree step stopped part-way down (SVO_DESC_ROUNDS); resume at desc_node
    uint32_t desc_node = 0;
    // brick
    unsigned long long bmask = 0;

    for (;;) {
        // ==== refill retired lanes =============================================================
        unsigned long long dead = __ballot(mode == M_DONE);
        while (more && __popcll(dead) >= REFILL) {
            if (tile_next >= 64) {
                unsigned long long tix = 0;
                if (lane == 0) tix = atomicAdd(&A.work[0], 1ull);
                const int t32 = __builtin_amdgcn_readfirstlane((int)tix);
                if (t32 >= A.ntiles) { more = false; break; }
                tile_first = t32 * 64;
                tile_next = 0;
                // all 64 lanes generate the tile's rays (src/Traverse.cpp:135-140 included) and park them in LDS
                {
                    const int id = tile_first + lane;
                    bool ok;
                    int k = -1;
                    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
                    if (A.from_camera) {
                        const unsigned tile = (unsigned)t32;
                        const int lx = (int)(tile % (unsigned)A.tiles_per_row) * 8 + (lane & 7);
                        const int ly = (int)(tile / (unsigned)A.tiles_per_row) * 8 + (lane >> 3);
                        ok = (lx < A.w) & (ly < A.h);
                        k = ly * A.w + lx;
                        int px = 0, py = 0;
                        if (ok) local_to_pixel(A, lx, ly, px, py);
                        if (ok && (py >= A.imgh || px >= A.imgw)) { store_miss(A.out, k, 0); ok = false; }
                        if (ok) camera_ray(A, px, py, o, d);
                    } else {
                        ok = id < A.n;
                        k = id;
                        if (ok) { o = ld3(A.origins + 3 * (long long)id); d = ld3(A.dirs + 3 * (long long)id); }
                    }
                    const V3 gg = recip(d);
                    float t0 = 0.0f;
                    if (ok) {
                        bool hit = true;
                        if (!inside(o, wlo, whi)) t0 = enter(o, d, wlo, whi, hit) + eps;
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
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dead, 0u));
            const int avail = 64 - tile_next;
            if (mode == M_DONE && rank < avail) {
                const int slot = tile_next + rank;
                const int k = __float_as_int(tile_ray[10][slot]);
                if (k >= 0) {
                    outk = k;
                    alpha = mk(tile_ray[0][slot], tile_ray[1][slot], tile_ray[2][slot]);
                    beta = mk(tile_ray[3][slot], tile_ray[4][slot], tile_ray[5][slot]);
                    g = mk(tile_ray[6][slot], tile_ray[7][slot], tile_ray[8][slot]);
                    tw = tile_ray[9][slot];
                    is_shadow = false;
                    cw = 0; guard = 0;
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
        const bool run_world = n_world > 0 && (n_world >= SVO_VOTE_WORLD || n_busy < SVO_VOTE_BUSY);
        const bool run_hit = n_hit > 0 && (n_hit >= SVO_VOTE_HIT || n_busy < SVO_VOTE_BUSY);

        // ==== one march step per live lane ==================================================
        // `adv`: which accumulator receives escape(E_p, g, E_lo, E_hi) + eps at the end of the step
        //   0 none, 1 tb (brick cell), 2 tt (tree node / brick exit), 3 tw (chunk exit)
        int adv = 0;
        V3 E_p = mk(0, 0, 0), E_lo = mk(0, 0, 0);
        float E_size = 0.0f;

        if (mode != M_DONE && mode != M_HIT && ++guard > STEP_GUARD) {     // runaway ray: give up, flag it
            if (is_shadow) store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG);
            else store_miss(A.out, outk, SVO_ERR_FLAG);
            mode = M_DONE;
        }

        // ---- chunk step: src/Traverse.cpp:142-156 -------------------------------------------
        if (run_world && mode == M_WORLD) {
            bool miss = cw >= A.cap_chunk;
            if (!miss) {
                cw++;
                const V3 p = alpha + beta * tw;
                miss = !inside(p, wlo, whi);
                if (!miss) {
                    ci = chunk_index_pow2(A, p);
                    const DevChunk ch = A.chunks[ci];
                    clo = ld3(ch.bmin);
                    miss = !inside(p, clo, clo + csize);
                    if (!miss) {
                        tt = 0.0f; it = 0; valid = 0; valid_word = false; descending = false;
                        tree = A.tree + ch.tree_off;
                        twig_off = (uint32_t)ch.twig_off;
                        levels = (int)ch.levels;
                        mode = M_TREE;
                    }
                }
            }
            if (miss) {
                if (!is_shadow) store_miss(A.out, outk, 0);
                mode = M_DONE;                          // shadow miss: record already says "traced, lit"
            }
        }

        if (mode == M_TREE || mode == M_TWIG) {
            const float cell = csize * __uint_as_float((uint32_t)(127 - levels) << 23);    // csize / 2^levels, exact
            const V3 pw = alpha + beta * tw;            // the chunk march's origin (src/Traverse.cpp:144,158)

            // ---- tree step: src/Traverse.cpp:79-111 ---------------------------------------
            if (mode == M_TREE) {
                const float inv_cell = recip_pow2(cell);                                    // power of two, exact
                const int nmax = (1 << levels) - 1;
                const V3 p = pw + beta * tt;
                bool leave = false;
                int ux = pux, uy = puy, uz = puz, lvl = valid;
                uint32_t node = desc_node, word = last_word;
                bool same_node = false;
                if (!descending) {
                    leave = it >= A.cap_tree;
                    if (!leave) {
                        it++;
                        leave = !inside(p, clo, clo + csize);
                    }
                    if (!leave) {
                        // integer cell coordinates at level `levels`: number of cell boundaries <= p
                        ux = (int)((p.x - clo.x) * inv_cell); uy = (int)((p.y - clo.y) * inv_cell); uz = (int)((p.z - clo.z) * inv_cell);
                        ux = ux > nmax ? nmax : ux; uy = uy > nmax ? nmax : uy; uz = uz > nmax ? nmax : uz;
                        ux -= (clo.x + (float)ux * cell > p.x) ? 1 : 0;
                        uy -= (clo.y + (float)uy * cell > p.y) ? 1 : 0;
                        uz -= (clo.z + (float)uz * cell > p.z) ? 1 : 0;

                        // levels whose cached children base is still on the path
                        const uint32_t diff = (uint32_t)((ux ^ pux) | (uy ^ puy) | (uz ^ puz));
                        const int common = levels - (diff ? 32 - __clz((int)diff) : 0);
                        if (valid_word && common >= valid) {
                            // still inside the node the previous step ended at (a creeping ray): no load at all
                            same_node = true;
                        } else {
                            const int usable = common + 1 < valid ? common + 1 : valid;
                            node = 0; lvl = 0;
                            if (usable > 0) {
                                const int sh = levels - usable;
                                node = stk[usable - 1][lane] + (uint32_t)(((ux >> sh) & 1) | (((uy >> sh) & 1) << 1) | (((uz >> sh) & 1) << 2));
                                lvl = usable;
                            }
                            word = tree[node];
                        }
                    }
                }
                if (!leave) {
#ifdef SVO_STACK_TIMING
                    unsigned long long td0; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td0) :: "memory");
#endif
                    if (!same_node) {
                        // descend; at most SVO_DESC_ROUNDS further loads this iteration (0 = until the node is found):
                        // a lane with a long way down continues next iteration instead of stalling the wave
                        for (int r = 0; node_type(word) == BRANCH && lvl < levels && (SVO_DESC_ROUNDS == 0 || r < SVO_DESC_ROUNDS); ++r) {
                            const uint32_t base = node_offset(word);
                            stk[lvl][lane] = base;
                            const int sh = levels - 1 - lvl;
                            node = base + (uint32_t)(((ux >> sh) & 1) | (((uy >> sh) & 1) << 1) | (((uz >> sh) & 1) << 2));
                            ++lvl;
                            word = tree[node];
                        }
                    }
#ifdef SVO_STACK_TIMING
                    { unsigned long long td1; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(td1) :: "memory"); cyc_desc += td1 - td0; }
#endif
                    valid = lvl; pux = ux; puy = uy; puz = uz;
                    last_word = word;
                    descending = node_type(word) == BRANCH && lvl < levels;
                    desc_node = node;
                    valid_word = !descending;
                    if (!descending) {
                        const uint32_t type = node_type(word);
                        if (type == EMPTY) {
                            const int low = (1 << (levels - lvl)) - 1;
                            const V3 lo = mk(clo.x + (float)(ux & ~low) * cell, clo.y + (float)(uy & ~low) * cell, clo.z + (float)(uz & ~low) * cell);
                            const V3 hi = lo + cell * (float)(low + 1);
                            float e = escape(p, g, lo, hi);
                            tt += e + eps;
                            // creeping (pinned on a face: escape ~ 0): take the following steps here while they
                            // stay in this node — each round is one full reference step (src/Traverse.cpp:79-90)
                            for (int r = 0; e < eps && r < SVO_CREEP_ROUNDS && it < A.cap_tree && guard < STEP_GUARD; ++r) {
                                const V3 q = pw + beta * tt;
                                if (!inside(q, clo, clo + csize)) break;
                                int vx = (int)((q.x - clo.x) * inv_cell), vy = (int)((q.y - clo.y) * inv_cell), vz = (int)((q.z - clo.z) * inv_cell);
                                vx = vx > nmax ? nmax : vx; vy = vy > nmax ? nmax : vy; vz = vz > nmax ? nmax : vz;
                                vx -= (clo.x + (float)vx * cell > q.x) ? 1 : 0;
                                vy -= (clo.y + (float)vy * cell > q.y) ? 1 : 0;
                                vz -= (clo.z + (float)vz * cell > q.z) ? 1 : 0;
                                if ((((vx ^ ux) | (vy ^ uy) | (vz ^ uz)) & ~low) != 0) break;     // left the node
                                it++; guard++;
                                e = escape(q, g, lo, hi);
                                tt += e + eps;
                            }
                        } else if (type == LEAF) {
                            tw = tw + (tt - eps);                               // src/Traverse.cpp:93,160
                            ib = (int)SVO_CELL_NONE;                            // M_HIT keeps the hit cell in ib
                            mode = M_HIT;
                        } else if (type == TWIG) {
                            if (!same_node) bmask = A.mask[twig_off + node_offset(word)];
                            tb = 0.0f; ib = 0;
                            mode = M_TWIG;
                        } else {                                                // BRANCH at the last level: malformed
                            if (is_shadow) store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG);
                            else store_miss(A.out, outk, SVO_ERR_FLAG);
                            mode = M_DONE;
                        }
                    }
                }
                if (leave) {                                                    // src/Traverse.cpp:164-168
                    adv = 3; E_p = pw; E_lo = clo; E_size = csize;
                    mode = M_WORLD;
                }
            }

            // ---- brick step: src/Traverse.cpp:54-70 (a lane that just entered a brick steps at once) --
            if (mode == M_TWIG) {
                // the brick node's box from the cached cell coordinates (exact), and the brick march's origin
                const int low = (1 << (levels - valid)) - 1;
                const V3 nlo = mk(clo.x + (float)(pux & ~low) * cell, clo.y + (float)(puy & ~low) * cell, clo.z + (float)(puz & ~low) * cell);
                const float nsize = cell * (float)(low + 1);
                const V3 pt = pw + beta * tt;                                   // src/Traverse.cpp:81,99
                bool leave = ib >= A.cap_twig;
                if (!leave) {
                    ib++;
                    const V3 p = pt + beta * tb;
                    leave = !inside(p, nlo, nlo + nsize);
                    if (!leave) {
                        const float voxel = nsize * 0.25f;                      // size / 4, exact
                        const float inv_voxel = recip_pow2(voxel);              // power of two: (p-lo)/voxel == (p-lo)*inv
                        const int ox = (int)((p.x - nlo.x) * inv_voxel), oy = (int)((p.y - nlo.y) * inv_voxel), oz = (int)((p.z - nlo.z) * inv_voxel);
                        leave = ((ox | oy | oz) < 0) | (ox > 3) | (oy > 3) | (oz > 3);
                        if (!leave) {
                            const uint32_t w = (uint32_t)(oz * 16 + oy * 4 + ox);
                            if ((bmask >> w) & 1ull) {
                                float s = tb;                                   // src/Traverse.cpp:63
                                s += tt;                                        // :101
                                tw = tw + s;                                    // :160
                                ib = (int)w;                                    // M_HIT keeps the hit cell in ib
                                mode = M_HIT;
                            } else {
                                const V3 vlo = mk(nlo.x + (float)ox * voxel, nlo.y + (float)oy * voxel, nlo.z + (float)oz * voxel);
                                const V3 vhi = vlo + voxel;
                                float e = escape(p, g, vlo, vhi);
                                tb += e + eps;
                                // creeping inside one empty cell: each round is one full reference step (src/Traverse.cpp:54-70)
                                for (int r = 0; e < eps && r < SVO_CREEP_ROUNDS && ib < A.cap_twig && guard < STEP_GUARD; ++r) {
                                    const V3 q = pt + beta * tb;
                                    if (!inside(q, nlo, nlo + nsize)) break;
                                    const int qx = (int)((q.x - nlo.x) * inv_voxel), qy = (int)((q.y - nlo.y) * inv_voxel), qz = (int)((q.z - nlo.z) * inv_voxel);
                                    if ((qx != ox) | (qy != oy) | (qz != oz)) break;                // left the cell
                                    ib++; guard++;
                                    e = escape(q, g, vlo, vhi);
                                    tb += e + eps;
                                }
                            }
                        }
                    }
                }
                if (leave) {                                                    // src/Traverse.cpp:104-105
                    adv = 2; E_p = pt; E_lo = nlo; E_size = nsize;
                    mode = M_TREE;
                }
            }
        }

        // ---- the one escape evaluation of the step: t += escape + EPS ---------------------------
        if (adv) {
            const float e = escape(E_p, g, E_lo, E_lo + E_size) + eps;
            if (adv == 2) tt += e;
            else tw += e;
        }

        // ---- hits.  A shadow ray only sets a flag; a primary hit waits (M_HIT) until the wave votes to
        //      resolve: G-buffer record, then the lane becomes its own shadow ray -------------------
        if (mode == M_HIT && is_shadow) {
            store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_SHADOWED);
            mode = M_DONE;
        }
        if (run_hit && mode == M_HIT) {
            const uint32_t hit_cell = (uint32_t)ib;
            // which voxel: re-derive the node from the descent cache (no state was kept for it)
            const float cell = csize * __uint_as_float((uint32_t)(127 - levels) << 23);
            const int low = (1 << (levels - valid)) - 1;
            const int sh = levels - valid;
            const uint32_t node = valid > 0
                ? stk[valid - 1][lane] + (uint32_t)(((pux >> sh) & 1) | (((puy >> sh) & 1) << 1) | (((puz >> sh) & 1) << 2))
                : 0u;
            const uint32_t word = tree[node];
            V3 vlo = mk(clo.x + (float)(pux & ~low) * cell, clo.y + (float)(puy & ~low) * cell, clo.z + (float)(puz & ~low) * cell);
            float vsize = cell * (float)(low + 1);
            uint32_t material = node_offset(word) & 0xFFFFu;
            if (hit_cell != SVO_CELL_NONE) {
                vsize = vsize * 0.25f;
                vlo = mk(vlo.x + (float)(hit_cell & 3u) * vsize, vlo.y + (float)((hit_cell >> 2) & 3u) * vsize, vlo.z + (float)(hit_cell >> 4) * vsize);
                material = A.twig[((unsigned long long)twig_off + node_offset(word)) * TWIG_WORDS + hit_cell];
            }
            const V3 point = alpha + beta * (tw - eps);
            const V3 n = cube_normal_pow2(point, vlo, vsize, eps);
            const uint32_t flags = SVO_HIT_FLAG | (A.shadow ? SVO_SHADOW_TRACED : 0u);
            store_hit(A.out, outk, tw, n, material, flags, (uint32_t)ci, node, hit_cell);
            mode = M_DONE;
            if (A.shadow) {                             // the lane becomes its own shadow ray
                alpha = point; beta = sdir; g = sg;
                is_shadow = true;
                tw = 0.0f; cw = 0; guard = 0;
                bool hit = true;
                if (!inside(alpha, wlo, whi)) tw = enter(alpha, beta, wlo, whi, hit) + eps;
                mode = hit ? M_WORLD : M_DONE;          // a shadow ray that misses the world box stays "lit"
                rays_marched++;
            }
        }
    }

    unsigned total = rays_marched;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if (lane == 0 && total) atomicAdd(&A.work[1], (unsigned long long)total);
#ifdef SVO_STACK_TIMING
    if (lane == 0 && A.counters) {       // diagnostic build only: per-wave [start, end] in 10 ns ticks, iterations, rays
        uint4 c; c.x = (uint32_t)t_begin; c.y = (uint32_t)__builtin_amdgcn_s_memrealtime(); c.z = n_iters; c.w = total;
        reinterpret_cast<uint4 *>(A.counters)[2 * blockIdx.x] = c;
        uint4 e; e.x = (uint32_t)(cyc_desc >> 4); e.y = 0; e.z = n_tree_lanes; e.w = n_twig_lanes | (n_world_lanes << 20);
        reinterpret_cast<uint4 *>(A.counters)[2 * blockIdx.x + 1] = e;
    }
#endif
}

} // namespace svo
