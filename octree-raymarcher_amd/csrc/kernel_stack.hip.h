// kernel_stack.hip.h — SVO_KERNEL_STACK: the CDNA4 fast path of the SVO march.
//
// Same results as kernel_literal.hip.h / src/Traverse.cpp, bit for bit, on worlds whose voxel
// corners are exact floats (svo_world_info.exact_geometry), reached differently:
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
//
// All float arithmetic that decides t is evaluated exactly as in the reference; only loads and
// integer bookkeeping differ.
#pragma once
#include "march.hip.h"

namespace svo {

enum : int { M_DONE = 0, M_WORLD = 1, M_TREE = 2, M_TWIG = 3 };

template <int MAXLV, int REFILL>
__global__ __launch_bounds__(64) void k_trace_stack(TraceArgs A)
{
    __shared__ uint32_t stk[MAXLV > 0 ? MAXLV : 1][64];
    const int lane = threadIdx.x;

    const V3 wlo = ld3(A.worldmin), whi = ld3(A.worldmax);
    const V3 sdir = ld3(A.sdir);
    const float eps = A.eps;
    const float csize = A.chunksize;

    // ---- wave state (uniform) ------------------------------------------------------------
    long long tile_first = 0;       // first ray id of the tile being handed out
    int tile_next = 64;             // next unassigned slot of that tile (64 = exhausted)
    bool more = true;               // tiles left in the global cursor
    unsigned rays_marched = 0;      // per lane, summed at exit

    // ---- lane state ------------------------------------------------------------------------
    int mode = M_DONE;
    bool is_shadow = false;
    long long outk = 0;
    V3 alpha = mk(0, 0, 0), beta = mk(0, 0, 1), g = mk(0, 0, 0);
    float tw = 0.0f, tt = 0.0f, tb = 0.0f;
    int cw = 0, it = 0, ib = 0;
    uint32_t guard = 0;
    // chunk
    V3 pw = mk(0, 0, 0), clo = mk(0, 0, 0);
    const uint32_t *tree = A.tree;
    unsigned long long twig_off = 0;
    int levels = 0, ci = 0;
    float cell = 1.0f, inv_cell = 1.0f;
    // descent cache
    int pux = 0, puy = 0, puz = 0, valid = 0;
    // brick
    V3 pt = mk(0, 0, 0), nlo = mk(0, 0, 0);
    float nsize = 0.0f;
    unsigned long long bmask = 0;
    uint32_t brick = 0, nodeidx = 0;

    // Start (or restart, for the shadow ray) the world-level march of this lane's ray:
    // src/Traverse.cpp:135-140.
    auto begin_march = [&]() {
        g = recip(beta);
        tw = 0.0f;
        cw = 0;
        guard = 0;
        bool hit = true;
        if (!inside(alpha, wlo, whi)) tw = enter(alpha, beta, wlo, whi, hit) + eps;
        mode = hit ? M_WORLD : M_DONE;
        rays_marched++;
        return hit;
    };

    for (;;) {
        // ==== refill retired lanes =============================================================
        unsigned long long dead = __ballot(mode == M_DONE);
        while (more && __popcll(dead) >= REFILL) {
            if (tile_next >= 64) {
                unsigned long long tix = 0;
                if (lane == 0) tix = atomicAdd(&A.work[0], 1ull);
                tix = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(tix >> 32)) << 32) |
                      (unsigned)__builtin_amdgcn_readfirstlane((int)(tix & 0xFFFFFFFFull));
                if (tix >= (unsigned long long)A.ntiles) { more = false; break; }
                tile_first = (long long)tix * 64;
                tile_next = 0;
            }
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(dead >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)dead, 0u));
            const int avail = 64 - tile_next;
            if (mode == M_DONE && rank < avail) {
                const long long id = tile_first + tile_next + rank;
                bool ok;
                V3 o, d;
                if (A.from_camera) {
                    const int l = (int)(id & 63);
                    const unsigned tile = (unsigned)(id >> 6);
                    const int lx = (int)(tile % (unsigned)A.tiles_per_row) * 8 + (l & 7);
                    const int ly = (int)(tile / (unsigned)A.tiles_per_row) * 8 + (l >> 3);
                    ok = (lx < A.w) & (ly < A.h);
                    outk = (long long)ly * A.w + lx;
                    int px = 0, py = 0;
                    if (ok) local_to_pixel(A, lx, ly, px, py);
                    if (ok && (py >= A.imgh || px >= A.imgw)) { store_miss(A.out, outk, 0); ok = false; }
                    if (ok) camera_ray(A, px, py, o, d);
                } else {
                    ok = id < A.n;
                    outk = id;
                    if (ok) { o = ld3(A.origins + 3 * id); d = ld3(A.dirs + 3 * id); }
                }
                if (ok) {
                    alpha = o; beta = d;
                    is_shadow = false;
                    if (!begin_march()) store_miss(A.out, outk, 0);
                }
            }
            const int ndead = __popcll(dead);
            tile_next += (ndead < avail) ? ndead : avail;
            dead = __ballot(mode == M_DONE);
        }
        if (dead == ~0ull) break;                       // nothing alive and nothing left to fetch

        // ==== one march step per live lane ==================================================
        // `adv`: which accumulator receives escape(E_p, g, E_lo, E_hi) + eps at the end of the step
        //   0 none, 1 tb (brick cell), 2 tt (tree node / brick exit), 3 tw (chunk exit)
        int adv = 0;
        V3 E_p = mk(0, 0, 0), E_lo = mk(0, 0, 0), E_hi = mk(0, 0, 0);
        bool hit_now = false;
        Voxel vox; vox.lo = mk(0, 0, 0); vox.size = 0; vox.material = 0; vox.node = 0; vox.cell = 0;

        if (mode != M_DONE && ++guard > STEP_GUARD) {     // runaway ray: give up, flag it
            if (is_shadow) store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG);
            else store_miss(A.out, outk, SVO_ERR_FLAG);
            mode = M_DONE;
        }

        // ---- chunk step: src/Traverse.cpp:142-156 -------------------------------------------
        if (mode == M_WORLD) {
            bool miss = cw >= A.cap_chunk;
            if (!miss) {
                cw++;
                const V3 p = alpha + beta * tw;
                miss = !inside(p, wlo, whi);
                if (!miss) {
                    ci = chunk_index(A, p);
                    const DevChunk ch = A.chunks[ci];
                    clo = ld3(ch.bmin);
                    miss = !inside(p, clo, clo + csize);
                    if (!miss) {
                        pw = p;
                        tt = 0.0f; it = 0; valid = 0;
                        tree = A.tree + ch.tree_off;
                        twig_off = ch.twig_off;
                        levels = (int)ch.levels;
                        cell = csize * __uint_as_float((uint32_t)(127 - levels) << 23);      // csize / 2^levels, exact
                        inv_cell = 1.0f / cell;                                               // power of two, exact
                        mode = M_TREE;
                    }
                }
            }
            if (miss) {
                if (!is_shadow) store_miss(A.out, outk, 0);
                mode = M_DONE;                          // shadow miss: record already says "traced, lit"
            }
        }

        // ---- tree step: src/Traverse.cpp:79-111 -------------------------------------------
        if (mode == M_TREE) {
            bool leave = it >= A.cap_tree;
            if (!leave) {
                it++;
                const V3 p = pw + beta * tt;
                leave = !inside(p, clo, clo + csize);
                if (!leave) {
                    // integer cell coordinates at level `levels`: count of cell boundaries <= p.
                    const int nmax = (1 << levels) - 1;
                    int ux = (int)((p.x - clo.x) * inv_cell), uy = (int)((p.y - clo.y) * inv_cell), uz = (int)((p.z - clo.z) * inv_cell);
                    ux = ux > nmax ? nmax : ux; uy = uy > nmax ? nmax : uy; uz = uz > nmax ? nmax : uz;
                    ux -= (clo.x + (float)ux * cell > p.x) ? 1 : 0;
                    uy -= (clo.y + (float)uy * cell > p.y) ? 1 : 0;
                    uz -= (clo.z + (float)uz * cell > p.z) ? 1 : 0;

                    // levels whose cached children base is still on the path
                    const uint32_t diff = (uint32_t)((ux ^ pux) | (uy ^ puy) | (uz ^ puz));
                    const int common = levels - (diff ? 32 - __clz((int)diff) : 0);   // leading bit-levels shared with the previous step
                    int usable = common + 1 < valid ? common + 1 : valid;
                    uint32_t node = 0;
                    int lvl = 0;
                    if (usable > 0) {
                        const int k = usable - 1, sh = levels - 1 - k;
                        node = stk[k][lane] + (uint32_t)(((ux >> sh) & 1) | (((uy >> sh) & 1) << 1) | (((uz >> sh) & 1) << 2));
                        lvl = usable;
                    }
                    uint32_t word = tree[node];
                    while (node_type(word) == BRANCH && lvl < levels) {
                        const uint32_t base = node_offset(word);
                        stk[lvl][lane] = base;
                        const int sh = levels - 1 - lvl;
                        node = base + (uint32_t)(((ux >> sh) & 1) | (((uy >> sh) & 1) << 1) | (((uz >> sh) & 1) << 2));
                        ++lvl;
                        word = tree[node];
                    }
                    valid = lvl; pux = ux; puy = uy; puz = uz;

                    // the node's box, exact: lo = clo + (u & ~low) * cell, size = cell * 2^(levels-lvl)
                    const int low = (1 << (levels - lvl)) - 1;
                    const float size = cell * (float)(low + 1);
                    const V3 lo = mk(clo.x + (float)(ux & ~low) * cell, clo.y + (float)(uy & ~low) * cell, clo.z + (float)(uz & ~low) * cell);
                    const uint32_t type = node_type(word);
                    if (type == EMPTY) {
                        adv = 2; E_p = p; E_lo = lo; E_hi = lo + size;
                    } else if (type == LEAF) {
                        const float s = tt - eps;                              // src/Traverse.cpp:93
                        tw = tw + s;
                        hit_now = true;
                        vox.lo = lo; vox.size = size; vox.material = node_offset(word) & 0xFFFFu; vox.node = node; vox.cell = SVO_CELL_NONE;
                    } else if (type == TWIG) {
                        brick = node_offset(word);
                        bmask = A.mask[twig_off + brick];
                        nodeidx = node;
                        pt = p; nlo = lo; nsize = size;
                        tb = 0.0f; ib = 0;
                        mode = M_TWIG;
                    } else {                                                    // BRANCH at the last level: malformed
                        if (is_shadow) store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_ERR_FLAG);
                        else store_miss(A.out, outk, SVO_ERR_FLAG);
                        mode = M_DONE;
                    }
                }
            }
            if (leave) {                                                        // src/Traverse.cpp:164-168
                adv = 3; E_p = pw; E_lo = clo; E_hi = clo + csize;
                mode = M_WORLD;
            }
        }

        // ---- brick step: src/Traverse.cpp:54-70 (lanes that just entered a brick step at once) --
        if (mode == M_TWIG && !hit_now) {
            bool leave = ib >= A.cap_twig;
            if (!leave) {
                ib++;
                const V3 p = pt + beta * tb;
                leave = !inside(p, nlo, nlo + nsize);
                if (!leave) {
                    const float voxel = nsize * 0.25f;                         // size / 4, exact
                    const float inv_voxel = 1.0f / voxel;                      // power of two: (p-lo)/voxel == (p-lo)*inv
                    const int ox = (int)((p.x - nlo.x) * inv_voxel), oy = (int)((p.y - nlo.y) * inv_voxel), oz = (int)((p.z - nlo.z) * inv_voxel);
                    leave = ((ox | oy | oz) < 0) | (ox > 3) | (oy > 3) | (oz > 3);
                    if (!leave) {
                        const uint32_t w = (uint32_t)(oz * 16 + oy * 4 + ox);
                        const V3 vlo = mk(nlo.x + (float)ox * voxel, nlo.y + (float)oy * voxel, nlo.z + (float)oz * voxel);
                        if ((bmask >> w) & 1ull) {
                            float s = tb;                                       // src/Traverse.cpp:63
                            s += tt;                                            // :101
                            tw = tw + s;                                        // :160
                            hit_now = true;
                            vox.lo = vlo; vox.size = voxel; vox.node = nodeidx; vox.cell = w;
                            vox.material = A.twig[(twig_off + brick) * TWIG_WORDS + w];
                        } else {
                            adv = 1; E_p = p; E_lo = vlo; E_hi = vlo + voxel;
                        }
                    }
                }
            }
            if (leave) {                                                        // src/Traverse.cpp:104-105
                adv = 2; E_p = pt; E_lo = nlo; E_hi = nlo + nsize;
                mode = M_TREE;
            }
        }

        // ---- the one escape evaluation of the step: t += escape + EPS ---------------------------
        if (adv) {
            const float e = escape(E_p, g, E_lo, E_hi) + eps;
            if (adv == 1) tb += e;
            else if (adv == 2) tt += e;
            else tw += e;
        }

        // ---- hits: G-buffer record, then the lane becomes its own shadow ray ---------------
        if (hit_now) {
            if (!is_shadow) {
                const V3 point = alpha + beta * (tw - eps);
                const V3 n = cube_normal(point, vox.lo, vox.lo + vox.size, eps);
                const uint32_t flags = SVO_HIT_FLAG | (A.shadow ? SVO_SHADOW_TRACED : 0u);
                store_hit(A.out, outk, tw, n, vox.material, flags, (uint32_t)ci, vox.node, vox.cell);
                mode = M_DONE;
                if (A.shadow) {
                    alpha = point; beta = sdir;
                    is_shadow = true;
                    begin_march();                      // a shadow ray that misses the world box stays "lit"
                }
            } else {
                store_flags(A.out, outk, SVO_HIT_FLAG | SVO_SHADOW_TRACED | SVO_SHADOWED);
                mode = M_DONE;
            }
        }
    }

    unsigned total = rays_marched;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if (lane == 0 && total) atomicAdd(&A.work[1], (unsigned long long)total);
}

} // namespace svo
