// kernel_literal.hip.h — SVO_KERNEL_LITERAL: one thread per ray, the reference's control flow as
// it stands (nested chunk / tree / brick loops, descent restarted from the chunk root at every
// step, float child selection).  Works for any geometry; also produces the reference work
// counters (node words, brick cells, chunk descriptors, tree steps per ray) that bench.py prices
// the algorithmic bytes from.  The fast path is kernel_stack.hip.h.
//
//   traverse   src/Traverse.cpp:34-48      twigmarch  src/Traverse.cpp:50-72
//   treemarch  src/Traverse.cpp:74-113     chunkmarch src/Traverse.cpp:127-171
#pragma once
#include "march.hip.h"

namespace svo {

struct LitCounters { uint32_t node_words, brick_cells, chunk_descs, tree_steps; };

// Creeping rays in closed form (see the creep block of kernel_stack.hip.h for the argument): a ray that sits exactly on the
// lower face of an empty cell [lo, lo + size) on an axis it moves down along gets cubeEscapeDistance == -0 and advances by
// EPS alone (src/Traverse.cpp:25-32 has no guard) - for thousands of steps.  After one such step, lit_creep_run returns how
// many FURTHER consecutive steps are of that kind, K >= 0: while the position q_k = a + b*t_k stays inside the cell
// (lo <= q_k < hi on every axis; for a brick cell also the reference's own truncated cell index, :58), an axis pinned at the
// first of them stays pinned (monotone), the escape is -0 and t_{k+1} = t_k + EPS - and that sum is exact while t_0 and EPS
// (a power of two) are multiples of ulp(t_k), so t_k = t_0 + k*EPS without rounding.  K is found by doubling + bit descent
// (~2 log2 K probes); the caller then takes the K steps at once: same t, same counters, caps honoured (kmax).
__device__ inline int lit_creep_run(V3 a, V3 b, V3 g, float t0, V3 lo, float size, float eps, int kmax,
                                    bool brick, V3 blo, float voxel)
{
    const uint32_t eb = __float_as_uint(eps);
    const bool eps_pow2 = (eb & 0x807FFFFFu) == 0u && eb >= 0x00800000u && eb < 0x7F800000u;
    const float fmax = __uint_as_float(0x7F7FFFFFu);
    if (!eps_pow2 || kmax <= 0 || !(fabsf(g.x) <= fmax) || !(fabsf(g.y) <= fmax) || !(fabsf(g.z) <= fmax)) return 0;
    const V3 hi = lo + size;
    const V3 q0 = a + b * t0;
    bool go = (q0.x >= lo.x) & (q0.y >= lo.y) & (q0.z >= lo.z) & (q0.x < hi.x) & (q0.y < hi.y) & (q0.z < hi.z);
    go &= ((b.x < 0.0f) & (q0.x == lo.x)) | ((b.y < 0.0f) & (q0.y == lo.y)) | ((b.z < 0.0f) & (q0.z == lo.z));
    float fvx = 0.0f, fvy = 0.0f, fvz = 0.0f;
    if (brick) {        // the cell the reference's truncation finds for q0 must be this one
        const V3 f = (q0 - blo) / voxel;
        fvx = (float)(int)f.x; fvy = (float)(int)f.y; fvz = (float)(int)f.z;
        go &= (blo.x + fvx * voxel == lo.x) & (blo.y + fvy * voxel == lo.y) & (blo.z + fvz * voxel == lo.z);
    }
    if (!go) return 0;
    const uint32_t t0b = __float_as_uint(t0);
    const int e_eps = (int)(eb >> 23);
    int K = 0;
    bool up = true;
    for (int bit = 1; bit > 0;) {
        const int cand = up ? bit : K + bit;
        const float tk = t0 + (float)(cand - 1) * eps;          // position before the cand-th step
        const float tn = t0 + (float)cand * eps;                // parameter after it: must be exact
        const int e_n = (int)(__float_as_uint(tn) >> 23), shift = e_n - (int)(t0b >> 23);
        bool ok = cand <= kmax && e_n - 23 <= e_eps && tn < __uint_as_float(0x7F800000u);
        ok = ok && (t0b == 0u || (t0b >= 0x00800000u && shift < 24 && (((t0b & 0x007FFFFFu) | 0x00800000u) & ((1u << (shift < 0 ? 0 : shift)) - 1u)) == 0u));
        const V3 q = a + b * tk;
        ok &= (q.x >= lo.x) & (q.y >= lo.y) & (q.z >= lo.z) & (q.x < hi.x) & (q.y < hi.y) & (q.z < hi.z);
        if (brick) {
            const V3 f = (q - blo) / voxel;
            ok &= (f.x >= fvx) & (f.x < fvx + 1.0f) & (f.y >= fvy) & (f.y < fvy + 1.0f) & (f.z >= fvz) & (f.z < fvz + 1.0f);
        }
        K = ok ? cand : K;
        if (!up) bit >>= 1;
        else if (!ok) { up = false; bit >>= 2; }                // K <= bit/2, the bits below that are open
        else if (bit >= (1 << 13)) { up = false; bit >>= 1; }
        else bit <<= 1;
    }
    return K;
}

__device__ inline bool lit_brick(const TraceArgs &A, V3 a, V3 b, V3 g, V3 lo, float size, float voxel,
                                 const uint16_t *cells, float &s, Voxel &vox, LitCounters &cnt, uint32_t &guard)
{
    const V3 hi = lo + size;
    float t = 0.0f;
    for (int c = 0; c < A.cap_twig; ++c) {
        if (++guard > STEP_GUARD) return false;
        const V3 p = a + b * t;
        if (!inside(p, lo, hi)) return false;
        const V3 f = A.glsl ? (p - lo) * (1.0f / voxel) : (p - lo) / voxel;       // shaders/Chunkmarch.glsl:201,212 / src/Traverse.cpp:58
        const int ox = (int)f.x, oy = (int)f.y, oz = (int)f.z;
        if (!inside(mk((float)ox, (float)oy, (float)oz), mk(0.0f, 0.0f, 0.0f), mk(3.0f, 3.0f, 3.0f))) return false;
        const uint32_t word = (uint32_t)(oz * 16 + oy * 4 + ox);
        cnt.brick_cells++;
        const V3 vlo = lo + mk((float)ox, (float)oy, (float)oz) * voxel;
        const uint32_t m = cells[word];
        if (m != 0) {
            s = t;
            vox.lo = vlo; vox.size = voxel; vox.material = m; vox.cell = word;
            return true;
        }
        const float e = guarded(escape(p, g, vlo, vlo + voxel), A.guard_eps) + A.eps;
        t += e;
        if (A.exact_geometry && e < 2.0f * A.eps) {             // a pinned step: the following ones in closed form (exact corners: the cell test is the reference's)
            const int K = lit_creep_run(a, b, g, t, vlo, voxel, A.eps, min(A.cap_twig - 1 - c, (int)(STEP_GUARD - guard)), true, lo, voxel);
            t += (float)K * A.eps;
            c += K; guard += (uint32_t)K; cnt.brick_cells += (uint32_t)K;
        }
    }
    return false;
}

__device__ inline bool lit_tree(const TraceArgs &A, V3 a, V3 b, V3 g, const DevChunk &ch, float rootsize,
                                float &s, Voxel &vox, LitCounters &cnt, uint32_t &guard)
{
    const uint32_t *tree = A.tree + ch.tree_off;
    const V3 rlo = ld3(ch.bmin), rhi = rlo + rootsize;
    float t = 0.0f;
    for (int i = 0; i < A.cap_tree; ++i) {
        if (++guard > STEP_GUARD) return false;
        const V3 p = a + b * t;
        if (!inside(p, rlo, rhi)) return false;
        cnt.tree_steps++;

        V3 lo = rlo;
        float size = rootsize;
        uint32_t node = 0, word;
        uint32_t words = 0;                                     // node words this step reads (the same for every step into this leaf)
        for (int lv = 0;; ++lv) {
            cnt.node_words++; words++;
            word = tree[node];
            if (node_type(word) != BRANCH || lv >= 32) break;
            const float half = size * 0.5f;
            const V3 mid = lo + half;
            const bool gx = p.x >= mid.x, gy = p.y >= mid.y, gz = p.z >= mid.z;
            lo = lo + mk(gx ? 1.0f : 0.0f, gy ? 1.0f : 0.0f, gz ? 1.0f : 0.0f) * half;
            node = node_offset(word) + (uint32_t)gx + 2u * (uint32_t)gy + 4u * (uint32_t)gz;
            size = half;
        }
        const uint32_t type = node_type(word);
        if (type == LEAF) {
            s = t - A.leaf_back;                                // src/Traverse.cpp:93 (t - EPS) / shaders/Chunkmarch.glsl:266 (t)
            vox.lo = lo; vox.size = size; vox.material = node_offset(word) & 0xFFFFu; vox.node = node; vox.cell = SVO_CELL_NONE;
            return true;
        }
        if (type == TWIG) {
            const float voxel = size / 4.0f;
            const uint16_t *cells = A.twig + (ch.twig_off + node_offset(word)) * TWIG_WORDS;
            if (lit_brick(A, p, b, g, lo, size, voxel, cells, s, vox, cnt, guard)) {
                s += t;
                vox.node = node;
                return true;
            }
        } else if (type == BRANCH) {
            return false;                                   // deeper than 32 levels: malformed
        }
        const float e = guarded(escape(p, g, lo, lo + size), A.guard_eps) + A.eps;
        t += e;
        if (A.exact_geometry && type == EMPTY && e < 2.0f * A.eps) {   // a pinned step over an EMPTY node: the following ones in closed form
            const int K = lit_creep_run(a, b, g, t, lo, size, A.eps, min(A.cap_tree - 1 - i, (int)(STEP_GUARD - guard)), false, lo, size);
            t += (float)K * A.eps;
            i += K; guard += (uint32_t)K; cnt.tree_steps += (uint32_t)K; cnt.node_words += (uint32_t)K * words;
        }
    }
    return false;
}

// `runaway` is set when the ray used up STEP_GUARD march steps (the kernels' bound on a single ray; the reference itself
// would keep going): the caller flags the record with SVO_ERR_FLAG, as the stack kernel does.
__device__ inline bool lit_world(const TraceArgs &A, V3 alpha, V3 beta, float &tout, Voxel &vox, uint32_t &chunk,
                                 LitCounters &cnt, bool &runaway)
{
    const V3 wlo = ld3(A.worldmin), whi = ld3(A.worldmax);
    const V3 g = recip(beta);
    float t = 0.0f;
    bool hit = true;
    if (!inside(alpha, wlo, whi)) t = (A.glsl ? enter_glsl(alpha, g, wlo, whi, hit) : enter(alpha, beta, wlo, whi, hit)) + A.eps;
    if (!hit) return false;
    uint32_t guard = 0;
    for (int c = 0; c < A.cap_chunk; ++c) {
        if (++guard > STEP_GUARD) { runaway = true; return false; }
        const V3 p = alpha + beta * t;
        if (!inside(p, wlo, whi)) return false;
        const int ci = chunk_index(A, p);
        cnt.chunk_descs++;
        const DevChunk ch = A.chunks[ci];
        const V3 clo = ld3(ch.bmin), chi = clo + A.chunksize;
        if (!A.glsl && !inside(p, clo, chi)) return false;      // (src/Traverse.cpp:154-155; the shader has no such check: its treemarch just fails)
        float s = 0.0f;
        const float rootsize = A.chunksize;             // Ocroot::size == chunksize (checked on create)
        if (lit_tree(A, p, beta, g, ch, rootsize, s, vox, cnt, guard)) {
            t += s;
            tout = t;
            chunk = (uint32_t)ci;
            return true;
        }
        if (guard > STEP_GUARD) { runaway = true; return false; }      // the tree / brick march gave up, not the reference's caps
        t += guarded(escape(p, g, clo, chi), A.guard_eps) + A.eps;
    }
    return false;
}

__global__ __launch_bounds__(256) void k_trace_literal(TraceArgs A)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned rays = 0;
    bool live = k < A.n;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    if (live) {
        if (A.from_camera) {
            int px, py;
            local_to_pixel(A, (int)(k % A.w), (int)(k / A.w), px, py);
            if (py >= A.imgh || px >= A.imgw) { store_miss(A.out, k, 0); live = false; }
            else camera_ray(A.cams[0], A.imgw, A.imgh, px, py, o, d);
        } else {
            o = ld3(A.origins + 3 * k);
            d = ld3(A.dirs + 3 * k);
        }
    }
    if (live) {
        LitCounters cnt = { 0, 0, 0, 0 };
        Voxel vox; vox.lo = mk(0, 0, 0); vox.size = 0; vox.material = 0; vox.node = 0; vox.cell = 0;
        float t = 0.0f;
        uint32_t chunk = 0;
        rays = 1;
        bool runaway = false;
        if (lit_world(A, o, d, t, vox, chunk, cnt, runaway)) {
            const V3 point = o + d * (t - A.eps);
            const bool face = A.normal_mode == SVO_NORMAL_FACE;
            const V3 n = face ? face_normal(point, vox.lo, vox.lo + vox.size, d) : cube_normal(point, vox.lo, vox.lo + vox.size, A.eps);
            uint32_t flags = SVO_HIT_FLAG | (face ? (uint32_t)SVO_FACE_NORMAL : 0u);
            if (A.shadow) {
                Voxel sv; float st; uint32_t sc;
                const bool occluded = lit_world(A, point, ld3(A.sdir), st, sv, sc, cnt, runaway);
                flags |= SVO_SHADOW_TRACED | (occluded ? SVO_SHADOWED : 0u) | (runaway ? (uint32_t)SVO_ERR_FLAG : 0u);
                rays = 2;
            }
            store_hit(A.out, k, t, n, vox.material, flags, chunk, vox.node, vox.cell);
        } else {
            store_miss(A.out, k, runaway ? (uint32_t)SVO_ERR_FLAG : 0u);
        }
        if (A.counters) {
            uint4 c; c.x = cnt.node_words; c.y = cnt.brick_cells; c.z = cnt.chunk_descs; c.w = cnt.tree_steps;
            reinterpret_cast<uint4 *>(A.counters)[k] = c;
        }
    }
    // rays marched: one atomic per wave (all 64 lanes reach this point)
    unsigned total = rays;
    for (int off = 32; off > 0; off >>= 1) total += __shfl_down(total, off, 64);
    if ((threadIdx.x & 63) == 0 && total) atomicAdd(&A.work[1], (unsigned long long)total);
}

} // namespace svo
