/*
 * svo_oracle.c — CPU ORACLE for the SVO march hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C restatement of the reference's CPU algorithm (jfjell/Octree-Raymarcher @ 2024_10_08):
 *   src/Traverse.cpp (all), src/Octree.cpp:22-176,203-443, src/BoundsPyramid.cpp,
 *   src/World.cpp:19-43,276-332, plus the per-hit extras of shaders/Chunkmarch.glsl:128-136 and
 *   shaders/World.Fragment.glsl:162-178.
 *
 * PARITY UNPINNED — see svo_oracle.h.  GLM (the reference's only arithmetic dependency, version
 * unpinned, not vendored) is restated from its published generic implementation:
 *   min(x,y) = (y < x) ? y : x      max(x,y) = (x < y) ? y : x      clamp = min(max(x,lo),hi)
 *   v / s, v * s, v + s component-wise;  ivec3(vec3) truncates;  vec3(bvec3) -> 0/1
 *   normalize(v) = v * (1 / sqrt(dot(v,v)));  simplex(vec2) = Ashima/McEwan 2-D simplex noise.
 *
 * Build: gcc -O2 -ffp-contract=off (no -ffast-math): float ops must stay separately rounded.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use this file's library.
 */
#include "svo_oracle.h"

#include <assert.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_EMPTY  0u
#define ORC_LEAF   1u
#define ORC_BRANCH 2u
#define ORC_TWIG   3u
#define TWIG_LEVELS 2
#define TWIG_SIZE   4
#define TWIG_WORDS  64

#define HIT_FLAG      (1u << 0)
#define SHADOW_TRACED (1u << 1)
#define SHADOWED      (1u << 2)
#define FACE_NORMAL   (1u << 3)
#define CELL_NONE     0xFFu

/* ---------------------------------------------------------------- tiny vec3 layer (GLM) ---- */
typedef orc_vec3 vec3;
static inline vec3 v3(float x, float y, float z) { vec3 r = { x, y, z }; return r; }
static inline vec3 v3add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 v3sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 v3mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 v3div(vec3 a, vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline vec3 v3adds(vec3 a, float s) { return v3(a.x + s, a.y + s, a.z + s); }
static inline vec3 v3muls(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 v3divs(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline float gmin(float x, float y) { return (y < x) ? y : x; }
static inline float gmax(float x, float y) { return (x < y) ? y : x; }
static inline vec3 v3min(vec3 a, vec3 b) { return v3(gmin(a.x, b.x), gmin(a.y, b.y), gmin(a.z, b.z)); }
static inline vec3 v3max(vec3 a, vec3 b) { return v3(gmax(a.x, b.x), gmax(a.y, b.y), gmax(a.z, b.z)); }
static inline float v3dot(vec3 a, vec3 b) { vec3 t = v3mul(a, b); return t.x + t.y + t.z; }
static inline vec3 v3normalize(vec3 v) { return v3muls(v, 1.0f / sqrtf(v3dot(v, v))); }

/* Octree node word, src/Octree.cpp:38-65 */
static inline uint32_t node_make(uint32_t type, uint32_t offset) { return (type << 30) | (offset & ~((uint32_t)3 << 30)); }
static inline uint32_t node_type(uint32_t v)   { return v >> 30; }
static inline uint64_t node_offset(uint32_t v) { return v & ~((uint32_t)3 << 30); }
static inline unsigned node_branch(int xg, int yg, int zg) { return (unsigned)(xg + yg * 2 + zg * 4); }
static inline void node_cut(unsigned i, int *xg, int *yg, int *zg) { *xg = (i & 1) != 0; *yg = (i & 2) != 0; *zg = (i & 4) != 0; }
/* Octwig::word, src/Octree.cpp:22-30 */
static inline unsigned twig_word(unsigned x, unsigned y, unsigned z) { return z * TWIG_SIZE * TWIG_SIZE + y * TWIG_SIZE + x; }

/* ================================================================ glm::simplex(vec2) ======= */
/* Ashima Arts / Ian McEwan 2-D simplex noise as shipped in GLM's gtc/noise (restated from the
 * published algorithm).  Parity with GLM's bits is unpinned and irrelevant to
 * traversal parity, which is defined on the same octree. */
static inline float mod289(float x) { return x - floorf(x * (1.0f / 289.0f)) * 289.0f; }
static inline float permute(float x) { return mod289(((x * 34.0f) + 1.0f) * x); }
static inline float fractf(float x) { return x - floorf(x); }

float orc_simplex2(float vx, float vy)
{
    const float Cx = 0.211324865405187f;   /* (3 - sqrt 3) / 6 */
    const float Cy = 0.366025403784439f;   /* (sqrt 3 - 1) / 2 */
    const float Cz = -0.577350269189626f;  /* -1 + 2 Cx */
    const float Cw = 0.024390243902439f;   /* 1 / 41 */

    float s  = vx * Cy + vy * Cy;
    float ix = floorf(vx + s), iy = floorf(vy + s);
    float u  = ix * Cx + iy * Cx;
    float x0x = vx - ix + u, x0y = vy - iy + u;

    float i1x = (x0x > x0y) ? 1.0f : 0.0f;
    float i1y = (x0x > x0y) ? 0.0f : 1.0f;

    float x12x = x0x + Cx, x12y = x0y + Cx, x12z = x0x + Cz, x12w = x0y + Cz;
    x12x = x12x - i1x;
    x12y = x12y - i1y;

    ix = ix - 289.0f * floorf(ix / 289.0f);      /* mod(i, 289) */
    iy = iy - 289.0f * floorf(iy / 289.0f);

    float p0 = permute(permute(iy + 0.0f) + ix + 0.0f);
    float p1 = permute(permute(iy + i1y) + ix + i1x);
    float p2 = permute(permute(iy + 1.0f) + ix + 1.0f);

    float m0 = gmax(0.5f - (x0x * x0x + x0y * x0y), 0.0f);
    float m1 = gmax(0.5f - (x12x * x12x + x12y * x12y), 0.0f);
    float m2 = gmax(0.5f - (x12z * x12z + x12w * x12w), 0.0f);
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;
    m0 = m0 * m0; m1 = m1 * m1; m2 = m2 * m2;

    float gx0 = 2.0f * fractf(p0 * Cw) - 1.0f;
    float gx1 = 2.0f * fractf(p1 * Cw) - 1.0f;
    float gx2 = 2.0f * fractf(p2 * Cw) - 1.0f;
    float h0 = fabsf(gx0) - 0.5f, h1 = fabsf(gx1) - 0.5f, h2 = fabsf(gx2) - 0.5f;
    float a0 = gx0 - floorf(gx0 + 0.5f);
    float a1 = gx1 - floorf(gx1 + 0.5f);
    float a2 = gx2 - floorf(gx2 + 0.5f);

    m0 *= 1.79284291400159f - 0.85373472095314f * (a0 * a0 + h0 * h0);
    m1 *= 1.79284291400159f - 0.85373472095314f * (a1 * a1 + h1 * h1);
    m2 *= 1.79284291400159f - 0.85373472095314f * (a2 * a2 + h2 * h2);

    float g0 = a0 * x0x + h0 * x0y;
    float g1 = a1 * x12x + h1 * x12y;
    float g2 = a2 * x12z + h2 * x12w;
    return 130.0f * (m0 * g0 + m1 * g1 + m2 * g2);
}

/* ================================================================ BoundsPyramid ============ */
static inline size_t pyr_index(size_t x, size_t z, size_t s) { return z * s + x; }     /* BoundsPyramid.cpp:31-34 */
static inline float pyr_lerp(float x0, float x1, float t) { return (float)((double)(x1 * t) + (1.0 - (double)t) * (double)x0); }  /* :36-39 */
static inline float pyr_blerp(float a, float b, float c, float d, float t, float s)    /* :41-46 */
{
    float y0 = pyr_lerp(a, b, t);
    float y1 = pyr_lerp(c, d, t);
    return pyr_lerp(y0, y1, s);
}

static void pyr_computeBase(orc_pyramid *p, float period, float xshift, float zshift)  /* :92-104 */
{
    size_t i = 0;
    for (size_t z = 0; z < p->size; ++z)
        for (size_t x = 0; x < p->size; ++x, ++i) {
            float px = ((float)x + xshift) * period;
            float pz = ((float)z + zshift) * period;
            p->basequad[i] = orc_simplex2(px, pz);
        }
}

static void pyr_computeBoundsAbove(orc_pyramid *p, size_t lv)                           /* :106-135 */
{
    size_t above = lv - 1;
    size_t s = (size_t)1 << lv;
    for (size_t z = 0; z < s; ++z)
        for (size_t x = 0; x < s; x += 2) {
            size_t i = pyr_index(x + 0, z, s);
            size_t j = pyr_index(x + 1, z, s);
            size_t k = pyr_index(x / 2, z / 2, s / 2);
            float min0 = p->minquad[above][k], min1 = p->minquad[lv][i], min2 = p->minquad[lv][j];
            float max0 = p->maxquad[above][k], max1 = p->maxquad[lv][i], max2 = p->maxquad[lv][j];
            p->minquad[above][k] = gmin(min0, gmin(min1, min2));
            p->maxquad[above][k] = gmax(max0, gmax(max1, max2));
        }
}

void orc_pyramid_init(orc_pyramid *p, size_t size, float ampl, float period, float xshift, float yshift, float zshift)
{                                                                                        /* :47-78 */
    assert(size && !(size & (size - 1)));
    p->size = size;
    p->levels = (size_t)__builtin_ctzl(size);
    p->amplitude = ampl;
    p->shift = yshift;
    p->basequad = (float *)malloc(size * size * sizeof(float));
    p->minquad = (float **)malloc((p->levels + 1) * sizeof(float *));
    p->maxquad = (float **)malloc((p->levels + 1) * sizeof(float *));
    size_t s = 1;
    for (size_t i = 0; i < p->levels; ++i, s *= 2) {
        p->minquad[i] = (float *)malloc(s * s * sizeof(float));
        for (size_t j = 0; j < s * s; ++j) p->minquad[i][j] = 1.0f;
        p->maxquad[i] = (float *)malloc(s * s * sizeof(float));
        for (size_t j = 0; j < s * s; ++j) p->maxquad[i][j] = -1.0f;
    }
    p->minquad[p->levels] = p->maxquad[p->levels] = p->basequad;
    pyr_computeBase(p, period, xshift, zshift);
    for (size_t i = 0; i < p->levels; ++i)
        pyr_computeBoundsAbove(p, p->levels - i);
}

void orc_pyramid_deinit(orc_pyramid *p)
{
    if (!p->basequad) return;
    free(p->basequad);
    for (size_t i = 0; i < p->levels; ++i) { free(p->minquad[i]); free(p->maxquad[i]); }
    free(p->minquad); free(p->maxquad);
    memset(p, 0, sizeof *p);
}

static float pyr_bound(const orc_pyramid *p, float x, float z, size_t lv, const float *q)  /* :146-174 */
{
    size_t a = (size_t)(x * (float)p->size);
    size_t b = (size_t)(z * (float)p->size);
    if (lv <= p->levels) {
        size_t d = (size_t)1 << (p->levels - lv);
        size_t i = pyr_index(a / d, b / d, p->size / d);
        return q[i] * p->amplitude + p->shift;
    }
    /* beyond the base: wrapped bilinear interpolation of the base (q is not used) */
    size_t mask = p->size - 1;
    size_t a0 = a, a1 = (a0 + 1) & mask;
    size_t b0 = b, b1 = (b0 + 1) & mask;
    float t = (float)(x * (float)p->size) - (float)a0;
    float s = (float)(z * (float)p->size) - (float)b0;
    float ba00 = p->basequad[pyr_index(a0, b0, p->size)];
    float ba01 = p->basequad[pyr_index(a1, b0, p->size)];
    float ba10 = p->basequad[pyr_index(a0, b1, p->size)];
    float ba11 = p->basequad[pyr_index(a1, b1, p->size)];
    return pyr_blerp(ba00, ba01, ba10, ba11, t, s) * p->amplitude + p->shift;
}

/* The reference evaluates minquad[lv] before the call even when lv > levels (an out-of-bounds
 * pointer read whose value is unused); the restatement passes NULL there. */
float orc_pyramid_min(const orc_pyramid *p, float x, float z, size_t lv) { return pyr_bound(p, x, z, lv, lv <= p->levels ? p->minquad[lv] : NULL); }
float orc_pyramid_max(const orc_pyramid *p, float x, float z, size_t lv) { return pyr_bound(p, x, z, lv, lv <= p->levels ? p->maxquad[lv] : NULL); }

/* ================================================================ grow() =================== */
static uint16_t heightMaterial(float y)                                                  /* Octree.cpp:69-72 */
{
    double v = (double)y / 0.03;
    v = (v < 1.0) ? 1.0 : v;          /* max(x, lo) = (x < lo) ? lo : x */
    v = (4.0 < v) ? 4.0 : v;          /* min(.., hi) = (hi < ..) ? hi : .. */
    return (uint16_t)v;
}

typedef struct { vec3 pos; float size; uint32_t depth; uint32_t offset; } ocentry;

void orc_grow(orc_root *root, vec3 position, float size, uint32_t depth, const orc_pyramid *pyr)  /* Octree.cpp:74-176 */
{
    root->position = position;
    root->size = size;
    root->depth = depth;
    root->treestoragesize = 16;
    root->trees = 1;
    root->tree = (uint32_t *)malloc(root->treestoragesize * sizeof(uint32_t));
    root->twigstoragesize = 16;
    root->twigs = 0;
    root->twig = (uint16_t *)malloc(root->twigstoragesize * TWIG_WORDS * sizeof(uint16_t));

    size_t qcap = 1024, qhead = 0, qtail = 0;          /* std::queue: FIFO */
    ocentry *q = (ocentry *)malloc(qcap * sizeof(ocentry));
    q[qtail++] = (ocentry){ position, size, 0, 0 };

    while (qhead < qtail) {
        ocentry t = q[qhead++];

        vec3 p = v3divs(v3sub(t.pos, position), size);
        float low = orc_pyramid_min(pyr, p.x, p.z, t.depth);
        float high = orc_pyramid_max(pyr, p.x, p.z, t.depth);

        if (high < t.pos.y) {
            root->tree[t.offset] = node_make(ORC_EMPTY, 0);
        } else if (low > t.pos.y + t.size) {
            root->tree[t.offset] = node_make(ORC_LEAF, heightMaterial(p.y));
        } else if (t.depth == root->depth - TWIG_LEVELS) {
            float twigLeafSize = t.size / (float)(1 << TWIG_LEVELS);
            uint16_t twig[TWIG_WORDS];
            for (int y = 0; y < TWIG_SIZE; ++y)
                for (int z = 0; z < TWIG_SIZE; ++z)
                    for (int x = 0; x < TWIG_SIZE; ++x) {
                        float dx = ((float)x * twigLeafSize) / size;
                        float dz = ((float)z * twigLeafSize) / size;
                        float h = orc_pyramid_max(pyr, p.x + dx, p.z + dz, t.depth + TWIG_LEVELS);
                        unsigned w = twig_word((unsigned)x, (unsigned)y, (unsigned)z);
                        if (h >= t.pos.y + (float)y * twigLeafSize)
                            twig[w] = heightMaterial(p.y);
                        else
                            twig[w] = 0;
                    }
            if (root->twigs >= root->twigstoragesize)
                root->twig = (uint16_t *)realloc(root->twig, (root->twigstoragesize *= 2) * TWIG_WORDS * sizeof(uint16_t));
            uint64_t offset = root->twigs++;
            root->tree[t.offset] = node_make(ORC_TWIG, (uint32_t)offset);
            memcpy(root->twig + offset * TWIG_WORDS, twig, sizeof twig);
        } else {
            if (root->trees + 8 >= root->treestoragesize)
                root->tree = (uint32_t *)realloc(root->tree, (root->treestoragesize *= 2) * sizeof(uint32_t));
            uint64_t offset = root->trees;
            if (qtail + 8 > qcap) {
                if (qhead > qcap / 2) {            /* compact the consumed prefix */
                    memmove(q, q + qhead, (qtail - qhead) * sizeof(ocentry));
                    qtail -= qhead; qhead = 0;
                }
                if (qtail + 8 > qcap) q = (ocentry *)realloc(q, (qcap *= 2) * sizeof(ocentry));
            }
            for (unsigned i = 0; i < 8; ++i) {
                int xg, yg, zg;
                node_cut(i, &xg, &yg, &zg);
                float half = t.size / 2;
                vec3 cpos = v3add(t.pos, v3muls(v3((float)xg, (float)yg, (float)zg), half));
                q[qtail++] = (ocentry){ cpos, half, t.depth + 1, (uint32_t)offset + i };
            }
            root->trees += 8;
            root->tree[t.offset] = node_make(ORC_BRANCH, (uint32_t)offset);
        }
    }
    free(q);
}

void orc_root_free(orc_root *r) { free(r->tree); free(r->twig); r->tree = NULL; r->twig = NULL; }

/* ================================================================ cube predicates ========== */
static int cubesIntersect(vec3 bmin0, vec3 bmax0, vec3 bmin1, vec3 bmax1)                /* Traverse.cpp:173-178 */
{
    return (bmax0.x >= bmin1.x && bmax0.y >= bmin1.y && bmax0.z >= bmin1.z) &&
           (bmax1.x >= bmin0.x && bmax1.y >= bmin0.y && bmax1.z >= bmin0.z);
}
static int cubeIsInside(vec3 omin, vec3 omax, vec3 imin, vec3 imax)                     /* Traverse.cpp:180-185 */
{
    return (imin.x >= omin.x && imin.y >= omin.y && imin.z >= omin.z) &&
           (omax.x >= imax.x && omax.y >= imax.y && omax.z >= imax.z);
}

static inline void delta_reset(orc_delta *d) { d->left = SIZE_MAX; d->right = 0; d->realloc_ = 0; }
static inline uint64_t u64min(uint64_t a, uint64_t b) { return b < a ? b : a; }
static inline uint64_t u64max(uint64_t a, uint64_t b) { return a < b ? b : a; }

static void twig_fill(uint16_t *t, uint16_t v) { for (int i = 0; i < TWIG_WORDS; ++i) t[i] = v; }

/* ================================================================ buildCube / destroyCube == */
static void buildCube(orc_root *root, uint64_t offset, vec3 bmin, float size, size_t depth,
                      vec3 cmin, vec3 cmax, uint16_t material, orc_delta *tree, orc_delta *twig)  /* Octree.cpp:320-430 */
{
    vec3 bmax = v3adds(bmin, size);
    if (!cubesIntersect(bmin, bmax, cmin, cmax)) return;

    uint32_t t = root->tree[offset];
    if (node_type(t) == ORC_EMPTY) {
        if (cubeIsInside(cmin, cmax, bmin, bmax)) {
            tree->left = u64min(tree->left, offset);
            tree->right = u64max(tree->right, offset + 1);
            root->tree[offset] = node_make(ORC_LEAF, material);
        } else if (depth == root->depth - TWIG_LEVELS) {
            if (root->twigs >= root->twigstoragesize) {
                root->twig = (uint16_t *)realloc(root->twig, (root->twigstoragesize *= 2) * TWIG_WORDS * sizeof(uint16_t));
                twig->realloc_ = 1;
            }
            size_t pos = root->twigs++;
            twig->left = u64min(twig->left, pos);
            twig->right = u64max(twig->right, pos + 1);
            twig_fill(root->twig + pos * TWIG_WORDS, 0);
            tree->left = u64min(tree->left, offset);
            tree->right = u64max(tree->right, offset + 1);
            root->tree[offset] = node_make(ORC_TWIG, (uint32_t)pos);
            buildCube(root, offset, bmin, size, depth, cmin, cmax, material, tree, twig);
        } else {
            if (root->trees + 8 >= root->treestoragesize) {
                root->tree = (uint32_t *)realloc(root->tree, (root->treestoragesize *= 2) * sizeof(uint32_t));
                tree->realloc_ = 1;
            }
            size_t pos = root->trees;
            tree->left = u64min(tree->left, offset);
            root->tree[offset] = node_make(ORC_BRANCH, (uint32_t)pos);
            tree->right = u64max(tree->right, pos + 7 + 1);
            for (size_t i = 0; i < 8; ++i) root->tree[pos + i] = node_make(ORC_EMPTY, 0);
            root->trees += 8;
            buildCube(root, offset, bmin, size, depth, cmin, cmax, material, tree, twig);
        }
    } else if (node_type(t) == ORC_LEAF) {
        return;
    } else if (node_type(t) == ORC_TWIG) {
        float leafsize = size / (float)(1 << TWIG_LEVELS);
        uint64_t to = node_offset(t);
        twig->left = u64min(twig->left, to);
        twig->right = u64max(twig->right, to + 1);
        for (unsigned z = 0; z < TWIG_SIZE; ++z)
            for (unsigned y = 0; y < TWIG_SIZE; ++y)
                for (unsigned x = 0; x < TWIG_SIZE; ++x) {
                    size_t i = twig_word(x, y, z);
                    vec3 leafmin = v3add(bmin, v3muls(v3((float)x, (float)y, (float)z), leafsize));
                    vec3 leafmax = v3adds(leafmin, leafsize);
                    if (root->twig[to * TWIG_WORDS + i] == 0 && cubesIntersect(leafmin, leafmax, cmin, cmax))
                        root->twig[to * TWIG_WORDS + i] = material;
                }
    } else {
        float halfsize = size * 0.5f;
        for (unsigned i = 0; i < 8; ++i) {
            int xg, yg, zg;
            node_cut(i, &xg, &yg, &zg);
            vec3 nextmin = v3add(bmin, v3muls(v3((float)xg, (float)yg, (float)zg), halfsize));
            buildCube(root, node_offset(t) + i, nextmin, halfsize, depth + 1, cmin, cmax, material, tree, twig);
        }
    }
}

void orc_build(orc_root *root, vec3 cmin, vec3 cmax, uint16_t mat, orc_delta *dtree, orc_delta *dtwig)   /* Octree.cpp:432-436 */
{
    delta_reset(dtree); delta_reset(dtwig);
    buildCube(root, 0, root->position, root->size, 0, cmin, cmax, mat, dtree, dtwig);
}

static void destroyCube(orc_root *root, uint64_t offset, vec3 bmin, float size, size_t depth,
                        vec3 cmin, vec3 cmax, orc_delta *tree, orc_delta *twig)           /* Octree.cpp:203-312 */
{
    vec3 bmax = v3adds(bmin, size);
    if (!cubesIntersect(bmin, bmax, cmin, cmax)) return;

    uint32_t t = root->tree[offset];
    if (node_type(t) == ORC_EMPTY) {
        return;
    } else if (cubeIsInside(cmin, cmax, bmin, bmax)) {
        tree->left = u64min(tree->left, offset);
        tree->right = u64max(tree->right, offset + 1);
        root->tree[offset] = node_make(ORC_EMPTY, 0);
    } else if (node_type(t) == ORC_LEAF) {
        if (depth == root->depth - TWIG_LEVELS) {
            if (root->twigs >= root->twigstoragesize) {
                root->twig = (uint16_t *)realloc(root->twig, (root->twigstoragesize *= 2) * TWIG_WORDS * sizeof(uint16_t));
                twig->realloc_ = 1;
            }
            size_t pos = root->twigs++;
            twig->left = u64min(twig->left, pos);
            twig->right = u64max(twig->right, pos + 1);
            twig_fill(root->twig + pos * TWIG_WORDS, (uint16_t)node_offset(t));
            tree->left = u64min(tree->left, offset);
            tree->right = u64max(tree->right, offset + 1);
            root->tree[offset] = node_make(ORC_TWIG, (uint32_t)pos);
            destroyCube(root, offset, bmin, size, depth, cmin, cmax, tree, twig);
        } else {
            if (root->trees + 8 >= root->treestoragesize) {
                root->tree = (uint32_t *)realloc(root->tree, (root->treestoragesize *= 2) * sizeof(uint32_t));
                tree->realloc_ = 1;
            }
            size_t pos = root->trees;
            tree->left = u64min(tree->left, offset);
            root->tree[offset] = node_make(ORC_BRANCH, (uint32_t)pos);
            tree->right = u64max(tree->right, pos + 7 + 1);
            for (size_t i = 0; i < 8; ++i) root->tree[pos + i] = node_make(ORC_LEAF, (uint32_t)node_offset(t));
            root->trees += 8;
            destroyCube(root, offset, bmin, size, depth, cmin, cmax, tree, twig);
        }
    } else if (node_type(t) == ORC_TWIG) {
        float leafsize = size / (float)(1 << TWIG_LEVELS);
        uint64_t to = node_offset(t);
        twig->left = u64min(twig->left, to);
        twig->right = u64max(twig->right, to + 1);
        for (unsigned z = 0; z < TWIG_SIZE; ++z)
            for (unsigned y = 0; y < TWIG_SIZE; ++y)
                for (unsigned x = 0; x < TWIG_SIZE; ++x) {
                    size_t i = twig_word(x, y, z);
                    vec3 leafmin = v3add(bmin, v3muls(v3((float)x, (float)y, (float)z), leafsize));
                    vec3 leafmax = v3adds(leafmin, leafsize);
                    if (cubesIntersect(leafmin, leafmax, cmin, cmax))
                        root->twig[to * TWIG_WORDS + i] = 0;
                }
    } else {
        float halfsize = size * 0.5f;
        for (unsigned i = 0; i < 8; ++i) {
            int xg, yg, zg;
            node_cut(i, &xg, &yg, &zg);
            vec3 nextmin = v3add(bmin, v3muls(v3((float)xg, (float)yg, (float)zg), halfsize));
            destroyCube(root, node_offset(t) + i, nextmin, halfsize, depth + 1, cmin, cmax, tree, twig);
        }
    }
}

void orc_destroy(orc_root *root, vec3 cmin, vec3 cmax, orc_delta *dtree, orc_delta *dtwig)  /* Octree.cpp:314-318 */
{
    delta_reset(dtree); delta_reset(dtwig);
    destroyCube(root, 0, root->position, root->size, 0, cmin, cmax, dtree, dtwig);
}

/* ================================================================ World ==================== */
static int modulo(int n, int m) { return (m + (n % m)) % m; }                            /* World.cpp:276-279 */

static int world_index2(const orc_world *w, int x, int z)                                /* World.cpp:281-286 */
{
    return modulo(z, w->depth) * w->width + modulo(x, w->width);
}

int orc_world_index3(const orc_world *w, int x, int y, int z)                            /* World.cpp:288-293 */
{
    return modulo(y, w->height) * w->width * w->depth + modulo(z, w->depth) * w->width + modulo(x, w->width);
}

void orc_world_index_float(const orc_world *w, vec3 p, int q[3])                         /* World.cpp:323-332 */
{
    float cs = (float)w->chunksize;
    float f[3] = { p.x / cs, p.y / cs, p.z / cs };
    for (int i = 0; i < 3; ++i) {
        if (f[i] < 0.0) f[i] = (float)((double)f[i] - 1.0);
        q[i] = (int)f[i];
    }
}

int orc_world_init(orc_world *w, int width, int height, int depth, int s, const int ccm[3], const orc_terrain *tp)
{                                                                                        /* World.cpp:19-43,296-321 */
    memset(w, 0, sizeof *w);
    w->width = width; w->height = height; w->depth = depth;
    w->plane = width * depth;
    w->volume = w->plane * height;
    w->chunksize = s;
    for (int i = 0; i < 3; ++i) w->chunkcoordmin[i] = ccm ? ccm[i] : 0;

    uint32_t res = tp->pyramid_resolution ? tp->pyramid_resolution : (1u << tp->depth);
    w->heightmap = (orc_pyramid *)calloc((size_t)w->plane, sizeof(orc_pyramid));
    for (int z = 0; z < depth; ++z)
        for (int x = 0; x < width; ++x) {                                                /* g_pyramid, :296-306 */
            int cx = w->chunkcoordmin[0] + x, cz = w->chunkcoordmin[2] + z;
            int i = world_index2(w, cx, cz);
            float period = 1.0f / (float)res;
            float xshift = (float)cx * (float)res + (float)tp->seed;
            float zshift = (float)cz * (float)res + (float)tp->seed;
            orc_pyramid_init(&w->heightmap[i], res, tp->amplitude, period, xshift, tp->yshift, zshift);
        }

    w->chunk = (orc_root *)calloc((size_t)w->volume, sizeof(orc_root));
    for (int z = 0; z < depth; ++z)
        for (int y = 0; y < height; ++y)
            for (int x = 0; x < width; ++x) {                                            /* g_chunk, :308-321 */
                int cx = w->chunkcoordmin[0] + x, cy = w->chunkcoordmin[1] + y, cz = w->chunkcoordmin[2] + z;
                int i = orc_world_index3(w, cx, cy, cz);
                int j = world_index2(w, cx, cz);
                vec3 p = v3muls(v3((float)cx, (float)cy, (float)cz), (float)s);
                orc_grow(&w->chunk[i], p, (float)s, tp->depth, &w->heightmap[j]);
                if (tp->water) {
                    orc_root *c = &w->chunk[i];
                    vec3 watermin = c->position;
                    vec3 watermax = v3(c->position.x + c->size, tp->water_level, c->position.z + c->size);
                    orc_delta d;
                    orc_build(c, watermin, watermax, (uint16_t)tp->water_material, &d, &d);
                }
            }
    return 0;
}

void orc_world_deinit(orc_world *w)
{
    if (w->chunk) { for (int i = 0; i < w->volume; ++i) orc_root_free(&w->chunk[i]); free(w->chunk); }
    if (w->heightmap) { for (int i = 0; i < w->plane; ++i) orc_pyramid_deinit(&w->heightmap[i]); free(w->heightmap); }
    memset(w, 0, sizeof *w);
}

/* ================================================================ Traverse.cpp ============= */
int orc_isInsideCube(vec3 p, vec3 cmin, vec3 cmax)                                       /* Traverse.cpp:18-23 */
{
    int geq = (p.x >= cmin.x) && (p.y >= cmin.y) && (p.z >= cmin.z);
    int leq = (cmax.x >= p.x) && (cmax.y >= p.y) && (cmax.z >= p.z);
    return geq && leq;
}

float orc_cubeEscapeDistance(vec3 a, vec3 b, vec3 cmin, vec3 cmax)                       /* Traverse.cpp:25-32 */
{
    vec3 gamma = v3((float)(1.0 / (double)b.x), (float)(1.0 / (double)b.y), (float)(1.0 / (double)b.z));
    vec3 tmin = v3mul(v3sub(cmin, a), gamma);
    vec3 tmax = v3mul(v3sub(cmax, a), gamma);
    vec3 t = v3max(tmin, tmax);
    return gmin(t.x, gmin(t.y, t.z));
}

typedef struct { vec3 bmin; float size; uint64_t offset; } tree_t;                       /* Traverse.h:12-19 */

/* Which twin is marched (orc_params.semantics): the CPU march of src/Traverse.cpp, or the fragment shader's,
 * shaders/Chunkmarch.glsl.  GLSL's min / max are the same selections as GLM's (GLSL 4.30 spec 8.3: "min returns y if y < x,
 * otherwise x; max returns y if x < y, otherwise x"); the shader precomputes gamma = 1.0 / beta once per ray
 * (World.Fragment.glsl:166), the same quotient the CPU code forms per call. */
#define GLSL_BIGEPS (1.0f / 16.0f)                                                        /* Chunkmarch.glsl:18 */
static float escape_of(int glsl, float eps, vec3 a, vec3 b, vec3 cmin, vec3 cmax)
{
    float d = orc_cubeEscapeDistance(a, b, cmin, cmax);
    if (glsl) d = d < eps ? GLSL_BIGEPS : d;                                             /* Chunkmarch.glsl:113 */
    return d;
}

static tree_t traverse(vec3 p, const orc_root *root, orc_counters *cnt)                  /* Traverse.cpp:34-48 */
{
    tree_t t = { root->position, root->size, 0 };
    for (;;) {
        if (cnt) cnt->node_words++;
        if (node_type(root->tree[t.offset]) != ORC_BRANCH) return t;
        float halfsize = t.size * 0.5f;
        vec3 mid = v3adds(t.bmin, halfsize);
        int gx = p.x >= mid.x, gy = p.y >= mid.y, gz = p.z >= mid.z;
        vec3 bmin = v3add(t.bmin, v3muls(v3((float)gx, (float)gy, (float)gz), halfsize));
        uint64_t i = node_branch(gx, gy, gz);
        uint64_t next = node_offset(root->tree[t.offset]) + i;
        t.bmin = bmin; t.size = halfsize; t.offset = next;
    }
}

typedef struct {            /* what the GLSL twin calls Leaf + the ids parity is graded on */
    vec3 bmin; float size; uint16_t material; uint32_t node, cell;
} voxel_t;

static int twigmarch_ex(vec3 a, vec3 b, vec3 bmin, float size, float leafsize, const uint16_t *twig,
                        float eps, int cap, int glsl, float *s, voxel_t *vox, orc_counters *cnt)   /* Traverse.cpp:50-72; Chunkmarch.glsl:190-238 */
{
    vec3 bmax = v3adds(bmin, size);
    float inv_leafsize = 1.0f / leafsize;                                                /* Chunkmarch.glsl:201 */
    float t = 0.0f;
    for (int c = 0; c < cap; ++c) {
        vec3 p = v3add(a, v3muls(b, t));
        if (!orc_isInsideCube(p, bmin, bmax)) return 0;
        vec3 f = glsl ? v3muls(v3sub(p, bmin), inv_leafsize) : v3divs(v3sub(p, bmin), leafsize);      /* :212 / Traverse.cpp:58 */
        int ox = (int)f.x, oy = (int)f.y, oz = (int)f.z;
        if (!orc_isInsideCube(v3((float)ox, (float)oy, (float)oz), v3(0, 0, 0), v3(TWIG_SIZE - 1, TWIG_SIZE - 1, TWIG_SIZE - 1))) return 0;
        uint32_t word = twig_word((unsigned)ox, (unsigned)oy, (unsigned)oz);
        if (cnt) cnt->brick_cells++;
        vec3 leafmin = v3add(bmin, v3muls(v3((float)ox, (float)oy, (float)oz), leafsize));
        if (twig[word] != 0) {
            *s = t;
            if (vox) { vox->bmin = leafmin; vox->size = leafsize; vox->material = twig[word]; vox->cell = word; }
            return 1;
        }
        vec3 leafmax = v3adds(leafmin, leafsize);
        float escape = escape_of(glsl, eps, p, b, leafmin, leafmax);
        t += escape + eps;
    }
    return 0;
}

static int treemarch_ex(vec3 a, vec3 b, const orc_root *root, float eps, int cap, int twigcap, int glsl,
                        float *s, voxel_t *vox, orc_counters *cnt)                       /* Traverse.cpp:74-113; Chunkmarch.glsl:240-295 */
{
    vec3 rmin = root->position;
    vec3 rmax = v3adds(root->position, root->size);
    float t = 0.0f;
    for (int i = 0; i < cap; ++i) {
        vec3 p = v3add(a, v3muls(b, t));
        if (!orc_isInsideCube(p, rmin, rmax)) return 0;
        if (cnt) cnt->tree_steps++;
        tree_t tree = traverse(p, root, cnt);
        uint32_t word = root->tree[tree.offset];
        uint32_t type = node_type(word);
        if (type == ORC_EMPTY) {
            float escape = escape_of(glsl, eps, p, b, tree.bmin, v3adds(tree.bmin, tree.size));
            t += escape + eps;
        } else if (type == ORC_LEAF) {
            *s = glsl ? t : t - eps;                                                     /* Chunkmarch.glsl:266 / Traverse.cpp:93 */
            if (vox) { vox->bmin = tree.bmin; vox->size = tree.size; vox->material = (uint16_t)node_offset(word); vox->node = (uint32_t)tree.offset; vox->cell = CELL_NONE; }
            return 1;
        } else if (type == ORC_TWIG) {
            float leafsize = tree.size / (float)(1 << TWIG_LEVELS);
            if (twigmarch_ex(p, b, tree.bmin, tree.size, leafsize, root->twig + node_offset(word) * TWIG_WORDS,
                             eps, twigcap, glsl, s, vox, cnt)) {
                *s += t;
                if (vox) vox->node = (uint32_t)tree.offset;
                return 1;
            }
            float escape = escape_of(glsl, eps, p, b, tree.bmin, v3adds(tree.bmin, tree.size));
            t += escape + eps;
        } else {
            assert(0);
        }
    }
    return 0;
}

float orc_intersectCube(vec3 a, vec3 b, vec3 cmin, vec3 cmax, int *intersect)            /* Traverse.cpp:115-125 */
{
    vec3 tmin = v3div(v3sub(cmin, a), b);
    vec3 tmax = v3div(v3sub(cmax, a), b);
    vec3 t1 = v3min(tmin, tmax);
    vec3 t2 = v3max(tmin, tmax);
    float tnear = gmax(gmax(t1.x, t1.y), t1.z);
    float tfar = gmin(gmin(t2.x, t2.y), t2.z);
    *intersect = tfar > tnear;
    return tnear;
}

/* cubeEnterDistance, Chunkmarch.glsl:116-126: the slabs by multiplication with gamma = 1 / b, and the box must lie ahead */
static float glsl_enter(vec3 a, vec3 b, vec3 cmin, vec3 cmax, int *intersect)
{
    vec3 g = v3((float)(1.0 / (double)b.x), (float)(1.0 / (double)b.y), (float)(1.0 / (double)b.z));
    vec3 tmin = v3mul(v3sub(cmin, a), g);
    vec3 tmax = v3mul(v3sub(cmax, a), g);
    vec3 t1 = v3min(tmin, tmax);
    vec3 t2 = v3max(tmin, tmax);
    float tnear = gmax(gmax(t1.x, t1.y), t1.z);
    float tfar = gmin(gmin(t2.x, t2.y), t2.z);
    *intersect = tfar > tnear && tnear > 0;
    return tnear;
}

static int chunkmarch_core(vec3 alpha, vec3 beta, const orc_world *world, float eps, int cap, int treecap, int twigcap, int glsl,
                           float *tout, vec3 *sigma, voxel_t *vox, uint32_t *chunk_out, orc_counters *cnt)
{                                                                                        /* Traverse.cpp:127-171; rootmarch, Chunkmarch.glsl:297-330 */
    float chunksize = (float)world->chunksize;
    int ccmax[3] = { world->chunkcoordmin[0] + world->width, world->chunkcoordmin[1] + world->height, world->chunkcoordmin[2] + world->depth };
    vec3 chunkcoordmax = v3((float)ccmax[0], (float)ccmax[1], (float)ccmax[2]);
    int ics = (int)chunksize;
    vec3 chunkmin = v3((float)(world->chunkcoordmin[0] * ics), (float)(world->chunkcoordmin[1] * ics), (float)(world->chunkcoordmin[2] * ics));
    vec3 chunkmax = v3muls(chunkcoordmax, chunksize);

    float t = 0.0f;
    int intersect = 1;
    if (!orc_isInsideCube(alpha, chunkmin, chunkmax))
        t = (glsl ? glsl_enter(alpha, beta, chunkmin, chunkmax, &intersect) : orc_intersectCube(alpha, beta, chunkmin, chunkmax, &intersect)) + eps;
    if (!intersect) return 0;

    for (int c = 0; c < cap; ++c) {
        vec3 p = v3add(alpha, v3muls(beta, t));
        if (!orc_isInsideCube(p, chunkmin, chunkmax)) return 0;

        int q[3];
        orc_world_index_float(world, p, q);
        int i = orc_world_index3(world, q[0], q[1], q[2]);
        if (cnt) cnt->chunk_descs++;

        vec3 cmin = world->chunk[i].position;
        vec3 cmax = v3adds(cmin, chunksize);
        if (!glsl && !orc_isInsideCube(p, cmin, cmax)) return 0;                         /* (the shader has no such check: its treemarch just fails) */

        float s = 0;
        if (treemarch_ex(p, beta, &world->chunk[i], eps, treecap, twigcap, glsl, &s, vox, cnt)) {
            t += s;
            if (sigma) *sigma = v3add(alpha, v3muls(beta, t));
            if (tout) *tout = t;
            if (chunk_out) *chunk_out = (uint32_t)i;
            return 1;
        } else {
            float escape = escape_of(glsl, eps, p, beta, cmin, cmax);
            t += escape + eps;
        }
    }
    return 0;
}

#define DEFAULT_EPS (1.0f / 8192.0f)                                                     /* Traverse.cpp:8 */

int orc_treemarch(vec3 a, vec3 b, const orc_root *root, float *s)
{
    return treemarch_ex(a, b, root, DEFAULT_EPS, 1000, 1000, 0, s, NULL, NULL);
}

int orc_chunkmarch(vec3 alpha, vec3 beta, const orc_world *world, vec3 *sigma)
{
    return chunkmarch_core(alpha, beta, world, DEFAULT_EPS, 1000, 1000, 1000, 0, NULL, sigma, NULL, NULL, NULL);
}

/* cubeNormal, shaders/Chunkmarch.glsl:128-136 (EPS = the build's) */
static vec3 cubeNormal(vec3 s, vec3 cmin, vec3 cmax, float eps)
{
    vec3 c = v3muls(v3add(cmin, cmax), 0.5f);
    vec3 p = v3sub(s, c);
    vec3 dd = v3sub(cmin, cmax);
    vec3 d = v3muls(v3(fabsf(dd.x), fabsf(dd.y), fabsf(dd.z)), 0.5f);
    vec3 n = v3div(p, d);
    float b = 1.0f + eps;
    vec3 nb = v3muls(n, b);
    vec3 iv = v3((float)(int)nb.x, (float)(int)nb.y, (float)(int)nb.z);
    return v3normalize(iv);
}

/* The build's alternative to cubeNormal (svo_trace_params.normal_mode = 1, not in the reference): the unit vector of
 * the voxel face the sample point s = alpha + beta*(sigma - EPS) lies closest to - the axis on which |s - centre| is
 * largest (first axis on ties), signed like that component, or against the ray when the component is exactly 0.
 * s sits EPS in front of the hit, i.e. just outside the entered face, so this is the face the ray came in through;
 * unlike cubeNormal it is defined for every hit (cubeNormal's integer vector is (0,0,0) -> NaN when rounding puts s
 * inside the voxel, 14 % of the hits at depth 12). */
static vec3 faceNormal(vec3 s, vec3 cmin, vec3 cmax, vec3 beta)
{
    vec3 c = v3muls(v3add(cmin, cmax), 0.5f);
    vec3 p = v3sub(s, c);
    float ax = fabsf(p.x), ay = fabsf(p.y), az = fabsf(p.z);
    int k = 0; float pk = p.x, bk = beta.x, ak = ax;
    if (ay > ak) { k = 1; pk = p.y; bk = beta.y; ak = ay; }
    if (az > ak) { k = 2; pk = p.z; bk = beta.z; ak = az; }
    float sgn = pk > 0.0f ? 1.0f : pk < 0.0f ? -1.0f : (bk > 0.0f ? -1.0f : 1.0f);
    return v3(k == 0 ? sgn : 0.0f, k == 1 ? sgn : 0.0f, k == 2 ? sgn : 0.0f);
}

static int params_resolve(const orc_params *prm, float *eps, int *cc, int *tc, int *wc)
{
    const int glsl = prm && prm->semantics == 1;                                          /* defaults: Traverse.cpp:8,54,79,142 / Chunkmarch.glsl:1-3,17 */
    *eps = (prm && prm->eps != 0.0f) ? prm->eps : (glsl ? 1.0f / 4096.0f : DEFAULT_EPS);
    *cc = (prm && prm->max_chunk_steps > 0) ? prm->max_chunk_steps : (glsl ? 256 : 1000);
    *tc = (prm && prm->max_tree_steps > 0) ? prm->max_tree_steps : (glsl ? 512 : 1000);
    *wc = (prm && prm->max_twig_steps > 0) ? prm->max_twig_steps : (glsl ? 64 : 1000);
    return glsl;
}

int orc_chunkmarch_ex(vec3 alpha, vec3 beta, const orc_world *world, const orc_params *prm, orc_hit *hit, orc_counters *cnt)
{
    float eps; int cc, tc, wc;
    const int glsl = params_resolve(prm, &eps, &cc, &tc, &wc);
    voxel_t vox; memset(&vox, 0, sizeof vox);
    float t = 0; uint32_t chunk = 0;
    memset(hit, 0, sizeof *hit);
    if (!chunkmarch_core(alpha, beta, world, eps, cc, tc, wc, glsl, &t, NULL, &vox, &chunk, cnt)) return 0;
    hit->t = t;
    /* World.Fragment.glsl:171-175: point = alpha + beta * (sigma - EPS); normal = cubeNormal(point, leafmin, leafmax) */
    vec3 point = v3add(alpha, v3muls(beta, t - eps));
    const int face = prm && prm->normal_mode == 1;
    vec3 n = face ? faceNormal(point, vox.bmin, v3adds(vox.bmin, vox.size), beta)
                  : cubeNormal(point, vox.bmin, v3adds(vox.bmin, vox.size), eps);
    hit->normal[0] = n.x; hit->normal[1] = n.y; hit->normal[2] = n.z;
    hit->material = vox.material;
    hit->flags = HIT_FLAG | (face ? FACE_NORMAL : 0);
    hit->chunk = chunk; hit->node = vox.node; hit->cell = vox.cell;
    return 1;
}

/* one pixel: primary ray, then one shadow ray from the backed-off hit point
 * toward normalize(-light_dir). Returns rays marched (1 or 2). */
static unsigned trace_one(const orc_world *w, vec3 o, vec3 d, const orc_params *prm, orc_hit *out, orc_counters *cnt)
{
    unsigned rays = 1;
    if (orc_chunkmarch_ex(o, d, w, prm, out, cnt) && prm && prm->shadow) {
        float eps; int cc, tc, wc;
        params_resolve(prm, &eps, &cc, &tc, &wc);
        vec3 point = v3add(o, v3muls(d, out->t - eps));
        vec3 l = v3(-prm->light_dir[0], -prm->light_dir[1], -prm->light_dir[2]);
        vec3 sd = v3normalize(l);
        orc_hit sh;
        int occluded = orc_chunkmarch_ex(point, sd, w, prm, &sh, cnt);
        out->flags |= SHADOW_TRACED | (occluded ? SHADOWED : 0);
        rays = 2;
    }
    return rays;
}

void orc_camera_ray(const orc_camera *cam, int px, int py, vec3 *origin, vec3 *dir)
{
    float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    float u = ((fx / (float)cam->width) * 2.0f - 1.0f) * cam->tan_half_x;
    float v = (1.0f - (fy / (float)cam->height) * 2.0f) * cam->tan_half_y;
    vec3 f = v3(cam->forward[0], cam->forward[1], cam->forward[2]);
    vec3 r = v3(cam->right[0], cam->right[1], cam->right[2]);
    vec3 up = v3(cam->up[0], cam->up[1], cam->up[2]);
    vec3 d = v3add(v3add(f, v3muls(r, u)), v3muls(up, v));
    *dir = v3normalize(d);
    *origin = v3(cam->eye[0], cam->eye[1], cam->eye[2]);
}

typedef struct {
    const orc_world *w; const orc_params *prm; const orc_camera *cam;
    const float *origins, *dirs; orc_hit *out; orc_counters *cnt;
    int64_t n; int x0, y0, rw, rh; int tid, nthreads; uint64_t rays;
} job_t;

static void *job_run(void *arg)
{
    job_t *j = (job_t *)arg;
    uint64_t rays = 0;
    if (j->cam) {
        for (int y = j->tid; y < j->rh; y += j->nthreads)
            for (int x = 0; x < j->rw; ++x) {
                int64_t k = (int64_t)y * j->rw + x;
                vec3 o, d;
                orc_camera_ray(j->cam, j->x0 + x, j->y0 + y, &o, &d);
                if (j->cnt) memset(&j->cnt[k], 0, sizeof(orc_counters));
                rays += trace_one(j->w, o, d, j->prm, &j->out[k], j->cnt ? &j->cnt[k] : NULL);
            }
    } else {
        for (int64_t k = j->tid; k < j->n; k += j->nthreads) {
            vec3 o = v3(j->origins[3 * k], j->origins[3 * k + 1], j->origins[3 * k + 2]);
            vec3 d = v3(j->dirs[3 * k], j->dirs[3 * k + 1], j->dirs[3 * k + 2]);
            if (j->cnt) memset(&j->cnt[k], 0, sizeof(orc_counters));
            rays += trace_one(j->w, o, d, j->prm, &j->out[k], j->cnt ? &j->cnt[k] : NULL);
        }
    }
    j->rays = rays;
    return NULL;
}

static uint64_t run_jobs(job_t proto, int threads)
{
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    job_t jobs[256]; pthread_t th[256];
    for (int i = 0; i < threads; ++i) { jobs[i] = proto; jobs[i].tid = i; jobs[i].nthreads = threads; }
    if (threads == 1) { job_run(&jobs[0]); return jobs[0].rays; }
    for (int i = 0; i < threads; ++i) pthread_create(&th[i], NULL, job_run, &jobs[i]);
    uint64_t rays = 0;
    for (int i = 0; i < threads; ++i) { pthread_join(th[i], NULL); rays += jobs[i].rays; }
    return rays;
}

uint64_t orc_trace_rays(const orc_world *w, const float *origins, const float *dirs, int64_t n,
                        const orc_params *prm, orc_hit *out, orc_counters *cnt, int threads)
{
    job_t j; memset(&j, 0, sizeof j);
    j.w = w; j.prm = prm; j.origins = origins; j.dirs = dirs; j.n = n; j.out = out; j.cnt = cnt;
    return run_jobs(j, threads);
}

uint64_t orc_trace_image(const orc_world *w, const orc_camera *cam, const orc_params *prm,
                         int x0, int y0, int rw, int rh, orc_hit *out, orc_counters *cnt, int threads)
{
    job_t j; memset(&j, 0, sizeof j);
    j.w = w; j.prm = prm; j.cam = cam; j.x0 = x0; j.y0 = y0; j.rw = rw; j.rh = rh; j.out = out; j.cnt = cnt;
    return run_jobs(j, threads);
}

/* ================================================================ shading stage ============ */
/* shaders/World.Fragment.glsl:75-138 (attenuation, three Blinn-Phong lights) and :180-197 (main), with the
 * albedo taken from the material table (the texture atlas is not part of the reference repository). */
static float v3length(vec3 v) { return sqrtf(v3dot(v, v)); }
static vec3 v3neg(vec3 v) { return v3(-v.x, -v.y, -v.z); }
static vec3 v3pow(vec3 v, float e) { return v3(powf(v.x, e), powf(v.y, e), powf(v.z, e)); }
static vec3 f3(const float *p) { return v3(p[0], p[1], p[2]); }
static float attenuation(float kc, float kl, float kq, float d) { return 1.0f / (kc + kl * d + kq * d * d); }
static float max0(float x) { return gmax(x, 0.0f); }

void orc_shade_image(const orc_camera *cam, const orc_shade_params *P, int x0, int y0, int w, int h,
                     const orc_hit *gbuffer, float *rgba)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const int64_t k = (int64_t)y * w + x;
            const orc_hit *r = &gbuffer[k];
            float *out = rgba + 4 * k;
            if (!(r->flags & HIT_FLAG)) { out[0] = out[1] = out[2] = 0.0f; out[3] = 1.0f; continue; }
            vec3 eye, beta;
            orc_camera_ray(cam, x0 + x, y0 + y, &eye, &beta);
            const vec3 p = v3add(eye, v3muls(beta, r->t - P->eps));
            const vec3 n = v3(r->normal[0], r->normal[1], r->normal[2]);
            const orc_material *M = &P->materials[r->material < 8 ? r->material : 0];
            const vec3 diffuse = v3pow(f3(M->diffuse), P->gamma), specular = v3pow(f3(M->specular), P->gamma);
            const float lit = (r->flags & SHADOWED) ? 0.0f : 1.0f;
            const vec3 vdir = v3normalize(v3sub(eye, p));
            vec3 color = v3(0, 0, 0);
            {
                const vec3 l = v3normalize(v3sub(f3(P->point.position), p));
                const vec3 hv = v3normalize(v3add(l, vdir));
                const float d = max0(v3dot(n, l));
                const float s = powf(max0(v3dot(vdir, hv)), M->shininess);
                const float att = attenuation(P->point.constant, P->point.linear, P->point.quadratic, v3length(v3sub(p, f3(P->point.position))));
                const vec3 amb = v3mul(f3(P->point.ambient), diffuse);
                const vec3 dif = v3muls(v3mul(v3muls(f3(P->point.diffuse), d), diffuse), lit);
                const vec3 spe = v3muls(v3mul(v3muls(f3(P->point.specular), s), specular), lit);
                color = v3add(color, v3muls(v3add(v3add(amb, dif), spe), att));
            }
            {
                const vec3 l = v3normalize(v3neg(f3(P->directional.direction)));
                const vec3 hv = v3normalize(v3add(l, vdir));
                const float d = max0(v3dot(n, l));
                const float s = powf(max0(v3dot(vdir, hv)), M->shininess);
                const vec3 amb = v3mul(f3(P->directional.ambient), diffuse);
                const vec3 dif = v3muls(v3mul(v3muls(f3(P->directional.diffuse), d), diffuse), lit);
                const vec3 spe = v3muls(v3mul(v3muls(f3(P->directional.specular), s), specular), lit);
                color = v3add(color, v3add(v3add(amb, dif), spe));
            }
            {
                const vec3 l = v3normalize(v3sub(f3(P->spot.position), p));
                const vec3 hv = v3normalize(v3add(l, vdir));
                const float d = max0(v3dot(n, l));
                const float s = powf(max0(v3dot(vdir, hv)), M->shininess);
                const float att = attenuation(P->spot.constant, P->spot.linear, P->spot.quadratic, v3length(v3sub(p, f3(P->spot.position))));
                const float theta = v3dot(l, v3normalize(v3neg(f3(P->spot.direction))));
                const float delta = P->spot.cos_phi - P->spot.cos_gamma;
                float intensity = gmin(gmax((theta - P->spot.cos_gamma) / delta, 0.0f), 1.0f);
                const vec3 amb = v3mul(f3(P->spot.ambient), diffuse);
                const vec3 dif = v3muls(v3mul(v3muls(f3(P->spot.diffuse), d), diffuse), lit);
                const vec3 spe = v3muls(v3mul(v3muls(f3(P->spot.specular), s), specular), lit);
                color = v3add(color, v3muls(v3add(amb, v3muls(v3add(dif, spe), intensity)), att));
            }
            const float inv_z = 1.0f / v3length(v3sub(p, eye)), inv_near = 1.0f / P->near_plane, inv_far = 1.0f / P->far_plane;
            out[0] = color.x; out[1] = color.y; out[2] = color.z;
            out[3] = (inv_z - inv_near) / (inv_far - inv_near);
        }
}
