/*
 * svo_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CPU algorithm for the SVO march hot path
 * (jfjell/Octree-Raymarcher @ 2024_10_08).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; nothing under octree-raymarcher_amd/ links or calls it.
 *
 * PARITY UNPINNED: the reference holds no tests, golden vectors or fixtures for this path
 *  and its sources need GLM, which is neither vendored nor installed here, so the
 * reference itself cannot be compiled in this image (no oracle/_ref).  This restatement is pinned
 * only by hand-derived known-answer cases (tests/test_oracle_known_answers.py).
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef SVO_ORACLE_H
#define SVO_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } orc_vec3;

/* Ocroot, src/Octree.h:56-76 */
typedef struct orc_root {
    orc_vec3  position;
    float     size;
    uint32_t  depth;
    uint64_t  trees, twigs;
    uint64_t  treestoragesize, twigstoragesize;
    uint32_t *tree;               /* node words, src/Octree.h:16-26 */
    uint16_t *twig;               /* 64 cells per brick, src/Octree.h:35-45 */
} orc_root;

/* Ocdelta, src/Octree.h:47-54 */
typedef struct orc_delta { uint64_t left, right; int realloc_; } orc_delta;

/* BoundsPyramid, src/BoundsPyramid.h:6-22 */
typedef struct orc_pyramid {
    float  *basequad, **minquad, **maxquad;
    size_t  size, levels;
    float   amplitude, shift;
} orc_pyramid;

/* World (the fields chunkmarch reads), src/World.h:44-57 */
typedef struct orc_world {
    orc_root    *chunk;
    orc_pyramid *heightmap;
    int          width, height, depth, plane, volume, chunksize;
    int          chunkcoordmin[3];
} orc_world;

typedef struct orc_terrain {      /* constants of World::g_pyramid / g_chunk, src/World.cpp:296-321 */
    uint32_t depth;               /* TREE_MAX_DEPTH */
    uint32_t pyramid_resolution;  /* PYRAMID_RESOLUTION (0 = 2^depth) */
    float    amplitude, yshift;
    int32_t  seed;
    int32_t  water;
    float    water_level;
    uint32_t water_material;
} orc_terrain;

/* extended hit record (same fields as svo_hit in include/svo.h) */
typedef struct orc_hit {
    float    t;
    float    normal[3];
    uint16_t material;
    uint16_t flags;
    uint32_t chunk, node, cell;
} orc_hit;

typedef struct orc_counters {     /* reference restart-from-root work counts per ray */
    uint32_t node_words, brick_cells, chunk_descs, tree_steps;
} orc_counters;

typedef struct orc_params {
    float   eps;                  /* 0 -> 1/8192 */
    int32_t max_chunk_steps, max_tree_steps, max_twig_steps;   /* 0 -> 1000 */
    int32_t shadow;
    float   light_dir[3];
    int32_t normal_mode;          /* 0: cubeNormal (shaders/Chunkmarch.glsl:128-136); 1: the build's entered-face normal (never NaN) */
    int32_t semantics;            /* 0: the CPU march, src/Traverse.cpp.  1: its GLSL twin, shaders/Chunkmarch.glsl - what the reference renders
                                     with: EPS 1/4096 (:17), caps 256 / 512 / 64 (:1-3), cubeEscapeDistance returns BIGEPS = 1/16
                                     for d < EPS (:107-114), world entry needs tnear > 0 and multiplies by 1/b (:116-126), no chunk containment
                                     re-check (:297-330), a LEAF hit is s = t without the back-off (:263-268), the brick cell index multiplies by
                                     1 / leafsize (:201,212) */
} orc_params;

typedef struct orc_camera {       /* identical to svo_camera */
    float   eye[3], forward[3], right[3], up[3];
    float   tan_half_x, tan_half_y;
    int32_t width, height;
} orc_camera;

/* --- scene ------------------------------------------------------------------------------- */
float orc_simplex2(float x, float y);                                       /* glm::simplex(vec2) */
void  orc_pyramid_init(orc_pyramid *, size_t size, float ampl, float period,
                       float xshift, float yshift, float zshift);           /* BoundsPyramid.cpp:47-78 */
void  orc_pyramid_deinit(orc_pyramid *);
float orc_pyramid_min(const orc_pyramid *, float x, float z, size_t lv);    /* BoundsPyramid.cpp:136-139 */
float orc_pyramid_max(const orc_pyramid *, float x, float z, size_t lv);
void  orc_grow(orc_root *, orc_vec3 position, float size, uint32_t depth, const orc_pyramid *);  /* Octree.cpp:74-176 */
void  orc_build(orc_root *, orc_vec3 cmin, orc_vec3 cmax, uint16_t mat, orc_delta *dtree, orc_delta *dtwig);   /* Octree.cpp:432-436 */
void  orc_destroy(orc_root *, orc_vec3 cmin, orc_vec3 cmax, orc_delta *dtree, orc_delta *dtwig);               /* Octree.cpp:314-318 */
void  orc_root_free(orc_root *);

int   orc_world_init(orc_world *, int w, int h, int d, int s, const int chunkcoordmin[3],
                     const orc_terrain *);                                  /* World.cpp:19-43 */
void  orc_world_deinit(orc_world *);
int   orc_world_index3(const orc_world *, int x, int y, int z);             /* World.cpp:288-293 */
void  orc_world_index_float(const orc_world *, orc_vec3 p, int q[3]);       /* World.cpp:323-332 */

/* --- literal Traverse.h surface (src/Traverse.h:21-30) --------------------------------------- */
int   orc_isInsideCube(orc_vec3 p, orc_vec3 cmin, orc_vec3 cmax);
float orc_cubeEscapeDistance(orc_vec3 a, orc_vec3 b, orc_vec3 cmin, orc_vec3 cmax);
float orc_intersectCube(orc_vec3 a, orc_vec3 b, orc_vec3 cmin, orc_vec3 cmax, int *intersect);
int   orc_treemarch(orc_vec3 a, orc_vec3 b, const orc_root *root, float *s);
int   orc_chunkmarch(orc_vec3 alpha, orc_vec3 beta, const orc_world *world, orc_vec3 *sigma);

/* --- extended march: same control flow, also reports the hit voxel + work counters ------- */
int   orc_chunkmarch_ex(orc_vec3 alpha, orc_vec3 beta, const orc_world *world,
                        const orc_params *prm, orc_hit *hit, orc_counters *cnt);

/* ray i of the build's pinhole camera (include/svo.h svo_camera) */
void  orc_camera_ray(const orc_camera *cam, int px, int py, orc_vec3 *origin, orc_vec3 *dir);

/* primary (+shadow) over a list / an image rectangle; threads<=1 = scalar.  Returns rays marched. */
uint64_t orc_trace_rays(const orc_world *, const float *origins, const float *dirs, int64_t n,
                        const orc_params *, orc_hit *out, orc_counters *cnt, int threads);
uint64_t orc_trace_image(const orc_world *, const orc_camera *, const orc_params *,
                         int x0, int y0, int w, int h, orc_hit *out, orc_counters *cnt, int threads);

/* --- shading stage (restatement of shaders/World.Fragment.glsl:63-138,180-197; same layout as svo_shade_params) --- */
typedef struct orc_material { float ambient[3], diffuse[3], specular[3]; float shininess; } orc_material;
typedef struct orc_shade_params {
    struct { float position[3], ambient[3], diffuse[3], specular[3]; float constant, linear, quadratic; } point;
    struct { float position[3], direction[3], ambient[3], diffuse[3], specular[3]; } directional;
    struct { float position[3], direction[3], ambient[3], diffuse[3], specular[3];
             float cos_phi, cos_gamma, constant, linear, quadratic; } spot;
    orc_material materials[8];
    float eps, gamma, near_plane, far_plane;
} orc_shade_params;
void orc_shade_image(const orc_camera *cam, const orc_shade_params *p, int x0, int y0, int w, int h,
                     const orc_hit *gbuffer, float *rgba);

#ifdef __cplusplus
}
#endif
#endif
