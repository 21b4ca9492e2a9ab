/*
 * svo.h — C ABI of the MI355X-native sparse-voxel-octree ray traverser.
 *
 * This is the drop-in boundary for the ONE hot path of jfjell/Octree-Raymarcher that this
 * repository rebuilds: the per-pixel SVO march.  Every entry point names the reference
 * interface it replaces (paths relative to the reference checkout):
 *
 *   svo_world_generate   <- World::init / g_pyramid / g_chunk      src/World.cpp:19-43,296-321
 *                           grow()                                 src/Octree.cpp:74-176
 *                           BoundsPyramid::init                    src/BoundsPyramid.cpp:47-78
 *                           Ocroot::build (water plane)            src/Octree.cpp:319-436
 *   svo_world_create     <- a World whose chunk[] the caller already owns (Ocroot, src/Octree.h:56-76)
 *   svo_world_upload     <- World::load_gpu + RootAllocator::alloc src/World.cpp:57-94, src/Allocator.cpp:28-35
 *   svo_world_update     <- World::modify + RootAllocator::subst   src/World.cpp:268-274, src/Allocator.cpp:37-55
 *   svo_trace            <- World::draw (+ draw_shadowmap)         src/World.cpp:162-266
 *                           fragment main                          shaders/World.Fragment.glsl:162-203
 *   svo_trace_frames     <- several World::draw calls in one launch (the frame's light and eye passes,
 *                           src/Main.cpp:190-222; stereo / split-screen views; a pipelined renderer's next frames)
 *   svo_trace_rows(_frames) <- the same over the interleaved row bands of one rank (multi-GPU partition)
 *   svo_trace_rays       <- chunkmarch over a ray list             src/Traverse.cpp:127-171
 *   svo_tile_order       <- (no counterpart: the GL rasteriser schedules fragments itself) longest-first tile order of the
 *                           next World::draw from the previous one's per-tile step counts
 *   svo_world_destroy    <- World::deinit                          src/World.cpp:129-151
 *   svo_world_index*     <- World::index / index_float             src/World.cpp:276-293,323-332
 *   svo_chunk_write/read <- Ocroot::write / Ocroot::read           src/Octree.cpp:178-201
 *   svo_world_shift      <- World::shift                           src/World.cpp:334-378
 *   svo_world_edit_box   <- Ocroot::build / destroy / replace + World::modify   src/Octree.cpp:203-443, src/World.cpp:268-274
 *                           (the caller's pattern: src/Main.cpp:340-367)
 *   svo_shade(_packed)   <- lighting of fragment main               shaders/World.Fragment.glsl:63-138,180-197
 *
 * Conventions
 *   - plain C, opaque handle, caller owns every buffer it passes in;
 *   - every function returns an int status: SVO_OK (0), a positive "done, but" status (SVO_OK_LITERAL_ONLY) or a negative
 *     svo_status (nothing was changed unless the entry point says otherwise); nothing aborts
 *     or throws across this boundary (the reference uses assert/die(), src/Util.cpp:72-78);
 *   - pointers named *_dev are DEVICE pointers (HBM of the device the world was uploaded to),
 *     all others are host pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); launches are
 *     asynchronous on that stream, no host synchronisation happens inside svo_trace*;
 *   - one handle may be used by one host thread at a time; different handles are independent;
 *   - svo_trace* launches of one world may overlap on different streams (frames in flight); each
 *     launch owns a private work-cursor slot from a 64-entry ring, and a launch that comes round
 *     to a slot still in use is ordered on the device behind that earlier launch;
 *   - svo_world_update / svo_world_shift / svo_world_edit_box / svo_world_upload are ordered behind every launch issued
 *     before them on any stream (they drain the device before touching HBM, as World::modify is
 *     ordered on the GL queue) and have completed when they return: launches issued afterwards see
 *     the new world, launches issued before saw the old one, none sees a mixture.
 *
 * Semantics are those of the reference's CPU march (src/Traverse.cpp): EPS = 1/8192, step caps
 * 1000/1000/1000, closed-box containment, restart-from-root descent.  The extra per-hit outputs
 * (voxel box -> normal, material) follow shaders/Chunkmarch.glsl:128-136,190-295 and
 * shaders/World.Fragment.glsl:162-178.  There is no CPU fallback in this library: without a
 * usable HIP device every device entry point returns SVO_ERR_NO_DEVICE.
 */
#ifndef SVO_H
#define SVO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SVO_ABI_VERSION 4           /* 2: svo_trace_params.normal_mode, SVO_FACE_NORMAL, error bit in the packed record
                                       3: svo_trace_params.tile_cost_dev / tile_order_dev, svo_tile_order
                                       4: SVO_OK_LITERAL_ONLY, svo_device_cache_trim, svo_trace_params.semantics */

typedef enum svo_status {
    SVO_OK                 =  0,
    SVO_OK_LITERAL_ONLY    =  1,   /* svo_world_upload / svo_world_generate on a device: the world IS resident;
                                      svo_world_update / svo_world_edit_box / svo_world_shift: the change HAS been applied - every
                                      later launch sees the new world - but the stack kernel's wide trees could not be rebuilt (out
                                      of device memory, mostly): SVO_KERNEL_AUTO marches with the literal kernel, SVO_KERNEL_STACK is
                                      refused, until a later update / edit / shift / upload rebuilds them.  svo_last_error() says why */
    SVO_ERR_INVALID_ARG    = -1,
    SVO_ERR_NO_DEVICE      = -2,   /* no HIP device / HIP call failed; see svo_last_error() */
    SVO_ERR_OUT_OF_MEMORY  = -3,
    SVO_ERR_MALFORMED_TREE = -4,   /* node word points outside its pool, cycles, too deep */
    SVO_ERR_NOT_UPLOADED   = -5,
    SVO_ERR_UNSUPPORTED    = -6,
    SVO_ERR_HIP            = -7
} svo_status;

/* Node word (src/Octree.h:8-26, src/Octree.cpp:38-53): type = value >> 30, offset = value & 0x3FFFFFFF. */
enum { SVO_EMPTY = 0, SVO_LEAF = 1, SVO_BRANCH = 2, SVO_TWIG = 3 };
#define SVO_TWIG_LEVELS 2           /* src/Octree.h:30-33 */
#define SVO_TWIG_SIZE   4
#define SVO_TWIG_WORDS  64          /* uint16 cells per brick, index z*16 + y*4 + x */

/* One chunk as the caller holds it on the host == the public part of Ocroot (src/Octree.h:56-76).
 * tree[0] is the root; a BRANCH's 8 children are contiguous, slot = x + 2y + 4z. */
typedef struct svo_chunk_desc {
    float           position[3];
    float           size;
    uint32_t        depth;          /* leaf voxel edge = size / 2^depth; TWIG nodes sit at level depth-2 */
    uint32_t        _pad;
    const uint32_t *tree;
    uint64_t        trees;
    const uint16_t *twig;           /* twigs * 64 cells */
    uint64_t        twigs;
} svo_chunk_desc;

/* Terrain parameters of World::g_pyramid / g_chunk (src/World.cpp:296-321). */
typedef struct svo_terrain_params {
    uint32_t depth;                 /* TREE_MAX_DEPTH, reference 8 */
    uint32_t pyramid_resolution;    /* PYRAMID_RESOLUTION, reference 256; 0 = 2^depth */
    float    amplitude;             /* 64 */
    float    yshift;                /* 16 */
    int32_t  seed;                  /* integer offset added to the noise x/z shift; reference = 0 */
    int32_t  water;                 /* !=0: Ocroot::build(y < water_level, water_material) */
    float    water_level;           /* 6 */
    uint32_t water_material;        /* 6 */
    int32_t  threads;               /* host threads for generation; 0 = hardware concurrency */
    /* Sparse refinement (a build extension for deep trees, BASELINE configs[4]; 0 = off = the reference):
     * a node at level coarse_depth-2 whose box does not touch [refine_min, refine_max] (world units,
     * closed) becomes a brick sampled at level coarse_depth by the rule of src/Octree.cpp:120-154 instead
     * of being subdivided down to depth-2.  Full depth only inside the refine box. */
    uint32_t coarse_depth;
    float    refine_min[3];
    float    refine_max[3];
    /* 0 = generate on host threads; k > 0 = generate on HIP device k-1 (noise, mips, the level-synchronous BFS and
     * the water fill as kernels, bit-identical pools that never visit the host). */
    int32_t  build_device_plus1;
} svo_terrain_params;

/* Pinhole camera of the build (the reference rasterises the world box and uses
 * normalize(hitpoint - eye), shaders/World.Fragment.glsl:165; one ray per pixel here).
 * dir(px,py) = normalize(forward + right*u + up*v),
 *   u = (((px+0.5)/width )*2 - 1) * tan_half_x,   v = (1 - ((py+0.5)/height)*2) * tan_half_y,
 * all in float, no FMA contraction.  The caller supplies the orthonormal basis and the tangents. */
typedef struct svo_camera {
    float   eye[3];
    float   forward[3];
    float   right[3];
    float   up[3];
    float   tan_half_x;
    float   tan_half_y;
    int32_t width;                  /* full image size the pixel coordinates refer to */
    int32_t height;
} svo_camera;

enum {                              /* svo_trace_params.kernel */
    SVO_KERNEL_AUTO    = 0,         /* fastest kernel valid for this world */
    SVO_KERNEL_LITERAL = 1,         /* one thread per ray, restart-from-root, any geometry */
    SVO_KERNEL_STACK   = 2          /* persistent waves, LDS descent stack, ballot refill: exact geometry, chunk depth <= 24
                                       (chunks of one world may differ in depth); worlds beyond 4 GiB of wide nodes get its
                                       large-world instantiation; SVO_ERR_UNSUPPORTED otherwise (AUTO falls back to LITERAL) */
};

typedef struct svo_trace_params {
    float    eps;                   /* 0 = 1/8192 (src/Traverse.cpp:8) */
    int32_t  max_chunk_steps;       /* 0 = 1000   (src/Traverse.cpp:142) */
    int32_t  max_tree_steps;        /* 0 = 1000   (src/Traverse.cpp:79) */
    int32_t  max_twig_steps;        /* 0 = 1000   (src/Traverse.cpp:54) */
    int32_t  shadow;                /* !=0: one shadow ray per primary hit toward -light_dir */
    float    light_dir[3];          /* directionalLight.direction, default normalize(1,-1,0) (src/Main.cpp:116) */
    int32_t  kernel;                /* SVO_KERNEL_* */
    int32_t  tiles_per_wave;        /* SVO_KERNEL_STACK launch shape: every persistent wave is handed at least this
                                       many 8x8-pixel tiles.  0 / 1 = as many waves as there are tiles (up to what the
                                       device keeps resident): shortest single frame.  4 suits small images with
                                       several frames in flight (waves keep refilling instead of draining) */
    uint32_t *counters_dev;         /* optional [n][4] u32 per ray: node words, brick cells, chunk
                                       descriptors, tree steps (reference restart-from-root counts);
                                       only honoured by SVO_KERNEL_LITERAL */
    int32_t  normal_mode;           /* SVO_NORMAL_CUBE (0): svo_hit.normal = the reference's cubeNormal, bit for bit - NaN where
                                       its integer vector is (0,0,0): 0.1 % of the hits at depth 8, 14 % at depth 12.
                                       SVO_NORMAL_FACE (1): the unit vector of the voxel face the sample point
                                       alpha + beta*(t - EPS) lies closest to (axis of the largest |point - centre|, first axis
                                       on ties; signed like that component, against the ray if it is exactly 0) - the face the
                                       ray entered through, defined for every hit; such records carry SVO_FACE_NORMAL */
    int32_t  launches_in_flight;    /* SVO_KERNEL_STACK launch shape: how many launches of this world the caller keeps in flight on different
                                       streams (0 / 1 = one: the launch takes every wave slot of the device - shortest single launch).
                                       With n >= 2 a launch takes 2/n of the wave slots, so that at least two launches are resident
                                       side by side and one's drain (its longest rays) runs under another's bulk instead of
                                       holding slots the next launch waits for.  Never changes the records */
    /* Frame-to-frame tile scheduling for the shortest SINGLE frame (the reference's caller issues one World::draw per
     * displayed frame, src/Main.cpp:190-222).  A frame takes as long as its bulk or its longest ray, whichever is longer;
     * handing the tiles out longest-first starts the long rays at once.  SVO_KERNEL_STACK only; both optional:
     *   tile_cost_dev   [nframes][ntiles][2] u32, written: the largest step count of a primary ray of the 8x8-pixel tile and
     *                   of a shadow ray of it (ntiles = ceil(w/8) * ceil(h/8) of the traced raster, row-major);
     *   tile_order_dev  [ntiles] u32, read: the order in which every frame's tiles are handed out (a permutation of
     *                   0..ntiles-1, e.g. svo_tile_order of the previous frame's cost: temporal coherence makes it a good
     *                   predictor).  The records written are the same with any order. */
    uint32_t       *tile_cost_dev;
    const uint32_t *tile_order_dev;
    int32_t  semantics;             /* which of the reference's two marches (SURVEY.md App. B lists their differences):
                                       SVO_SEMANTICS_CPU (0): src/Traverse.cpp - what every default above quotes;
                                       SVO_SEMANTICS_GLSL (1): shaders/Chunkmarch.glsl, the march the reference RENDERS with - eps 0 means
                                       1/4096 (:17), step caps 0 mean 256 / 512 / 64 (:1-3), cubeEscapeDistance returns BIGEPS = 1/16 for a
                                       distance below EPS (:107-114: no ray creeps along a lattice plane for thousands of steps), a ray from
                                       outside enters the world only if the box lies ahead of it (tnear > 0, slabs by multiplication with
                                       1 / dir, :116-126), no chunk containment re-check (:297-330), a LEAF hit is reported at t without the
                                       CPU code's back-off (:263-268; svo_hit.t is then the shader's sigma), brick cells are found by
                                       multiplying with 1 / leafsize (:201,212).  Both kernels, the oracle and its Python twin implement it */
    int32_t  _pad_semantics;
} svo_trace_params;
enum { SVO_NORMAL_CUBE = 0, SVO_NORMAL_FACE = 1 };
enum { SVO_SEMANTICS_CPU = 0, SVO_SEMANTICS_GLSL = 1 };

/* G-buffer record, 32 bytes per pixel / per ray. */
enum {
    SVO_HIT_FLAG      = 1u << 0,    /* primary ray hit a voxel */
    SVO_SHADOW_TRACED = 1u << 1,    /* a shadow ray was cast from this hit */
    SVO_SHADOWED      = 1u << 2,    /* ... and it hit something */
    SVO_FACE_NORMAL   = 1u << 3,    /* normal[] is the entered-face normal (svo_trace_params.normal_mode = SVO_NORMAL_FACE) */
    SVO_ERR_FLAG      = 1u << 15    /* runaway ray: given up after 2^22 march steps of the kernel's own counting (only rays that
                                       creep through all three nested loops of the reference get there; the stack kernel
                                       takes creeping stretches in closed form and finishes rays the literal kernel gives
                                       up).  A primary ray is then recorded as a miss, a shadow ray as "traced, not occluded".  WHICH rays
                                       get there depends on the kernel (each counts its own steps; the stack kernel also counts the passes a
                                       lane waits for a vote): records carrying this flag are outside the cross-kernel equality */
};
#define SVO_CELL_NONE 0xFFu         /* hit a LEAF node, not a brick cell */

typedef struct svo_hit {
    float    t;                     /* sigma distance exactly as chunkmarch accumulates it (src/Traverse.cpp:160-161) */
    float    normal[3];             /* cubeNormal of alpha + beta*(t - EPS) on the hit voxel (shaders/Chunkmarch.glsl:128-136) */
    uint16_t material;              /* LEAF offset or brick cell value */
    uint16_t flags;
    uint32_t chunk;                 /* World::index() linear chunk index */
    uint32_t node;                  /* index in that chunk's tree[] of the LEAF/TWIG node hit */
    uint32_t cell;                  /* brick cell word z*16+y*4+x, or SVO_CELL_NONE */
} svo_hit;

typedef struct svo_world svo_world;

typedef struct svo_world_info {
    int32_t  width, height, depth, chunksize;
    int32_t  chunkcoordmin[3];
    int32_t  uploaded_device;       /* -1 if not uploaded */
    uint64_t total_trees, total_twigs;
    uint64_t tree_pool_bytes, twig_pool_bytes, mask_pool_bytes;
    int32_t  max_chunk_depth;
    int32_t  exact_geometry;        /* 1: all voxel corners are exact floats -> SVO_KERNEL_STACK allowed */
    uint64_t wide_pool_bytes;       /* the stack kernel's derived view of the trees (two levels per node) + its reference node indices */
    uint64_t wide_nodes;            /* wide nodes in use (64 entries each) */
} svo_world_info;

/* ---- world construction (host) ------------------------------------------------------------ */

/* World::init(w,h,d,s): generate w*h*d chunks of Simplex terrain.  Chunk (x,y,z) sits at
 * (chunkcoordmin + (x,y,z)) * chunksize; linear index = World::index(). */
int svo_world_generate(int w, int h, int d, int chunksize, const int chunkcoordmin[3],
                       const svo_terrain_params *terrain, svo_world **out);

/* Wrap chunks the caller generated/edited itself.  `chunks` has w*h*d entries in World::index()
 * order; pools are COPIED.  Node words are validated (SVO_ERR_MALFORMED_TREE). */
int svo_world_create(const svo_chunk_desc *chunks, int n, int w, int h, int d, int chunksize,
                     const int chunkcoordmin[3], svo_world **out);

int  svo_world_info_get(const svo_world *, svo_world_info *out);
/* Borrow the host copy of chunk i (valid until the world is destroyed or chunk i is replaced: svo_world_update,
 * svo_world_edit_box, a svo_world_shift that slides it out).  A chunk built or edited on the device is fetched on the first request. */
int  svo_world_chunk(const svo_world *, int i, svo_chunk_desc *out);
void svo_world_destroy(svo_world *);

/* Ocroot::write / Ocroot::read (src/Octree.cpp:178-201): one chunk per file, the reference's raw layout —
 * a 64-byte header (position f32x3 @0, size f32 @12, depth u32 @16, trees u64 @24, twigs u64 @32,
 * treestoragesize u64 @40, twigstoragesize u64 @48, modified u8 @56; the first 64 bytes of Ocroot on x86-64),
 * then trees*4 bytes of node words, then twigs*128 bytes of bricks.  svo_chunk_read allocates *tree / *twig
 * with malloc (caller frees with svo_chunk_free) and validates sizes against the file length. */
int  svo_chunk_write(const char *path, const svo_chunk_desc *chunk, uint64_t treestoragesize, uint64_t twigstoragesize);
int  svo_chunk_read(const char *path, svo_chunk_desc *out, uint64_t *treestoragesize, uint64_t *twigstoragesize);
void svo_chunk_free(svo_chunk_desc *chunk);

/* World::index_float / World::index (src/World.cpp:323-332, 288-293). */
int  svo_world_index_float(const svo_world *, const float p[3], int q[3]);
int  svo_world_index(const svo_world *, int x, int y, int z);

/* ---- device residency -------------------------------------------------------------------- */

/* Pack every chunk's tree[] / twig[] into one flat HBM pool each (+ the 64-bit brick occupancy
 * masks derived from twig[]), build the chunk table, on HIP device `device`. */
int svo_world_upload(svo_world *, int device);

/* Replace chunk `chunk` by `desc` (edited pools) and refresh HBM: ranges [tree_left,tree_right)
 * nodes and [twig_left,twig_right) bricks are re-sent in place; realloc!=0 (or growth beyond the
 * chunk's slot) re-packs the chunk at the pool tail == Ocdelta semantics, src/Octree.h:47-54.
 * As with the reference's glBufferSubData, with realloc == 0 only the ranges (and whatever desc holds beyond the chunk's previous
 * length) are taken from desc - the library's own host copy and HBM keep the rest - so they must cover every word that changed;
 * the result is validated as a whole and a malformed one is refused with nothing changed (SVO_ERR_MALFORMED_TREE). */
int svo_world_update(svo_world *, int chunk, const svo_chunk_desc *desc,
                     uint64_t tree_left, uint64_t tree_right,
                     uint64_t twig_left, uint64_t twig_right, int realloc);

/* World::shift (src/World.cpp:334-378): slide the grid by one chunk along one axis (offset = +-1 on exactly one
 * axis).  The entering plane of chunks is generated with the world's terrain parameters and replaces, at its
 * toroidal World::index(), the plane that leaves; chunkcoordmin moves.  Only for worlds made by svo_world_generate.
 * On an uploaded world the plane is generated on the device the pools live on and installed device-to-device (host
 * copies of those chunks are made on request, svo_world_chunk); otherwise on the host. */
int svo_world_shift(svo_world *, const int offset[3]);

/* Ocroot::build / destroy / replace (src/Octree.cpp:203-443) followed by World::modify (src/World.cpp:268-274) on chunk
 * `chunk` of an UPLOADED world, run on the device the pools live on: the closed box [lo, hi] is filled with `material`
 * where the chunk is empty (SVO_EDIT_BUILD), emptied (SVO_EDIT_DESTROY), or emptied and then filled (SVO_EDIT_REPLACE).
 * Node blocks and bricks are appended in the reference's depth-first order, so the pools equal what the reference's
 * edit leaves, index for index; its storage sizes double by the reference's rule.  The box is in world coordinates and may
 * extend beyond the chunk (the reference's caller applies the same cube to every chunk it overlaps, src/Main.cpp:322-338).
 * Nothing visits the host; a host copy of the chunk is made again on request (svo_world_chunk).
 * SVO_ERR_NOT_UPLOADED on a world that is not resident. */
enum { SVO_EDIT_BUILD = 0, SVO_EDIT_DESTROY = 1, SVO_EDIT_REPLACE = 2 };
int svo_world_edit_box(svo_world *, int chunk, int op, const float lo[3], const float hi[3], uint16_t material);

/* ---- the hot path ------------------------------------------------------------------------ */

/* Trace the pixel rectangle [x0,x0+w) x [y0,y0+h) of `cam`'s image.  out_dev receives w*h
 * records, row-major within the rectangle.  Rays launched = w*h (+ one per hit when shadow). */
int svo_trace(svo_world *, const svo_camera *cam, const svo_trace_params *params,
              int x0, int y0, int w, int h, svo_hit *out_dev, void *stream);

/* Trace the horizontal bands b = band0 + k*band_stride (k = 0..nbands-1) of the full-width image;
 * band b covers rows [b*band_height, (b+1)*band_height).  This is the interleaved tile-row
 * partition used for multi-GPU (rank r of N: band0 = r, band_stride = N, band_height = 8).
 * out_dev receives nbands*band_height*cam->width records, bands stacked in k order; rows that
 * fall below the image are written as misses. */
int svo_trace_rows(svo_world *, const svo_camera *cam, const svo_trace_params *params,
                   int band0, int band_stride, int nbands, int band_height,
                   svo_hit *out_dev, void *stream);

/* Several World::draw calls (src/World.cpp:205-266; the reference issues two marches per displayed frame, the light's
 * and the eye's, src/Main.cpp:190-222) in ONE launch: nframes (1..SVO_MAX_FRAMES) cameras of one image size, the same rectangle / bands of
 * each; frame f's records follow frame f-1's in out_dev (nframes consecutive rasters).  Results are those of nframes
 * separate svo_trace / svo_trace_rows calls.  What it buys: the stack kernel's persistent waves run through all the
 * frames' tiles behind one set of cursors, so they drain once per launch instead of once per frame (stereo pairs,
 * cube-map faces, shadow cascades, several viewports, or simply the next frames of a pipelined renderer). */
#define SVO_MAX_FRAMES 16
int svo_trace_frames(svo_world *, const svo_camera *cams, int nframes, const svo_trace_params *params,
                     int x0, int y0, int w, int h, svo_hit *out_dev, void *stream);
int svo_trace_rows_frames(svo_world *, const svo_camera *cams, int nframes, const svo_trace_params *params,
                          int band0, int band_stride, int nbands, int band_height,
                          svo_hit *out_dev, void *stream);

/* chunkmarch over an explicit list: origins_dev/dirs_dev are [n][3] float on the device. */
int svo_trace_rays(svo_world *, const float *origins_dev, const float *dirs_dev, int64_t n,
                   const svo_trace_params *params, svo_hit *out_dev, void *stream);

/* order_dev[0..ntiles) = the tile indices sorted by descending cost[i][0] + cost[i][1] (a stable device sort; cost_dev as
 * svo_trace_params.tile_cost_dev of ONE frame wrote it).  Asynchronous on `stream`; calls of one world on different streams are
 * ordered behind one another on the device (they share the world's sort scratch), each call's order_dev is complete when the
 * work issued on its own stream before reaches it.  A launch ignores entries of tile_order_dev that are not tile indices (such a
 * tile is skipped, its pixels stay unwritten) rather than read out of range. */
int svo_tile_order(svo_world *, const uint32_t *cost_dev, uint32_t *order_dev, int ntiles, void *stream);

/* ---- packed G-buffer (8 bytes / pixel) for the multi-GPU gather ------------------------------------------
 * { float t; uint32 w } with w = material (bits 0-15) | flags & 0xFF (bits 16-23) | normal code (bits 24-30) |
 * SVO_ERR_FLAG (bit 31): per axis 2 bits (0: -, 1: 0, 2: +) in bits 24-29, bit 30 = NaN normal.  cubeNormal only ever yields
 * normalize(ivec3 in {-1,0,1}^3) (shaders/Chunkmarch.glsl:128-136), so t, normal, material and flags survive the
 * round trip bit for bit; the parity ids (chunk, node, cell) are not carried (unpack zeroes them). */
int svo_gbuffer_pack(const svo_hit *gbuffer_dev, uint64_t *packed_dev, int64_t n, void *stream);
int svo_gbuffer_unpack(const uint64_t *packed_dev, svo_hit *gbuffer_dev, int64_t n, void *stream);

/* ---- shading stage: Blinn-Phong x 3 lights over the G-buffer --------------------
 * shaders/World.Fragment.glsl:63-138,180-197.  The reference multiplies the lights with gamma-decoded samples of
 * its Diffuse / Specular texture atlas, which is not part of the repository; here the albedo comes from the
 * material table's diffuse / specular colours instead (pow(colour, gamma)), the shadow term from SVO_SHADOWED.
 * Output per pixel: float4 {r, g, b, depth} with depth = (1/dist - 1/near) / (1/far - 1/near) (gl_FragDepth,
 * World.Fragment.glsl:193-197); misses give {0,0,0,1}. */
typedef struct svo_material { float ambient[3], diffuse[3], specular[3]; float shininess; } svo_material;
typedef struct svo_shade_params {
    struct { float position[3], ambient[3], diffuse[3], specular[3]; float constant, linear, quadratic; } point;
    struct { float position[3], direction[3], ambient[3], diffuse[3], specular[3]; } directional;
    struct { float position[3], direction[3], ambient[3], diffuse[3], specular[3];
             float cos_phi, cos_gamma, constant, linear, quadratic; } spot;
    svo_material materials[8];      /* ML[8], World.Fragment.glsl:63-73 */
    float eps;                      /* 0 = 1/8192 */
    float gamma;                    /* 0 = 2.2 */
    float near_plane, far_plane;    /* 0 = 0.125 / 8192 (shaders/Chunkmarch.glsl:20-21) */
} svo_shade_params;

/* Fill `p` with the reference's lights (src/Main.cpp:101-131) and material table. */
void svo_shade_defaults(svo_shade_params *p);
/* Shade the rectangle a svo_trace(cam, x0, y0, w, h) call filled: gbuffer_dev has w*h records, rgba_dev w*h float4. */
int svo_shade(const svo_camera *cam, const svo_shade_params *p, int x0, int y0, int w, int h,
              const svo_hit *gbuffer_dev, float *rgba_dev, void *stream);
/* The same over the 8-byte records of svo_gbuffer_pack (what rank 0 holds after the multi-GPU gather): 8 B read + 16 B
 * written per pixel instead of 32 + 16; identical colours (the packed record carries t, normal, material, flags). */
int svo_shade_packed(const svo_camera *cam, const svo_shade_params *p, int x0, int y0, int w, int h,
                     const uint64_t *packed_dev, float *rgba_dev, void *stream);

/* Number of rays the last launch on this world actually marched (primary + shadow, all frames of a
 * svo_trace_frames launch; a multi-frame call served by a kernel other than SVO_KERNEL_STACK is one launch per
 * frame and reports its last frame); synchronises `stream` internally — call it outside timed regions. */
int svo_trace_last_ray_count(svo_world *, void *stream, uint64_t *rays);

/* ---- small device helpers so that C/C++ callers need no HIP headers --------------------- */
int   svo_device_count(void);
void *svo_device_alloc(size_t bytes);
void  svo_device_free(void *p_dev);
/* The library keeps the large device buffers of a world that is destroyed or re-packed (tree, brick, mask, material and wide
 * pools: at most 8 buffers of 1 MiB and more per process) and hands them to the next world whose pools they fit - a caller that
 * replaces its world pays no hipFree / hipMalloc of multi-GB buffers.  This returns them to the driver (also done by itself when
 * an allocation fails). */
void  svo_device_cache_trim(void);
int   svo_memcpy_h2d(void *dst_dev, const void *src, size_t bytes);
int   svo_memcpy_d2h(void *dst, const void *src_dev, size_t bytes);
int   svo_stream_synchronize(void *stream);

const char *svo_last_error(void);   /* thread-local message of the last failing call */
int         svo_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SVO_H */
