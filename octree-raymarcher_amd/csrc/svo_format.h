// svo_format.h — data formats shared by the host and device halves of libsvo_amd.
//
// Host formats are the reference's (src/Octree.h:8-76): 32-bit node words (2-bit type, 30-bit
// offset), 4x4x4 uint16 bricks, BFS-ordered per-chunk pools.  The DEVICE layout is this build's:
//
//   tree pool   one flat uint32 array for all chunks.  Chunk c's node n lives at
//               tree_pool[tree_off[c] + n] with tree_off[c] % 8 == 7, so that every 8-child block
//               (node indices 1+8k .. 8+8k, src/Octree.cpp:155-174) is one aligned 32-byte
//               segment and never straddles a cache line.
//   twig pool   one flat array of 128-byte bricks (uint16[64], index z*16+y*4+x), 128-B aligned:
//               one brick == one cache line.
//   mask pool   one uint64 per brick, bit w set iff brick cell w != 0.  Derived on upload; lets
//               the brick march test cells from a register and touch the brick line only on a hit.
//   chunk table 32 bytes per chunk in World::index() order (the reference's GPUChunk is also 32 B,
//               src/World.h:16-27; its region/offset pairs become 64-bit element offsets here).
#pragma once
#include <stdint.h>

namespace svo {

enum : uint32_t { EMPTY = 0, LEAF = 1, BRANCH = 2, TWIG = 3 };
constexpr uint32_t TWIG_LEVELS = 2, TWIG_SIZE = 4, TWIG_WORDS = 64;
constexpr uint32_t OFFSET_MASK = 0x3FFFFFFFu;

constexpr uint32_t node_make(uint32_t type, uint32_t offset) { return (type << 30) | (offset & OFFSET_MASK); }
constexpr uint32_t node_type(uint32_t w) { return w >> 30; }
constexpr uint32_t node_offset(uint32_t w) { return w & OFFSET_MASK; }

// The device fill (builder.hip: Ocroot::build / destroy as three level-synchronous sweeps) keeps one word per visited node:
// its action (3 bits, FillAction) and the index of its child block in the next level's list.  A level holds fewer than
// FILL_KIDS_LIMIT child blocks (the fill refuses larger frontiers: 8 * 2^28 list entries is its 2^31 limit), so the index has
// 29 bits to itself - until round 4 it shared the word with an 8-bit action and lost its top bits from 2^24 blocks on.
constexpr uint32_t FILL_ACTION_BITS = 3;
constexpr uint64_t FILL_KIDS_LIMIT = 1ull << 28;
constexpr uint32_t fill_pack(uint32_t action, uint32_t kids) { return action | (kids << FILL_ACTION_BITS); }
constexpr uint32_t fill_action(uint32_t w) { return w & ((1u << FILL_ACTION_BITS) - 1u); }
constexpr uint32_t fill_kids(uint32_t w) { return w >> FILL_ACTION_BITS; }
static_assert(fill_kids(fill_pack(5u, (uint32_t)FILL_KIDS_LIMIT - 1u)) == FILL_KIDS_LIMIT - 1u && fill_action(fill_pack(5u, (uint32_t)FILL_KIDS_LIMIT - 1u)) == 5u,
              "the largest child-block index the fill admits survives the packing");

struct DevChunk {               // 32 bytes
    float    bmin[3];
    uint32_t levels;            // depth - TWIG_LEVELS: deepest level a node can sit at
    uint64_t tree_off;          // index of node 0 in the tree pool (uint32 units), % 8 == 7
    uint64_t twig_off;          // index of brick 0 in the twig pool / mask pool (brick units)
};
static_assert(sizeof(DevChunk) == 32, "chunk table entry is 32 bytes");

// The stack kernel's chunk table: the same 32 bytes with the chunk's wide tree (wide_tree.hip.h) in place of tree[].
struct DevWide {
    float    bmin[3];
    uint32_t levels;
    uint32_t wide_off;          // index of the chunk's top wide node in the wide pool (64 uint32 entries per wide node)
    uint32_t _pad;
    uint64_t twig_off;
};
static_assert(sizeof(DevWide) == 32, "wide chunk table entry is 32 bytes");

// Everything a trace kernel needs, passed by value.
// Launch slot (u64 words).  The stack kernel deals the image out as TILE_REGIONS screen regions, one per XCD (own L2),
// each with its own cursor; a wave whose region is empty moves on to the next one.
constexpr int TILE_REGIONS = 8;
// a tile = the 64 rays a wave picks up at once: TILE_W x TILE_H pixels of the raster
#ifndef SVO_TILE_W
#define SVO_TILE_W 8
#endif
constexpr int TILE_W = SVO_TILE_W, TILE_H = 64 / TILE_W;
static_assert(TILE_W * TILE_H == 64 && (TILE_W & (TILE_W - 1)) == 0, "a tile is one wave of rays");
constexpr uint32_t WIDE_BASE_WORDS = 9;  // per wide node (wide_tree.hip.h): reference index of the child block and of the 8 grandchild blocks
constexpr int WORK_CURSOR0 = 2;
constexpr int WORK_SLOT_WORDS = 16;     // 128 B: slots do not share a cache line

constexpr int MAX_FRAMES = 16;          // cameras one launch can march (svo_trace_frames)

// one camera of the launch (include/svo.h svo_camera without the image size, which all frames share)
struct FrameCam { float eye[3], fwd[3], right[3], up[3], tanx, tany; };

struct TraceArgs {
    // world (src/Traverse.cpp:129-133: chunkmin/chunkmax of the whole grid)
    float    worldmin[3], worldmax[3];
    float    chunksize;
    int32_t  dimw, dimh, dimd;
    const DevChunk *chunks;
    const DevWide  *wchunks;    // stack kernel: chunk table over the wide pool
    const uint32_t *wide;       // wide pool: 64 entries per wide node (wide_tree.hip.h)
    const uint32_t *wbase;      // per wide node: reference index of the child block and of the 8 grandchild blocks it expands (for svo_hit.node)
    const uint16_t *bmat;       // per brick: its one material, 0 if empty, 0xFFFF if it holds several
    const uint32_t *tree;
    const uint16_t *twig;
    const uint64_t *mask;
    // rays
    int32_t  from_camera;       // 1: generate from cam; 0: origins/dirs
    int32_t  nframes;           // camera mode: frames in this launch; frame f writes records [f*w*h, (f+1)*w*h)
    FrameCam cams[MAX_FRAMES];
    float    inv_chunksize;     // exact when chunksize is a power of two (stack kernel only)
    int32_t  ccm[3];            // chunkcoordmin
    int32_t  cbase[3];          // positive_mod(chunkcoordmin, dims): index of the grid's first chunk per axis
    int32_t  imgw, imgh;        // full image size (ray generation)
    int32_t  x0, y0, w, h;      // local rectangle: local (lx,ly) -> px = x0+lx, py = y0 + (ly/bh)*ystep + ly%bh
    int32_t  bh, ystep;
    const float *origins, *dirs;
    int64_t  n;                 // rays in list mode, w*h in camera mode
    // parameters
    float    eps;
    int32_t  cap_chunk, cap_tree, cap_twig;
    int32_t  shadow;
    int32_t  normal_mode;       // SVO_NORMAL_CUBE / SVO_NORMAL_FACE
    // which twin is marched (svo_trace_params.semantics): the shader's cubeEscapeDistance returns BIGEPS for d < EPS - guard_eps is
    // that EPS, or -inf (no distance is below it) for the CPU march; a LEAF hit is s = t - leaf_back (EPS for the CPU march, 0 for
    // the shader)
    int32_t  glsl;
    float    guard_eps, leaf_back;
    float    sdir[3];           // normalize(-light_dir), computed on the host
    // outputs
    void     *out;              // svo_hit[n]
    uint32_t *counters;         // optional [n][4]
    unsigned long long *work;   // launch slot: [1] rays marched, [WORK_CURSOR0 + r] tile cursor of screen region r
    int32_t  ntiles, tiles_per_row;     // 8x8 tiles of ONE frame's raster (64-ray groups of the list in list mode)
    int32_t  exact_geometry;    // every voxel corner is an exact float (svo_world_info.exact_geometry): the literal kernel's closed-form creep runs rely on it
    uint32_t *tile_cost;        // optional [nframes][ntiles][2]: largest primary / shadow step count per tile (stack kernel)
    const uint32_t *tile_order; // optional [ntiles]: the order in which a frame's tiles are handed out (stack kernel)
};

} // namespace svo
