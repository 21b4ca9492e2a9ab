// world.h — the svo_world handle behind include/svo.h.
#pragma once
#include <string>
#include <vector>
#include "../../include/svo.h"
#include "svo_format.h"
#include "terrain.h"

struct svo_world {
    // World (src/World.h:44-57)
    int width = 0, height = 0, depth = 0, chunksize = 0;
    int chunkcoordmin[3] = { 0, 0, 0 };
    std::vector<svo::ChunkPools> chunks;          // World::index() order
    svo::TerrainParams terrain;                   // generator parameters (svo_world_generate), for svo_world_shift
    bool has_terrain = false;

    // geometry class
    bool exact_geometry = false;                  // every voxel corner is an exact float
    int  max_levels = 0;                          // max over chunks of depth - TWIG_LEVELS

    // device residency
    int device = -1;
    svo::DevChunk *d_chunks = nullptr;
    uint32_t *d_tree = nullptr;
    uint16_t *d_twig = nullptr;
    uint64_t *d_mask = nullptr;
    uint16_t *d_bmat = nullptr;                   // per brick: its one material / 0 (empty) / 0xFFFF (several), written with the masks
    uint32_t *d_wide = nullptr, *d_wbase = nullptr; // wide tree of every chunk (wide_tree.hip.h) and, per wide node, the reference blocks it expands
    svo::DevWide *d_wchunks = nullptr;
    std::vector<svo::DevWide> wtable;             // host mirror of d_wchunks
    std::vector<uint64_t> wide_slot;              // capacity of each chunk's wide slot (wide nodes)
    uint64_t wide_pool_len = 0, wide_pool_cap = 0, wide_nodes_used = 0;
    uint32_t *d_wscratch = nullptr;               // builder scratch: fronts, flags, ranks
    uint32_t *h_wide_tail = nullptr;              // pinned: the wide builder's per-level read-back
    uint64_t wscratch_words = 0, wscan_words = 0;  // its size; the tail of it that is the scan's own scratch
    void *builder_ctx = nullptr;                  // builder.hip: working buffers svo_world_shift / svo_world_edit_box keep between calls
    void *d_sort = nullptr;                       // svo_tile_order scratch
    size_t sort_bytes = 0;
    void *sort_event = nullptr;                   // hipEvent_t behind the last svo_tile_order: the next one (any stream) waits for it before it reuses the scratch
    bool wide_ok = false;                         // every chunk's bricks fit the 26-bit payload
    unsigned long long *d_work = nullptr;         // WORK_SLOTS x {tile cursor, rays marched}: one slot per launch in flight
    unsigned work_next = 0, work_last = 0;        // ring cursor; slot of the most recent launch
    std::vector<void *> work_event;               // hipEvent_t per slot, recorded behind the launch that used it
    std::vector<svo::DevChunk> table;             // host mirror of d_chunks
    std::vector<uint64_t> tree_slot, twig_slot;   // capacity of each chunk's slot (nodes / bricks)
    uint64_t tree_pool_len = 0, twig_pool_len = 0;    // elements in use (incl. alignment padding)
    uint64_t tree_pool_cap = 0, twig_pool_cap = 0;    // elements allocated
    int occupancy_blocks = 0;                     // cached persistent-grid size
};

namespace svo {
constexpr unsigned WORK_SLOTS = 64;               // launches of one world that overlap freely; the 65th waits (on the device) for the 1st
void set_error(const std::string &msg);
int  validate_chunk(const ChunkPools &c, std::string &why);
bool chunk_is_exact(const ChunkPools &c, int chunksize);
void classify_world(svo_world &w);
int  release_device(svo_world &w);
// device.hip: HBM residency building blocks shared by svo_world_upload and the device-resident generator
int  plan_pools(svo_world &w);                    // slots, offsets and pool sizes from the chunks' capacities (host only)
int  alloc_pools(svo_world &w, int device);       // hipMalloc + clear of the pools planned above; sets w.device
int  launch_brick_masks(svo_world &w, uint64_t first, uint64_t count, void *stream);
int  fetch_pools(svo_world &w, int chunk);        // node words / bricks that live only on the device -> host copy of that chunk
int  build_wide_all(svo_world &w, void *stream);  // wide trees (wide_tree.hip.h) of all chunks from the node words in the tree pool
// a chunk built on the device (its pools at tree_dev / twig_dev, meta.trees_on_device nodes / meta.twigs_on_device bricks) takes
// slot `chunk` of an uploaded world: device-to-device, no host copy made
int  install_resident_chunk(svo_world &w, int chunk, const ChunkPools &meta, const uint32_t *tree_dev, const uint16_t *twig_dev);
int  rebuild_wide_chunk(svo_world &w, int chunk, void *stream);
// builder.hip: World::init on the device, pools left in HBM (the world is uploaded to `device` when this returns)
int  generate_world_resident(svo_world &w, int device);
void free_builder_context(svo_world &w);
// builder.hip: Ocroot::build / destroy / replace + World::modify on an uploaded world
int  edit_box_resident(svo_world &w, int chunk, int op, const float lo[3], const float hi[3], uint32_t material);
// builder.hip: World::shift's entering plane generated on the device the world is uploaded to
int  shift_world_resident(svo_world &w, int axis, int sign);
} // namespace svo
