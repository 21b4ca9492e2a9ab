// terrain.h — host-side chunk pools and the terrain generator of libsvo_amd.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <vector>
#include "svo_format.h"

namespace svo {

// Ocdelta (src/Octree.h:47-54): dirty node / brick index range of an edit.
struct DirtyRange {
    uint64_t left = UINT64_MAX, right = 0;
    bool realloc = false;
};

// The host copy of one chunk == Ocroot (src/Octree.h:56-76).
struct ChunkPools {
    float    position[3] = { 0, 0, 0 };
    float    size = 0;
    uint32_t depth = 0;
    uint64_t tree_capacity = 16, twig_capacity = 16;   // treestoragesize / twigstoragesize
    std::vector<uint32_t> tree;                        // node words
    std::vector<uint16_t> twig;                        // 64 cells per brick
    // > 0: the chunk was built on the device and its bricks have not been fetched to the host yet (`twig` is empty);
    // svo_world_chunk / a move to another device fetch them (device.hip: fetch_pools)
    uint64_t twigs_on_device = 0;
    uint64_t twig_count() const { return twigs_on_device ? twigs_on_device : twig.size() / TWIG_WORDS; }
    // the same for the node words (`tree` is empty until fetched): the device builder runs Ocroot::build on the device too
    uint64_t trees_on_device = 0;
    uint64_t tree_count() const { return trees_on_device ? trees_on_device : tree.size(); }
    void reserve_tree(uint64_t need);
};

// BoundsPyramid (src/BoundsPyramid.h) as one flat array per bound.
struct HeightPyramid {
    uint32_t size = 0, levels = 0;
    float amplitude = 0, shift = 0;
    std::vector<float> lo, hi;                         // level lv at level_offset(lv), (2^lv)^2 entries
    static size_t level_offset(uint32_t lv) { return (((size_t)1 << (2 * lv)) - 1) / 3; }
    void  build(uint32_t res, float ampl, float period, float xshift, float yshift, float zshift);
    float bound(const std::vector<float> &q, float x, float z, uint32_t lv) const;
    float min(float x, float z, uint32_t lv) const { return bound(lo, x, z, lv); }
    float max(float x, float z, uint32_t lv) const { return bound(hi, x, z, lv); }
};

struct TerrainParams {
    uint32_t depth = 8, pyramid_resolution = 0;
    float amplitude = 64.0f, yshift = 16.0f;
    int32_t seed = 0, water = 1;
    float water_level = 6.0f;
    uint32_t water_material = 6;
    int32_t threads = 0;
    uint32_t coarse_depth = 0;                         // sparse refinement, see include/svo.h
    float refine_min[3] = { 0, 0, 0 }, refine_max[3] = { 0, 0, 0 };
};

float simplex2(float x, float y);
void  grow_chunk(ChunkPools &c, const float position[3], float size, uint32_t depth, const HeightPyramid &pyr,
                 const TerrainParams *sparse = nullptr);
void  fill_box(ChunkPools &c, const float lo[3], const float hi[3], uint16_t material, DirtyRange &dtree, DirtyRange &dtwig);
int   generate_world(int w, int h, int d, int chunksize, const int chunkcoordmin[3], const TerrainParams &tp,
                     std::vector<ChunkPools> &chunks);


} // namespace svo
