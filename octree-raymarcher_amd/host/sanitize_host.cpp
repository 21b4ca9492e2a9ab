// ASan/UBSan driver for the product's host-side generator (terrain.cpp): World::init incl. water, the sparse
// extension and the bilinear pyramid path, plus the chunk validator.  Built by tests/test_sanitizers.py with g++.
#include <cstdio>
#include <string>
#include "../csrc/terrain.h"
#include "../csrc/world.h"

int main()
{
    using namespace svo;
    size_t nodes = 0;
    for (int variant = 0; variant < 3; ++variant) {
        TerrainParams tp;
        tp.depth = variant == 2 ? 9 : 7;
        tp.pyramid_resolution = variant == 1 ? 32 : 0;           // bilinear path beyond the pyramid base
        tp.threads = 3;
        if (variant == 2) { tp.coarse_depth = 6; tp.water = 0; tp.refine_min[0] = 60; tp.refine_max[0] = 70; tp.refine_min[1] = tp.refine_min[2] = -1e9f; tp.refine_max[1] = tp.refine_max[2] = 1e9f; }
        const int ccm[3] = { -1, -1, 0 };
        std::vector<ChunkPools> chunks;
        generate_world(2, 2, 2, 128, ccm, tp, chunks);
        for (const ChunkPools &c : chunks) {
            std::string why;
            if (validate_chunk(c, why) != 0) { std::printf("invalid chunk: %s\n", why.c_str()); return 1; }
            if (!chunk_is_exact(c, 128)) { std::printf("inexact chunk\n"); return 1; }
            nodes += c.tree.size();
        }
    }
    std::printf("nodes %zu\n", nodes);
    return nodes > 1000 ? 0 : 1;
}
