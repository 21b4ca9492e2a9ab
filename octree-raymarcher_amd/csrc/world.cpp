// world.cpp — host half of the C ABI: world construction, validation, World::index helpers.
//
// Reference surface mirrored (see include/svo.h for the full map):
//   World::init          src/World.cpp:19-43     -> svo_world_generate
//   World::index(_float) src/World.cpp:276-293,323-332
//   Ocroot               src/Octree.h:56-76      -> svo_chunk_desc / ChunkPools
// The reference aborts on malformed input (assert / die(), src/Util.cpp:72-78); this library
// validates and returns an error code instead.
#include "world.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

namespace svo {

static thread_local std::string g_error;
void set_error(const std::string &msg) { g_error = msg; }

// One forward pass over the node words.  Accepts what grow() and the box edits produce
// (children blocks always lie after their parent, at indices 1+8k); rejects anything a kernel
// could run off: out-of-pool offsets, back edges (cycles), branches deeper than depth-2.
// Blocks no reachable BRANCH points at (orphans left behind by Ocroot::destroy) are ignored.
int validate_chunk(const ChunkPools &c, std::string &why)
{
    const uint64_t n = c.tree.size();
    if (n == 0) { why = "empty tree pool"; return SVO_ERR_MALFORMED_TREE; }
    if (c.depth < TWIG_LEVELS || c.depth > 30) { why = "depth out of range [2,30]"; return SVO_ERR_MALFORMED_TREE; }
    if ((n - 1) % 8 != 0) { why = "tree pool is not 1 + 8k nodes"; return SVO_ERR_MALFORMED_TREE; }
    if (!(c.size > 0.0f) || !std::isfinite(c.size)) { why = "chunk size must be positive"; return SVO_ERR_MALFORMED_TREE; }
    const uint32_t maxlevel = c.depth - TWIG_LEVELS;
    const uint64_t nbricks = c.twig_count();
    std::vector<int8_t> block_level((n - 1) / 8, (int8_t)-1);
    for (uint64_t i = 0; i < n; ++i) {
        int level = 0;
        if (i > 0) {
            level = block_level[(i - 1) / 8];
            if (level < 0) continue;                       // orphan
        }
        const uint32_t word = c.tree[i];
        const uint64_t off = node_offset(word);
        switch (node_type(word)) {
        case BRANCH:
            if ((uint32_t)level >= maxlevel) { why = "BRANCH below level depth-2"; return SVO_ERR_MALFORMED_TREE; }
            if (off <= i || off + 8 > n || (off - 1) % 8 != 0) { why = "BRANCH offset outside pool or not a forward 8-block"; return SVO_ERR_MALFORMED_TREE; }
            if (block_level[(off - 1) / 8] >= 0) { why = "child block referenced twice"; return SVO_ERR_MALFORMED_TREE; }
            block_level[(off - 1) / 8] = (int8_t)(level + 1);
            break;
        case TWIG:
            if (off >= nbricks) { why = "TWIG offset outside brick pool"; return SVO_ERR_MALFORMED_TREE; }
            break;
        default:
            break;
        }
    }
    return SVO_OK;
}

// True when every corner of every possible node / brick cell of the chunk is an exactly
// representable float, i.e. the reference's incremental bmin arithmetic (src/Traverse.cpp:39-45)
// never rounds.  Then integer cell coordinates reproduce its comparisons exactly, which is what
// the stack kernel relies on.
bool chunk_is_exact(const ChunkPools &c, int chunksize)
{
    if (c.size != (float)chunksize) return false;
    int e = 0;
    if (std::frexp(c.size, &e) != 0.5f) return false;             // power-of-two edge
    const float voxel = std::ldexp(c.size, -(int)c.depth);
    if (!(voxel > 0.0f) || !std::isnormal(voxel)) return false;
    for (int a = 0; a < 3; ++a) {
        const float lo = c.position[a], hi = c.position[a] + c.size;
        if (!std::isfinite(lo) || !std::isfinite(hi)) return false;
        const double k = (double)lo / (double)voxel;
        if (k != std::floor(k)) return false;                      // position on the voxel lattice
        const double m = std::fmax(std::fabs((double)lo), std::fabs((double)hi)) / (double)voxel;
        if (m > 16777216.0) return false;                          // 2^24 lattice steps
    }
    return true;
}

void classify_world(svo_world &w)
{
    w.exact_geometry = true;
    w.max_levels = 0;
    for (const ChunkPools &c : w.chunks) {
        if (!chunk_is_exact(c, w.chunksize)) w.exact_geometry = false;
        w.max_levels = std::max(w.max_levels, (int)(c.depth - TWIG_LEVELS));
    }
}

static int positive_mod(int n, int m) { return (m + (n % m)) % m; }

} // namespace svo

using namespace svo;

extern "C" {

int svo_abi_version(void) { return SVO_ABI_VERSION; }
const char *svo_last_error(void) { return g_error.c_str(); }

int svo_world_generate(int w, int h, int d, int chunksize, const int ccm[3],
                       const svo_terrain_params *tp, svo_world **out)
{
    if (!out || !tp || w <= 0 || h <= 0 || d <= 0 || chunksize <= 0) { set_error("svo_world_generate: bad argument"); return SVO_ERR_INVALID_ARG; }
    if (tp->depth < TWIG_LEVELS || tp->depth > 20) { set_error("svo_world_generate: depth must be in [2,20]"); return SVO_ERR_INVALID_ARG; }
    if (tp->coarse_depth != 0 && (tp->coarse_depth < TWIG_LEVELS || tp->coarse_depth >= tp->depth)) { set_error("svo_world_generate: coarse_depth must be in [2, depth)"); return SVO_ERR_INVALID_ARG; }
    const uint32_t res = tp->pyramid_resolution ? tp->pyramid_resolution : (1u << tp->depth);
    if (res & (res - 1)) { set_error("svo_world_generate: pyramid_resolution must be a power of two"); return SVO_ERR_INVALID_ARG; }
    int status = SVO_OK;            // (SVO_OK_LITERAL_ONLY: generated on the device, resident there, but without the stack kernel's wide trees)
    try {
        svo_world *world = new svo_world();
        world->width = w; world->height = h; world->depth = d; world->chunksize = chunksize;
        for (int i = 0; i < 3; ++i) world->chunkcoordmin[i] = ccm ? ccm[i] : 0;
        TerrainParams p;
        p.depth = tp->depth; p.pyramid_resolution = tp->pyramid_resolution;
        p.amplitude = tp->amplitude; p.yshift = tp->yshift; p.seed = tp->seed;
        p.water = tp->water; p.water_level = tp->water_level; p.water_material = tp->water_material;
        p.threads = tp->threads;
        p.coarse_depth = tp->coarse_depth;
        for (int i = 0; i < 3; ++i) { p.refine_min[i] = tp->refine_min[i]; p.refine_max[i] = tp->refine_max[i]; }
        if (tp->build_device_plus1 > 0) {
            world->terrain = p;
            status = generate_world_resident(*world, tp->build_device_plus1 - 1);
            if (status < 0) { svo_world_destroy(world); return status; }
        } else {
            if (generate_world(w, h, d, chunksize, world->chunkcoordmin, p, world->chunks) != 0) {
                delete world;
                set_error("svo_world_generate: out of host memory in a generator thread");
                return SVO_ERR_OUT_OF_MEMORY;
            }
        }
        world->terrain = p;
        world->has_terrain = true;
        classify_world(*world);
        *out = world;
        return status;
    } catch (const std::bad_alloc &) {
        set_error("svo_world_generate: out of host memory");
        return SVO_ERR_OUT_OF_MEMORY;
    }
}

int svo_world_create(const svo_chunk_desc *chunks, int n, int w, int h, int d, int chunksize,
                     const int ccm[3], svo_world **out)
{
    if (!out || !chunks || w <= 0 || h <= 0 || d <= 0 || chunksize <= 0 || n != w * h * d) {
        set_error("svo_world_create: bad argument (n must equal w*h*d)");
        return SVO_ERR_INVALID_ARG;
    }
    try {
        svo_world *world = new svo_world();
        world->width = w; world->height = h; world->depth = d; world->chunksize = chunksize;
        for (int i = 0; i < 3; ++i) world->chunkcoordmin[i] = ccm ? ccm[i] : 0;
        world->chunks.resize((size_t)n);
        for (int i = 0; i < n; ++i) {
            const svo_chunk_desc &s = chunks[i];
            ChunkPools &c = world->chunks[(size_t)i];
            if (!s.tree || s.trees == 0 || (s.twigs && !s.twig)) { delete world; set_error("svo_world_create: chunk has no pools"); return SVO_ERR_INVALID_ARG; }
            std::memcpy(c.position, s.position, sizeof c.position);
            c.size = s.size; c.depth = s.depth;
            c.tree.assign(s.tree, s.tree + s.trees);
            c.twig.assign(s.twig, s.twig + s.twigs * TWIG_WORDS);
            while (c.tree_capacity <= c.tree.size() + 8) c.tree_capacity *= 2;
            while (c.twig_capacity < c.twig_count()) c.twig_capacity *= 2;
            std::string why;
            const int rc = validate_chunk(c, why);
            if (rc != SVO_OK) { delete world; set_error("svo_world_create: chunk " + std::to_string(i) + ": " + why); return rc; }
        }
        classify_world(*world);
        *out = world;
        return SVO_OK;
    } catch (const std::bad_alloc &) {
        set_error("svo_world_create: out of host memory");
        return SVO_ERR_OUT_OF_MEMORY;
    }
}

// World::shift, src/World.cpp:334-378: slide the grid one chunk along one axis.  The plane of chunks entering
// the grid is generated (g_pyramid + g_chunk) and stored at its toroidal index — where the plane leaving on the
// opposite side used to live — then chunkcoordmin moves.  Device copies are refreshed through svo_world_update.
int svo_world_shift(svo_world *w, const int offset[3])
{
    if (!w || !offset) return SVO_ERR_INVALID_ARG;
    if (!w->has_terrain) { set_error("svo_world_shift: world was not made by svo_world_generate (no terrain parameters)"); return SVO_ERR_UNSUPPORTED; }
    int axis = -1;
    for (int a = 0; a < 3; ++a)
        if (offset[a] != 0) { if (axis >= 0 || (offset[a] != 1 && offset[a] != -1)) { set_error("svo_world_shift: offset must be a unit axis step"); return SVO_ERR_INVALID_ARG; } axis = a; }
    if (axis < 0) { set_error("svo_world_shift: offset must be a unit axis step"); return SVO_ERR_INVALID_ARG; }
    const int sign = offset[axis];
    // an uploaded world: the entering plane is generated where the pools live (builder.hip), nothing visits the host
    if (w->device >= 0) return shift_world_resident(*w, axis, sign);
    const int dims[3] = { w->width, w->height, w->depth };
    const int u = sign < 0 ? w->chunkcoordmin[axis] - 1 : w->chunkcoordmin[axis] + dims[axis];
    const TerrainParams &tp = w->terrain;
    const uint32_t res = tp.pyramid_resolution ? tp.pyramid_resolution : (1u << tp.depth);
    try {
        // entering plane: all chunk coordinates with coordinate[axis] == u; columns (cx, cz) share a pyramid
        int lo[3], hi[3];
        for (int a = 0; a < 3; ++a) { lo[a] = w->chunkcoordmin[a]; hi[a] = w->chunkcoordmin[a] + dims[a]; }
        lo[axis] = u; hi[axis] = u + 1;
        HeightPyramid pyr;
        for (int cz = lo[2]; cz < hi[2]; ++cz)
            for (int cx = lo[0]; cx < hi[0]; ++cx) {
                pyr.build(res, tp.amplitude, 1.0f / (float)res, (float)cx * (float)res + (float)tp.seed, tp.yshift, (float)cz * (float)res + (float)tp.seed);
                for (int cy = lo[1]; cy < hi[1]; ++cy) {
                    ChunkPools c;
                    const float pos[3] = { (float)cx * (float)w->chunksize, (float)cy * (float)w->chunksize, (float)cz * (float)w->chunksize };
                    grow_chunk(c, pos, (float)w->chunksize, tp.depth, pyr, &tp);
                    if (tp.water) {
                        const float top[3] = { c.position[0] + c.size, tp.water_level, c.position[2] + c.size };
                        DirtyRange a, b;
                        fill_box(c, c.position, top, (uint16_t)tp.water_material, a, b);
                    }
                    svo_chunk_desc d;
                    std::memcpy(d.position, c.position, sizeof d.position);
                    d.size = c.size; d.depth = c.depth; d._pad = 0;
                    d.tree = c.tree.data(); d.trees = c.tree.size();
                    d.twig = c.twig.data(); d.twigs = c.twig_count();
                    const int idx = svo_world_index(w, cx, cy, cz);
                    const int rc = svo_world_update(w, idx, &d, 0, d.trees, 0, d.twigs, 1);
                    if (rc < 0) return rc;
                }
            }
    } catch (const std::bad_alloc &) {
        set_error("svo_world_shift: out of host memory");
        return SVO_ERR_OUT_OF_MEMORY;
    }
    w->chunkcoordmin[axis] += sign;
    return SVO_OK;
}

int svo_world_edit_box(svo_world *w, int chunk, int op, const float lo[3], const float hi[3], uint16_t material)
{
    if (!w || !lo || !hi || chunk < 0 || chunk >= (int)w->chunks.size() || op < SVO_EDIT_BUILD || op > SVO_EDIT_REPLACE) { set_error("svo_world_edit_box: bad argument"); return SVO_ERR_INVALID_ARG; }
    for (int a = 0; a < 3; ++a) if (!(lo[a] <= hi[a])) { set_error("svo_world_edit_box: lo must not exceed hi (and neither may be NaN)"); return SVO_ERR_INVALID_ARG; }
    if (w->device < 0) { set_error("svo_world_edit_box: the world is not uploaded (edit the host pools and pass them to svo_world_update instead)"); return SVO_ERR_NOT_UPLOADED; }
    return edit_box_resident(*w, chunk, op, lo, hi, material);
}

int svo_world_info_get(const svo_world *w, svo_world_info *o)
{
    if (!w || !o) return SVO_ERR_INVALID_ARG;
    std::memset(o, 0, sizeof *o);
    o->width = w->width; o->height = w->height; o->depth = w->depth; o->chunksize = w->chunksize;
    for (int i = 0; i < 3; ++i) o->chunkcoordmin[i] = w->chunkcoordmin[i];
    o->uploaded_device = w->device;
    for (const ChunkPools &c : w->chunks) { o->total_trees += c.tree_count(); o->total_twigs += c.twig_count(); }
    o->tree_pool_bytes = w->tree_pool_cap * sizeof(uint32_t);
    o->twig_pool_bytes = w->twig_pool_cap * TWIG_WORDS * sizeof(uint16_t);
    o->mask_pool_bytes = w->twig_pool_cap * (sizeof(uint64_t) + sizeof(uint16_t));     // occupancy masks + one material per brick
    o->max_chunk_depth = w->max_levels + (int)TWIG_LEVELS;
    o->exact_geometry = w->exact_geometry ? 1 : 0;
    o->wide_pool_bytes = w->wide_pool_cap * (64 + 9) * sizeof(uint32_t);               // 64 entries + 9 reference block indices per wide node
    o->wide_nodes = w->wide_nodes_used;
    return SVO_OK;
}

int svo_world_chunk(const svo_world *w, int i, svo_chunk_desc *o)
{
    if (!w || !o || i < 0 || i >= (int)w->chunks.size()) return SVO_ERR_INVALID_ARG;
    if (w->chunks[(size_t)i].twigs_on_device || w->chunks[(size_t)i].trees_on_device) {   // built on the device: the host copy is made on first request
        const int rc = fetch_pools(*const_cast<svo_world *>(w), i);
        if (rc != SVO_OK) return rc;
    }
    const ChunkPools &c = w->chunks[(size_t)i];
    std::memcpy(o->position, c.position, sizeof o->position);
    o->size = c.size; o->depth = c.depth; o->_pad = 0;
    o->tree = c.tree.data(); o->trees = c.tree.size();
    o->twig = c.twig.data(); o->twigs = c.twig_count();
    return SVO_OK;
}

void svo_world_destroy(svo_world *w)
{
    if (!w) return;
    release_device(*w);
    delete w;
}

// ---- Ocroot::write / read, src/Octree.cpp:178-201 ----------------------------------------------
namespace {
#pragma pack(push, 1)
struct ChunkFileHeader {            // == the first 64 bytes of Ocroot (TREE_STRUCT_SIZE, src/Octree.cpp:178)
    float    position[3];           // @0
    float    size;                  // @12
    uint32_t depth;                 // @16
    uint32_t pad0;                  // @20
    uint64_t trees, twigs;          // @24, @32
    uint64_t treestoragesize, twigstoragesize;   // @40, @48
    uint8_t  modified;              // @56
    uint8_t  pad1[7];
};
#pragma pack(pop)
static_assert(sizeof(ChunkFileHeader) == 64, "Ocroot file header is 64 bytes");
}

int svo_chunk_write(const char *path, const svo_chunk_desc *c, uint64_t treestoragesize, uint64_t twigstoragesize)
{
    if (!path || !c || !c->tree || c->trees == 0 || (c->twigs && !c->twig)) { set_error("svo_chunk_write: bad argument"); return SVO_ERR_INVALID_ARG; }
    ChunkFileHeader h;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.position, c->position, sizeof h.position);
    h.size = c->size; h.depth = c->depth;
    h.trees = c->trees; h.twigs = c->twigs;
    h.treestoragesize = treestoragesize > c->trees ? treestoragesize : c->trees;
    h.twigstoragesize = twigstoragesize > c->twigs ? twigstoragesize : c->twigs;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) { set_error(std::string("svo_chunk_write: cannot open ") + path); return SVO_ERR_INVALID_ARG; }
    bool ok = std::fwrite(&h, 1, sizeof h, fp) == sizeof h &&
              std::fwrite(c->tree, sizeof(uint32_t), c->trees, fp) == c->trees &&
              (c->twigs == 0 || std::fwrite(c->twig, TWIG_WORDS * sizeof(uint16_t), c->twigs, fp) == c->twigs);
    ok = (std::fclose(fp) == 0) && ok;
    if (!ok) { set_error("svo_chunk_write: short write"); return SVO_ERR_INVALID_ARG; }
    return SVO_OK;
}

int svo_chunk_read(const char *path, svo_chunk_desc *out, uint64_t *treestoragesize, uint64_t *twigstoragesize)
{
    if (!path || !out) { set_error("svo_chunk_read: bad argument"); return SVO_ERR_INVALID_ARG; }
    std::memset(out, 0, sizeof *out);
    FILE *fp = std::fopen(path, "rb");
    if (!fp) { set_error(std::string("svo_chunk_read: cannot open ") + path); return SVO_ERR_INVALID_ARG; }
    ChunkFileHeader h;
    int rc = SVO_OK;
    uint32_t *tree = nullptr;
    uint16_t *twig = nullptr;
    do {
        if (std::fread(&h, 1, sizeof h, fp) != sizeof h) { set_error("svo_chunk_read: truncated header"); rc = SVO_ERR_MALFORMED_TREE; break; }
        std::fseek(fp, 0, SEEK_END);
        const long long len = std::ftell(fp);
        std::fseek(fp, (long)sizeof h, SEEK_SET);
        // the reference trusts the header (src/Octree.cpp:189-201); here sizes are checked against the file
        if (h.trees == 0 || h.trees > (1ull << 31) || h.twigs > (1ull << 31) ||
            (unsigned long long)len != sizeof h + h.trees * 4ull + h.twigs * 128ull) { set_error("svo_chunk_read: header does not match file length"); rc = SVO_ERR_MALFORMED_TREE; break; }
        tree = (uint32_t *)std::malloc(h.trees * sizeof(uint32_t));
        twig = (uint16_t *)std::malloc((h.twigs ? h.twigs : 1) * TWIG_WORDS * sizeof(uint16_t));
        if (!tree || !twig) { set_error("svo_chunk_read: out of memory"); rc = SVO_ERR_OUT_OF_MEMORY; break; }
        if (std::fread(tree, sizeof(uint32_t), h.trees, fp) != h.trees ||
            (h.twigs && std::fread(twig, TWIG_WORDS * sizeof(uint16_t), h.twigs, fp) != h.twigs)) { set_error("svo_chunk_read: short read"); rc = SVO_ERR_MALFORMED_TREE; break; }
    } while (0);
    std::fclose(fp);
    if (rc != SVO_OK) { std::free(tree); std::free(twig); return rc; }
    std::memcpy(out->position, h.position, sizeof out->position);
    out->size = h.size; out->depth = h.depth;
    out->tree = tree; out->trees = h.trees;
    out->twig = twig; out->twigs = h.twigs;
    if (treestoragesize) *treestoragesize = h.treestoragesize;
    if (twigstoragesize) *twigstoragesize = h.twigstoragesize;
    return SVO_OK;
}

void svo_chunk_free(svo_chunk_desc *c)
{
    if (!c) return;
    std::free(const_cast<uint32_t *>(c->tree));
    std::free(const_cast<uint16_t *>(c->twig));
    c->tree = nullptr; c->twig = nullptr; c->trees = c->twigs = 0;
}

int svo_world_index_float(const svo_world *w, const float p[3], int q[3])
{   // src/World.cpp:323-332
    if (!w || !p || !q) return SVO_ERR_INVALID_ARG;
    const float cs = (float)w->chunksize;
    for (int i = 0; i < 3; ++i) {
        float f = p[i] / cs;
        if (f < 0.0f) f -= 1.0f;
        q[i] = (int)f;
    }
    return SVO_OK;
}

int svo_world_index(const svo_world *w, int x, int y, int z)
{   // src/World.cpp:288-293
    if (!w) return SVO_ERR_INVALID_ARG;
    return positive_mod(y, w->height) * w->width * w->depth + positive_mod(z, w->depth) * w->width + positive_mod(x, w->width);
}

} // extern "C"
