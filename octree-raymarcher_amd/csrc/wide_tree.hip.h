// wide_tree.hip.h — the stack kernel's view of a chunk's octree: two levels per node.
//
// The reference's tree (src/Octree.h:16-26: one 32-bit word per node, a BRANCH points at its 8 children) costs one
// dependent load per level.  On upload every chunk's tree[] is also expanded into WIDE nodes of 4x4x4 = 64 entries: the
// wide node of a BRANCH at level L holds, for each of its 64 grandchild positions, what a descent from that BRANCH ends
// in after at most two levels -
//     terminal  [type:2 | level:5 | payload:25]   EMPTY / LEAF (payload = material) / TWIG (payload = brick index) and the
//                                                 level of the reference node (a child of the BRANCH, level L+1, fills
//                                                 the 8 positions it covers; a grandchild, level L+2, fills one), or
//     branch    [2 | 0 | wide node index]         the grandchild is a BRANCH itself: descend into its wide node.
// For the hit record (svo_hit.node) every wide node keeps WIDE_BASE_WORDS = 9 reference indices (wbase): the block of the 8
// children it expands and the blocks of their children - entry (child ci, grandchild gi) stands for node wbase[0] + ci if it is a
// child-level terminal, wbase[1 + ci] + gi otherwise; nothing else of tree[] is needed by the stack kernel.  (The per-entry
// reference indices the builder links the levels with are scratch.)  The march is unchanged - the same leaf node, hence the same box and the same
// floats, is found for every position - in half the dependent loads and half the descent-loop rounds.
//
// Wide levels are counted from the top: wide level k covers the cell-coordinate bits [2(nw-1-k)+1 : 2(nw-1-k)],
// nw = max(1, ceil(levels / 2)) with levels = depth - 2.  When `levels` is odd (or 0) the chunk root sits pad = 2 nw -
// levels virtual levels below the top wide node, whose unreachable entries are EMPTY.
//
// Built on the device, level by level (one thread per entry, an exclusive scan numbers the new wide nodes in entry
// order), from the chunk's node words as they lie in the tree pool.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "svo_format.h"

namespace svo {

constexpr uint32_t WIDE_LEVEL_SHIFT = 25, WIDE_LEVEL_BITS = 5, WIDE_PAYLOAD_MASK = (1u << 25) - 1u;
constexpr uint32_t WIDE_MAX_LEVELS = 22;        // branch levels the stack kernel marches (chunk depth <= 24: cell coordinates stay exact floats)

__device__ __forceinline__ uint32_t wide_terminal(uint32_t word, uint32_t level)
{   // word: a reference node word that is not a BRANCH
    const uint32_t type = node_type(word);
    const uint32_t payload = type == LEAF ? (node_offset(word) & 0xFFFFu) : type == TWIG ? (node_offset(word) & WIDE_PAYLOAD_MASK) : 0u;
    return (type << 30) | (level << WIDE_LEVEL_SHIFT) | payload;
}

// One wide level of one chunk.  `front[i]` = reference node index of the BRANCH the i-th wide node of this level expands
// (ignored for the top level when pad > 0).  Writes the 64 entries and reference indices of every wide node; an entry
// whose grandchild is a BRANCH is left as (BRANCH << 30) with flag = 1 and ref = that grandchild: k_wide_link numbers it.
__global__ __launch_bounds__(256) void k_wide_expand(const uint32_t *tree, const uint32_t *front, uint32_t count, uint32_t first_wide,
                                                     int top_pad, uint32_t level_child, uint32_t *wide, uint32_t *wref, uint32_t *wbase, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count * 64u) return;
    const uint32_t wn = i >> 6, slot = i & 63u;
    const uint32_t sx = slot & 3u, sy = (slot >> 2) & 3u, sz = slot >> 4;
    const uint32_t ci = (sx >> 1) | ((sy >> 1) << 1) | ((sz >> 1) << 2);        // Octree::branch of the child, src/Octree.cpp:55-58
    const uint32_t gi = (sx & 1u) | ((sy & 1u) << 1) | ((sz & 1u) << 2);        // ... and of the grandchild
    uint32_t entry = 0u, ref = 0u, fl = 0u;                                     // default: EMPTY at level 0 (unreachable positions)
    uint32_t base0 = 0u, base1 = 0u;                                            // block of the children / of child ci's children
    if (top_pad == 2) {                         // levels == 0: the chunk root is a terminal at the grandchild position 0
        if (slot == 0u) { entry = wide_terminal(tree[0], 0u); ref = 0u; }
    } else {
        uint32_t child = 0u, wc = 0u;
        bool reachable = true;
        if (top_pad == 1) {                     // the chunk root is the child at position 0 of a virtual node
            reachable = ci == 0u;
            child = 0u;
        } else {
            const uint32_t r = front[wn];
            const uint32_t wr = tree[r];
            if (node_type(wr) != BRANCH) { entry = wide_terminal(wr, 0u); ref = r; reachable = false; }    // a chunk that is one terminal node
            else { base0 = node_offset(wr); child = base0 + ci; }
        }
        if (reachable) {
            wc = tree[child];
            if (node_type(wc) != BRANCH) { entry = wide_terminal(wc, level_child); ref = child; }
            else {
                base1 = node_offset(wc);
                const uint32_t g = base1 + gi;
                const uint32_t wg = tree[g];
                ref = g;
                if (node_type(wg) != BRANCH) entry = wide_terminal(wg, level_child + 1u);
                else { entry = BRANCH << 30; fl = 1u; }
            }
        }
    }
    const uint64_t o = (uint64_t)(first_wide + wn) * 64u + slot;
    wide[o] = entry; wref[o] = ref; flag[i] = fl;
    uint32_t *wb = wbase + (uint64_t)(first_wide + wn) * WIDE_BASE_WORDS;
    if (slot == 0u) wb[0] = base0;
    if (gi == 0u) wb[1 + ci] = base1;
}

// rank = exclusive scan of flag: the flagged entries become BRANCH -> wide node (next_first + rank), and their reference
// nodes the next level's front, in entry order
__global__ __launch_bounds__(256) void k_wide_link(uint32_t count, uint32_t first_wide, uint32_t next_first, const uint32_t *flag,
                                                   const uint32_t *rank, uint32_t *wide, const uint32_t *wref, uint32_t *next_front)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= count * 64u || !flag[i]) return;
    const uint64_t o = (uint64_t)first_wide * 64u + i;
    const uint32_t r = rank[i];
    wide[o] = (BRANCH << 30) | (next_first + r);
    next_front[r] = wref[o];
}

} // namespace svo
