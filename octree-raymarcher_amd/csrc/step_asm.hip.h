// step_asm.hip.h — the march step of k_trace_stack, hand-scheduled for gfx950 (one asm statement).
//
// Same arithmetic, operand for operand, as the C++ step in kernel_stack.hip.h (which stays the readable statement of the
// algorithm and the A/B partner: -DSVO_CXX_STEP); what differs is everything the compiler adds around it.  The step is
// bound by instruction issue, and of the ~285 instructions hipcc emits for it about 70 are scalar EXEC
// bookkeeping (one s_and_saveexec / s_or pair per source-level `if`, s_or chains that AND six compares together), 12 are
// s_nop pads behind v_cmp -> v_cndmask pairs it did not interleave, and up to 47 are register copies at control-flow joins.
// Here:
//   * the six box compares and the step cap narrow EXEC directly (v_cmpx chain: no scalar combine);
//   * every outcome class (leave / tree / brick / advance+enter / enter) is entered by ONE s_mov / s_and of a lane mask kept
//     in an SGPR pair, and nothing is restored in between - EXEC is put back once, at the end;
//   * the three axes of the escape evaluation are interleaved so that each v_cndmask finds its compare two issue slots
//     back (the VALU-writes-SGPR -> VALU-reads-it hazard of gfx940+ needs two wait states), two s_nop remain;
//   * lanes that enter a brick ride along with the advancing lanes through the one escape evaluation (their t_miss),
//     lanes that leave evaluate nothing (kernel_stack.hip.h, "the one escape evaluation of the step");
//   * no value is copied at a join: each outcome writes the lane state in place under its own mask.
//
//   * one load per lane and step: a BRANCH entry is not chased inside the step (see "a BRANCH entry" below);
//   * latency: the PMC counters of the first hand-written version (22 % fewer instructions, the same wave cycles, waves 67 %
//     of their time in s_waitcnt) showed what a wave-step really costs - its chain of dependent loads, the slowest lane's
//     at that.  So the statement runs SEVERAL steps (the steps of one pass of the outer loop), the descent's first load is
//     issued before the box test and everything else that does not need it (speculatively also for lanes about to leave:
//     the address stays inside the cached wide node), and the occupancy mask of a brick just entered is not waited for
//     where it is loaded but where the NEXT step tests it: behind that step's own descent load (loads return in order),
//     i.e. for free.
//
// Hazards honoured by hand (nothing inside an asm statement is padded by the compiler): v_cmp writing VCC / an SGPR pair
// -> v_cndmask / v_subbrev reading it: >= 2 instructions in between; every load is waited for inside the statement (the last
// brick mask at its very end), so that every output is valid when the statement ends.
//
// Lane modes are numbered so that "marching" is one unsigned compare: DONE 0, WORLD 1, HIT 2, TREE 3, TWIG 4.
#pragma once
#include "march.hip.h"

#ifndef SVO_DESCEND_SHIFT
#define SVO_DESCEND_SHIFT 2      // a wide level is taken inside the step when more than 1 / 2^this of the wave's tree lanes stand at a BRANCH entry
#endif
#ifdef SVO_STACK_TIMING
#define SVO_STAT(text) text
#else
#define SVO_STAT(text) ""
#endif
#define SVO_STR_(x) #x
#define SVO_STR(x) SVO_STR_(x)

namespace svo {

struct StepStats { unsigned steps = 0, lanes = 0, stalls = 0, chased = 0; };    // (-DSVO_STACK_TIMING) wave-steps, marching lanes summed over them, lanes that sat a BRANCH out, lanes that took a level inside the step

struct StepUniform {            // wave-uniform inputs (SGPRs)
    float csize, eps, eps2;
    int cap_twig;
    const uint32_t *wide;
    const uint64_t *mask;
};

// `nsteps` (>= 1, wave-uniform) steps of every marching lane.  All lanes of the wave must call this together (EXEC is
// saved and restored here).
__device__ __forceinline__ void march_steps_asm(
    int &mode, V3 &O, V3 &Blo, float &bsize, float &res, float &t, int &cnt, float &tt_saved, float &t_miss, int &it_saved,
    float &tw, int &cw, int &pux, int &puy, int &puz, int &valid, int &plev, unsigned long long &bmask, int &creepn,
    const V3 beta, const V3 g, const V3 clo, const V3 alpha, const int levels, const int nw, const float res_tree,
    const uint32_t wide_b, const uint32_t twig_off, const uint32_t lds_lane, const StepUniform U, const int nsteps
#ifdef SVO_STACK_TIMING
    , StepStats &stats
#endif
    )
{
    float px, py, pz, q1, q2, q3, q4, q5, q6, q7, r1, r2, r3;     // (r1..r3 double as the lattice quotients, low as 1/res, q7 as the brick cell index)
    int ux, uy, uz, low;
    uint32_t w;
    unsigned long long sall, smar, stw, sstay, sadv, sent, q64;
    int sctr, na, nb;
#ifdef SVO_STACK_TIMING
    unsigned st_steps = (unsigned)__builtin_amdgcn_readfirstlane((int)stats.steps), st_lanes = (unsigned)__builtin_amdgcn_readfirstlane((int)stats.lanes);
    unsigned st_stalls = (unsigned)__builtin_amdgcn_readfirstlane((int)stats.stalls), st_chased = (unsigned)__builtin_amdgcn_readfirstlane((int)stats.chased);
#endif
    asm volatile(
        "s_mov_b64 %[sall], exec\n\t"
        "s_mov_b32 %[sctr], %[nst]\n\t"
        "0:\n\t"
        "v_cmpx_lt_u32 vcc, 2, %[md]\n\t"                      // marching lanes: TREE (3) or TWIG (4)
        "s_mov_b64 %[smar], exec\n\t"
        "s_cbranch_execz 91f\n\t"                              // nobody: the remaining steps would do nothing either
        SVO_STAT("s_bcnt1_i32_b64 %[na], exec\n\t" "s_add_u32 %[st_steps], %[st_steps], 1\n\t" "s_add_u32 %[st_lanes], %[st_lanes], %[na]\n\t")
        // ---- p = O + beta*t and its lattice coordinates in the level's box (src/Traverse.cpp:80,55-58)
        "v_cmp_eq_u32_e64 %[stw], 4, %[md]\n\t"
        "v_mul_f32 %[px], %[bx], %[t]\n\t"
        "v_mul_f32 %[py], %[by], %[t]\n\t"
        "v_mul_f32 %[pz], %[bz], %[t]\n\t"
        "v_sub_u32 %[low], 0x7f000000, %[rs]\n\t"               // 1/res, res a power of two
        "v_add_f32 %[px], %[ox], %[px]\n\t"
        "v_add_f32 %[py], %[oy], %[py]\n\t"
        "v_add_f32 %[pz], %[oz], %[pz]\n\t"
        "v_add_u32 %[cnt], -1, %[cnt]\n\t"                     // steps left of the level's cap, this one taken
        "v_sub_f32 %[r1], %[px], %[lx]\n\t"
        "v_sub_f32 %[r2], %[py], %[ly]\n\t"
        "v_sub_f32 %[r3], %[pz], %[lz]\n\t"
        "v_sub_u32 %[q7], 0, %[crp]\n\t"
        "v_mul_f32 %[r1], %[r1], %[low]\n\t"
        "v_mul_f32 %[r2], %[r2], %[low]\n\t"
        "v_mul_f32 %[r3], %[r3], %[low]\n\t"
        "v_min_i32 %[crp], %[crp], %[q7]\n\t"                  // creepn = -|creepn|: disarmed unless this step advances
        "v_cvt_i32_f32 %[ux], %[r1]\n\t"
        "v_cvt_i32_f32 %[uy], %[r2]\n\t"
        "v_cvt_i32_f32 %[uz], %[r3]\n\t"
        // ---- tree level, first half: start the descent (src/Traverse.cpp:34-48 through the wide tree and the descent
        //      cache) for every marching tree lane, before it is known whether the lane stays in its box
        "s_andn2_b64 exec, %[smar], %[stw]\n\t"
        "s_cbranch_execz 20f\n\t"
        "v_fract_f32 %[q1], %[r1]\n\t"                         // integral quotient: p on (or rounded onto) a lattice plane
        "v_fract_f32 %[q2], %[r2]\n\t"
        "v_fract_f32 %[q3], %[r3]\n\t"
        "v_xor_b32 %[q4], %[ux], %[pux]\n\t"
        "v_min3_f32 %[q1], %[q1], %[q2], %[q3]\n\t"
        "v_xor_b32 %[q5], %[uy], %[puy]\n\t"
        "v_cmp_eq_f32 vcc, 0, %[q1]\n\t"
        "s_cbranch_vccnz 7f\n\t"
        "1:\n\t"
        "v_xor_b32 %[q6], %[uz], %[puz]\n\t"
        "v_or3_b32 %[q4], %[q4], %[q5], %[q6]\n\t"
        "v_or_b32 %[q4], 1, %[q4]\n\t"
        "v_ffbh_u32 %[q4], %[q4]\n\t"
        "v_sub_u32 %[q4], 33, %[q4]\n\t"
        "v_lshrrev_b32 %[q4], 1, %[q4]\n\t"
        "v_sub_u32 %[q4], %[nw], %[q4]\n\t"
        "v_med3_i32 %[val], %[q4], 0, %[val]\n\t"              // deepest cached wide level whose node is unchanged
        "v_lshl_add_u32 %[q2], %[val], 8, %[lds]\n\t"
        "ds_read_b32 %[q3], %[q2]\n\t"
        "v_sub_u32 %[q4], %[nw], %[val]\n\t"
        "v_lshl_add_u32 %[q4], %[q4], 1, -2\n\t"               // the two coordinate bits that select the entry
        "v_bfe_u32 %[q1], %[ux], %[q4], 2\n\t"
        "v_bfe_u32 %[q5], %[uy], %[q4], 2\n\t"
        "v_bfe_u32 %[q6], %[uz], %[q4], 2\n\t"
        "v_lshl_or_b32 %[q1], %[q5], 2, %[q1]\n\t"
        "v_lshl_or_b32 %[q1], %[q6], 4, %[q1]\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_lshl_or_b32 %[q1], %[q3], 6, %[q1]\n\t"
        "v_lshl_add_u32 %[q1], %[q1], 2, %[wb]\n\t"
        "global_load_dword %[w], %[q1], %[wide]\n\t"           // (waited for at 5:)
        "2:\n\t"
        // ---- isInsideCube(p, Blo, Blo + Bsize) and the cap: EXEC narrows to the lanes that stay (:81,56; cap :54,79)
        "s_mov_b64 exec, %[smar]\n\t"
        "v_add_f32 %[q5], %[lx], %[bs]\n\t"
        "v_add_f32 %[q6], %[ly], %[bs]\n\t"
        "v_add_f32 %[q7], %[lz], %[bs]\n\t"
        "v_cmpx_ge_f32 vcc, %[px], %[lx]\n\t"
        "v_cmpx_ge_f32 vcc, %[py], %[ly]\n\t"
        "v_cmpx_ge_f32 vcc, %[pz], %[lz]\n\t"
        "v_cmpx_ge_f32 vcc, %[q5], %[px]\n\t"
        "v_cmpx_ge_f32 vcc, %[q6], %[py]\n\t"
        "v_cmpx_ge_f32 vcc, %[q7], %[pz]\n\t"
        "v_cmpx_le_i32 vcc, 0, %[cnt]\n\t"
        "v_max3_i32 %[low], %[ux], %[uy], %[uz]\n\t"
        "v_cmp_lt_i32 vcc, 3, %[low]\n\t"                      // brick: isInsideCube(off, 0, 3), :59
        "s_and_b64 vcc, vcc, %[stw]\n\t"
        "s_andn2_b64 %[sstay], exec, vcc\n\t"
        // ---- lanes that leave their level: no escape evaluation here (see kernel_stack.hip.h)
        "s_andn2_b64 exec, %[smar], %[sstay]\n\t"
        "s_cbranch_execz 8f\n\t"
        "s_andn2_b64 exec, exec, %[stw]\n\t"                   // out of the chunk (:164-168): the chunk step advances tw
        "v_or_b32 %[cw], 0x80000000, %[cw]\n\t"
        "v_mov_b32 %[md], 1\n\t"
        "s_andn2_b64 exec, %[smar], %[sstay]\n\t"
        "s_and_b64 exec, exec, %[stw]\n\t"                     // out of a brick (:104-105): resume the tree level at t_miss
        "s_cbranch_execz 8f\n\t"
        "v_mul_f32 %[q1], %[bx], %[tw]\n\t"
        "v_mul_f32 %[q2], %[by], %[tw]\n\t"
        "v_mul_f32 %[q3], %[bz], %[tw]\n\t"
        "v_mov_b32 %[t], %[tms]\n\t"
        "v_mov_b32 %[cnt], %[its]\n\t"
        "v_add_f32 %[ox], %[ax], %[q1]\n\t"                    // the chunk march's p (:144,158)
        "v_add_f32 %[oy], %[ay], %[q2]\n\t"
        "v_add_f32 %[oz], %[az], %[q3]\n\t"
        "v_mov_b32 %[lx], %[clx]\n\t"
        "v_mov_b32 %[ly], %[cly]\n\t"
        "v_mov_b32 %[lz], %[clz]\n\t"
        "v_mov_b32 %[rs], %[rtr]\n\t"
        "v_mov_b32 %[bs], %[csz]\n\t"
        "v_mov_b32 %[md], 3\n\t"
        "8:\n\t"
        "s_mov_b64 %[sadv], 0\n\t"
        "s_mov_b64 %[sent], 0\n\t"
        // ---- brick level: the cell's occupancy bit (src/Traverse.cpp:58-66).  The mask of a brick entered in the previous
        //      step may still be on its way: it was issued before this step's descent load, so at most one load outstanding
        //      means it has arrived (20: below waits for it where no descent load was issued)
        "s_and_b64 exec, %[sstay], %[stw]\n\t"
        "s_cbranch_execz 5f\n\t"
        "v_lshl_add_u32 %[q7], %[uz], 2, %[uy]\n\t"
        "v_mov_b32 %[low], 0\n\t"
        "v_lshl_add_u32 %[q7], %[q7], 2, %[ux]\n\t"            // cell index z*16 + y*4 + x
        "v_sub_u32 %[q1], 63, %[q7]\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_lshlrev_b64 %[q64], %[q1], %[bm]\n\t"                // the cell's bit of the occupancy mask -> sign bit
        "v_cmp_gt_i64 vcc, 0, %[q64]\n\t"                       // occupied
        "s_andn2_b64 %[sadv], exec, vcc\n\t"
        "s_and_b64 exec, exec, vcc\n\t"                         // occupied: hit, :63,101,160
        "v_add_f32 %[q1], %[t], %[tts]\n\t"
        "v_mov_b32 %[cnt], %[q7]\n\t"
        "v_mov_b32 %[md], 2\n\t"
        "v_add_f32 %[tw], %[tw], %[q1]\n\t"
        "5:\n\t"
        // ---- tree level, second half: the entry has arrived
        "s_andn2_b64 exec, %[sstay], %[stw]\n\t"
        "s_cbranch_execz 6f\n\t"
        "s_mov_b64 %[smar], exec\n\t"
        "v_mov_b32 %[pux], %[ux]\n\t"                          // the descent cache is keyed to this cell either way
        "v_mov_b32 %[puy], %[uy]\n\t"
        "v_mov_b32 %[puz], %[uz]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        // a BRANCH entry (top bits 10) means one more wide level.  A lane does not chase it inside this step when few lanes are in that
        // position - the whole wave would wait out a second dependent load for the 1.4 % of lane-steps that need one, in 40 % of its
        // steps: it pushes the child node onto its descent cache (already keyed to this cell), takes its cap decrement back and sits the
        // step out; the next step recomputes the same p (t is unchanged), finds the cache valid down to the new level and loads the
        // deeper entry as its FIRST load.  When more than a quarter of the wave's tree lanes stand at a BRANCH - fresh rays of one tile
        // descending from the root together, deep trees - the level is taken inside the step, for all of them at once.
        "3:\n\t"
        "v_cmpx_gt_i32 vcc, -2.0, %[w]\n\t"
        "s_cbranch_execz 4f\n\t"
        "s_bcnt1_i32_b64 %[na], exec\n\t"
        "s_bcnt1_i32_b64 %[nb], %[smar]\n\t"
        "v_and_b32 %[q3], 0x1ffffff, %[w]\n\t"
        "v_add_u32 %[val], 1, %[val]\n\t"
        "s_lshr_b32 %[nb], %[nb], " SVO_STR(SVO_DESCEND_SHIFT) "\n\t"
        "v_lshl_add_u32 %[q2], %[val], 8, %[lds]\n\t"
        "ds_write_b32 %[q2], %[q3]\n\t"
        "s_cmp_gt_u32 %[na], %[nb]\n\t"
        "s_cbranch_scc0 35f\n\t"
        SVO_STAT("s_add_u32 %[st_chased], %[st_chased], %[na]\n\t")
        "v_sub_u32 %[q4], %[nw], %[val]\n\t"                   // many: the next wide level now
        "v_lshl_add_u32 %[q4], %[q4], 1, -2\n\t"
        "v_bfe_u32 %[q1], %[ux], %[q4], 2\n\t"
        "v_bfe_u32 %[q5], %[uy], %[q4], 2\n\t"
        "v_bfe_u32 %[q6], %[uz], %[q4], 2\n\t"
        "v_lshl_or_b32 %[q1], %[q5], 2, %[q1]\n\t"
        "v_lshl_or_b32 %[q1], %[q6], 4, %[q1]\n\t"
        "v_lshl_or_b32 %[q1], %[q3], 6, %[q1]\n\t"
        "v_lshl_add_u32 %[q1], %[q1], 2, %[wb]\n\t"
        "global_load_dword %[w], %[q1], %[wide]\n\t"
        "s_mov_b64 exec, %[smar]\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch 3b\n\t"
        "35:\n\t"
        SVO_STAT("s_add_u32 %[st_stalls], %[st_stalls], %[na]\n\t")
        "v_add_u32 %[cnt], 1, %[cnt]\n\t"                      // few: these lanes sit the step out
        "s_andn2_b64 %[smar], %[smar], exec\n\t"
        "4:\n\t"
        "s_mov_b64 exec, %[smar]\n\t"                          // the tree lanes whose entry is terminal
        "v_bfe_u32 %[plv], %[w], 25, 5\n\t"                    // the reference node's level
        "v_cmp_gt_u32_e64 %[sstay], 2.0, %[w]\n\t"             // EMPTY (type bits 00)
        "v_cmp_le_u32_e64 %[sent], -2.0, %[w]\n\t"             // TWIG  (type bits 11)
        "v_sub_u32 %[q1], %[lev], %[plv]\n\t"
        "v_bfm_b32 %[low], %[q1], 0\n\t"                       // the node spans low + 1 cells
        "s_or_b64 %[sadv], %[sadv], %[sstay]\n\t"
        "v_cmpx_le_i32 vcc, 2.0, %[w]\n\t"                     // LEAF (type bits 01): hit, src/Traverse.cpp:93,160
        "v_subrev_f32 %[q1], %[eps], %[t]\n\t"
        "v_mov_b32 %[cnt], 0xff\n\t"
        "v_mov_b32 %[md], 2\n\t"
        "v_add_f32 %[tw], %[tw], %[q1]\n\t"
        "6:\n\t"
        // ---- the one escape evaluation: out of the located cell from p (src/Traverse.cpp:25-32,89,104-105,67)
        "s_or_b64 exec, %[sadv], %[sent]\n\t"
        "s_cbranch_execz 90f\n\t"
        "v_not_b32 %[q7], %[low]\n\t"
        "v_add_u32 %[q4], 1, %[low]\n\t"
        "v_and_b32 %[q1], %[ux], %[q7]\n\t"
        "v_and_b32 %[q2], %[uy], %[q7]\n\t"
        "v_and_b32 %[q3], %[uz], %[q7]\n\t"
        "v_cvt_f32_u32 %[q4], %[q4]\n\t"
        "v_cvt_f32_i32 %[q1], %[q1]\n\t"
        "v_cvt_f32_i32 %[q2], %[q2]\n\t"
        "v_cvt_f32_i32 %[q3], %[q3]\n\t"
        "v_mul_f32 %[q4], %[rs], %[q4]\n\t"                    // cell edge
        "v_fma_f32 %[q1], %[q1], %[rs], %[lx]\n\t"             // cell lo = l + k*res: product and sum are exact on exact geometry
        "v_fma_f32 %[q2], %[q2], %[rs], %[ly]\n\t"             // (lattice values below 2^24 steps), so the fused form rounds
        "v_fma_f32 %[q3], %[q3], %[rs], %[lz]\n\t"             // nowhere the reference's mul + add would
        "v_add_f32 %[q5], %[q1], %[q4]\n\t"                    // cell hi
        "v_add_f32 %[q6], %[q2], %[q4]\n\t"
        "v_add_f32 %[q7], %[q3], %[q4]\n\t"
        "v_sub_f32 %[r1], %[q1], %[px]\n\t"
        "v_sub_f32 %[r2], %[q2], %[py]\n\t"
        "v_sub_f32 %[r3], %[q3], %[pz]\n\t"
        "v_sub_f32 %[q5], %[q5], %[px]\n\t"
        "v_sub_f32 %[q6], %[q6], %[py]\n\t"
        "v_sub_f32 %[q7], %[q7], %[pz]\n\t"
        "v_mul_f32 %[r1], %[r1], %[gx]\n\t"                    // (lo - p) * gamma
        "v_mul_f32 %[r2], %[r2], %[gy]\n\t"
        "v_mul_f32 %[r3], %[r3], %[gz]\n\t"
        "v_mul_f32 %[q5], %[q5], %[gx]\n\t"                    // (hi - p) * gamma
        "v_mul_f32 %[q6], %[q6], %[gy]\n\t"
        "v_mul_f32 %[q7], %[q7], %[gz]\n\t"
        "v_cmp_lt_f32 vcc, %[r1], %[q5]\n\t"                   // glm::max(tmin, tmax) = (tmin < tmax) ? tmax : tmin per axis
        "v_cmp_lt_f32_e64 %[smar], %[r2], %[q6]\n\t"
        "v_cmp_lt_f32_e64 %[sstay], %[r3], %[q7]\n\t"
        "v_cndmask_b32 %[r1], %[r1], %[q5], vcc\n\t"
        "v_cndmask_b32_e64 %[r2], %[r2], %[q6], %[smar]\n\t"
        "v_cndmask_b32_e64 %[r3], %[r3], %[q7], %[sstay]\n\t"
        "v_sub_u32 %[q5], 1, %[crp]\n\t"                       // |creepn| + 1
        "v_cmp_lt_f32 vcc, %[r3], %[r2]\n\t"                   // glm::min(t.y, t.z) = (t.z < t.y) ? t.z : t.y
        "v_and_b32 %[q6], 0x1ffffff, %[w]\n\t"                 // (entering lanes) brick index ...
        "v_add_u32 %[q6], %[tof], %[q6]\n\t"
        "v_cndmask_b32 %[r2], %[r2], %[r3], vcc\n\t"
        "v_cmp_lt_f32 vcc, %[r2], %[r1]\n\t"                   // glm::min(t.x, .)
        "v_lshlrev_b32 %[q6], 3, %[q6]\n\t"                    // ... as a byte offset into the mask pool
        "s_nop 0\n\t"
        "v_cndmask_b32 %[r1], %[r1], %[r2], vcc\n\t"
        "v_add_f32 %[r1], %[eps], %[r1]\n\t"                   // escape + EPS
        // advance: t += e; creepn = e < 2 EPS ? |creepn| + 1 : 0
        "s_mov_b64 exec, %[sadv]\n\t"
        "v_cmp_gt_f32 vcc, %[eps2], %[r1]\n\t"
        "v_add_f32 %[t], %[t], %[r1]\n\t"
        "s_nop 0\n\t"
        "v_cndmask_b32 %[crp], 0, %[q5], vcc\n\t"
        // enter the brick: twigmarch(p, b, node box, ...), a = p, t = 0 (src/Traverse.cpp:99,53)
        "s_mov_b64 exec, %[sent]\n\t"
        "s_cbranch_execz 90f\n\t"
        "global_load_dwordx2 %[bm], %[q6], %[maskp]\n\t"       // (waited for by the next step's brick test, or at 91:)
        "v_mov_b32 %[tts], %[t]\n\t"
        "v_add_f32 %[tms], %[t], %[r1]\n\t"
        "v_mov_b32 %[its], %[cnt]\n\t"
        "v_mov_b32 %[ox], %[px]\n\t"
        "v_mov_b32 %[oy], %[py]\n\t"
        "v_mov_b32 %[oz], %[pz]\n\t"
        "v_mov_b32 %[t], 0\n\t"
        "v_mov_b32 %[cnt], %[captw]\n\t"
        "v_mov_b32 %[lx], %[q1]\n\t"
        "v_mov_b32 %[ly], %[q2]\n\t"
        "v_mov_b32 %[lz], %[q3]\n\t"
        "v_mov_b32 %[bs], %[q4]\n\t"
        "v_mul_f32 %[rs], 0.25, %[q4]\n\t"                     // leafsize = node size / 4
        "v_mov_b32 %[md], 4\n\t"
        "90:\n\t"
        "s_mov_b64 exec, %[sall]\n\t"
        "s_add_i32 %[sctr], %[sctr], -1\n\t"
        "s_cmp_lg_u32 %[sctr], 0\n\t"
        "s_cbranch_scc1 0b\n\t"
        "s_branch 91f\n\t"
        // ---- out of line: no tree lane marching - nothing of this step's is in flight, a brick mask may be
        "20:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_branch 2b\n\t"
        // ---- out of line: the integral-quotient fix-up (rare per wave): the reference's own comparison settles the cell
        "7:\n\t"
        "s_and_saveexec_b64 %[sadv], vcc\n\t"
        "v_bfm_b32 %[q1], %[lev], 0\n\t"                       // nmax = 2^levels - 1
        "v_min_i32 %[ux], %[ux], %[q1]\n\t"
        "v_min_i32 %[uy], %[uy], %[q1]\n\t"
        "v_min_i32 %[uz], %[uz], %[q1]\n\t"
        "v_cvt_f32_i32 %[q1], %[ux]\n\t"
        "v_cvt_f32_i32 %[q2], %[uy]\n\t"
        "v_cvt_f32_i32 %[q3], %[uz]\n\t"
        "v_mul_f32 %[q1], %[q1], %[rs]\n\t"
        "v_mul_f32 %[q2], %[q2], %[rs]\n\t"
        "v_mul_f32 %[q3], %[q3], %[rs]\n\t"
        "v_add_f32 %[q1], %[lx], %[q1]\n\t"
        "v_add_f32 %[q2], %[ly], %[q2]\n\t"
        "v_add_f32 %[q3], %[lz], %[q3]\n\t"
        "v_cmp_gt_f32 vcc, %[q1], %[px]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[ux], vcc, 0, %[ux], vcc\n\t"
        "v_cmp_gt_f32 vcc, %[q2], %[py]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[uy], vcc, 0, %[uy], vcc\n\t"
        "v_cmp_gt_f32 vcc, %[q3], %[pz]\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32 %[uz], vcc, 0, %[uz], vcc\n\t"
        "s_mov_b64 exec, %[sadv]\n\t"
        "v_xor_b32 %[q4], %[ux], %[pux]\n\t"
        "v_xor_b32 %[q5], %[uy], %[puy]\n\t"
        "s_branch 1b\n\t"
        "91:\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[sall]\n\t"
        : [md] "+v"(mode), [ox] "+v"(O.x), [oy] "+v"(O.y), [oz] "+v"(O.z), [lx] "+v"(Blo.x), [ly] "+v"(Blo.y), [lz] "+v"(Blo.z),
          [bs] "+v"(bsize), [rs] "+v"(res), [t] "+v"(t), [cnt] "+v"(cnt), [tts] "+v"(tt_saved), [tms] "+v"(t_miss), [its] "+v"(it_saved),
          [tw] "+v"(tw), [cw] "+v"(cw), [pux] "+v"(pux), [puy] "+v"(puy), [puz] "+v"(puz), [val] "+v"(valid), [plv] "+v"(plev),
          [bm] "+v"(bmask), [crp] "+v"(creepn),
          [px] "=&v"(px), [py] "=&v"(py), [pz] "=&v"(pz), [ux] "=&v"(ux), [uy] "=&v"(uy), [uz] "=&v"(uz),
          [q1] "=&v"(q1), [q2] "=&v"(q2), [q3] "=&v"(q3), [q4] "=&v"(q4), [q5] "=&v"(q5), [q6] "=&v"(q6), [q7] "=&v"(q7),
          [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [low] "=&v"(low), [w] "=&v"(w), [q64] "=&v"(q64),
          [sall] "=&s"(sall), [smar] "=&s"(smar), [stw] "=&s"(stw), [sstay] "=&s"(sstay), [sadv] "=&s"(sadv), [sent] "=&s"(sent), [sctr] "=&s"(sctr), [na] "=&s"(na), [nb] "=&s"(nb)
#ifdef SVO_STACK_TIMING
          , [st_steps] "+s"(st_steps), [st_lanes] "+s"(st_lanes), [st_stalls] "+s"(st_stalls), [st_chased] "+s"(st_chased)
#endif
        : [bx] "v"(beta.x), [by] "v"(beta.y), [bz] "v"(beta.z), [gx] "v"(g.x), [gy] "v"(g.y), [gz] "v"(g.z),
          [clx] "v"(clo.x), [cly] "v"(clo.y), [clz] "v"(clo.z), [ax] "v"(alpha.x), [ay] "v"(alpha.y), [az] "v"(alpha.z),
          [lev] "v"(levels), [nw] "v"(nw), [rtr] "v"(res_tree), [wb] "v"(wide_b), [tof] "v"(twig_off), [lds] "v"(lds_lane),
          [csz] "s"(U.csize), [eps] "s"(U.eps), [eps2] "s"(U.eps2), [captw] "s"(U.cap_twig), [wide] "s"(U.wide), [maskp] "s"(U.mask), [nst] "s"(nsteps)
        : "vcc", "scc", "memory");
#ifdef SVO_STACK_TIMING
    stats.steps = st_steps; stats.lanes = st_lanes; stats.stalls = st_stalls; stats.chased = st_chased;
#endif
}

} // namespace svo
