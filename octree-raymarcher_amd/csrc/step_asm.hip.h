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
//   * (round 4) a draining wave - bit 16 of the statement's step count - does not enter a brick whose march is bound to miss: the reference resumes
//     the tree level from the brick's entry point alone (src/Traverse.cpp:99-105), so a miss proven from the brick's mask needs none of the
//     brick's steps (step_asm_body.inc, "sure").  Variants suremiss / suremiss64 run the test in every wave-step for the parity tests.
//
// Hazards honoured by hand (nothing inside an asm statement is padded by the compiler): v_cmp writing VCC / an SGPR pair
// -> v_cndmask / v_subbrev reading it: >= 2 instructions in between; every load is waited for inside the statement (the last
// brick mask at its very end), so that every output is valid when the statement ends.
//
// Lane modes are numbered so that "marching" is one unsigned compare: DONE 0, WORLD 1, HIT 2, TREE 3, TWIG 4.
//
// Two addressing variants of the same statement (step_asm_body.inc is included twice):
//   march_steps_asm      the chunk's wide tree as a 32-bit byte offset into the wide pool, the brick mask as a 32-bit byte offset
//                        into the mask pool, both against a scalar base (one VGPR of address each): wide pools below 4 GiB, fewer
//                        than 2^29 bricks - every world up to a few tens of GB;
//   march_steps_asm_big  the chunk's wide tree as a 64-bit address per lane, entry and mask addresses formed by one v_mad_u64_u32
//                        each (32-bit index * size + 64-bit base): any world that fits the device.  One VGPR more of lane state,
//                        the same instruction count.
#pragma once
#include "march.hip.h"

// A wide level is taken inside the step when more than 1 / 2^shift of the wave's tree lanes stand at a BRANCH entry (StepUniform::descend_shift).
// Round 4 re-swept it: a HALF (shift 1) instead of a quarter is +1.2 % on C3, +3 % on C2, nothing on C4 - and -2.6 % on C5, whose fresh rays
// descend seven wide levels from the root: shallow trees (<= 10 branch levels) use 1, deeper ones 2 (shift 3: -3 % on C3; 0 = never: C3 +0.2 %, C5 -7 %).
#ifndef SVO_DESCEND_SHIFT_SHALLOW
#define SVO_DESCEND_SHIFT_SHALLOW 1
#endif
#ifndef SVO_DESCEND_SHIFT_DEEP
#define SVO_DESCEND_SHIFT_DEEP 2
#endif
#ifdef SVO_STACK_TIMING
#define SVO_STAT(text) text
#else
#define SVO_STAT(text) ""
#endif
// (-DSVO_STACK_TIMING -DSVO_SURE_STAT=k: StepStats::sure counts the lanes that reach stage k of the sure-miss test - 1 entering a brick while
// the wave drains, 2 p(t_miss) provably outside, 3 guard passed, 4 skipped; of a draining wave: 5 lanes that leave a brick whose march missed, 6 lanes that hit a cell; default 4)
#ifndef SVO_SURE_STAT
#define SVO_SURE_STAT 4
#endif
#define SVO_SURE_COUNT(k) SVO_SURE_PICK(k, SVO_SURE_STAT)
#define SVO_SURE_PICK(k, want) SVO_SURE_PICK_(k, want)
#define SVO_SURE_PICK_(k, want) SVO_SURE_IS_##k##_##want
#define SVO_SURE_TEXT SVO_STAT("s_bcnt1_i32_b64 %[na], exec\n\t" "s_add_u32 %[st_sure], %[st_sure], %[na]\n\t")
#define SVO_SURE_DRAIN_TEXT SVO_STAT("s_bitcmp1_b32 %[nst], 16\n\t" "s_cbranch_scc0 69f\n\t" "s_bcnt1_i32_b64 %[na], exec\n\t" "s_add_u32 %[st_sure], %[st_sure], %[na]\n\t" "69:\n\t")
#define SVO_SURE_IS_5_5 SVO_SURE_DRAIN_TEXT
#define SVO_SURE_IS_6_6 SVO_SURE_DRAIN_TEXT
#define SVO_SURE_IS_1_1 SVO_SURE_TEXT
#define SVO_SURE_IS_2_2 SVO_SURE_TEXT
#define SVO_SURE_IS_3_3 SVO_SURE_TEXT
#define SVO_SURE_IS_4_4 SVO_SURE_TEXT
#define SVO_SURE_IS_1_2 ""
#define SVO_SURE_IS_1_3 ""
#define SVO_SURE_IS_1_4 ""
#define SVO_SURE_IS_2_1 ""
#define SVO_SURE_IS_2_3 ""
#define SVO_SURE_IS_2_4 ""
#define SVO_SURE_IS_3_1 ""
#define SVO_SURE_IS_3_2 ""
#define SVO_SURE_IS_3_4 ""
#define SVO_SURE_IS_4_1 ""
#define SVO_SURE_IS_4_2 ""
#define SVO_SURE_IS_4_3 ""
#define SVO_SURE_IS_1_5 ""
#define SVO_SURE_IS_1_6 ""
#define SVO_SURE_IS_2_5 ""
#define SVO_SURE_IS_2_6 ""
#define SVO_SURE_IS_3_5 ""
#define SVO_SURE_IS_3_6 ""
#define SVO_SURE_IS_4_5 ""
#define SVO_SURE_IS_4_6 ""
#define SVO_SURE_IS_5_1 ""
#define SVO_SURE_IS_5_2 ""
#define SVO_SURE_IS_5_3 ""
#define SVO_SURE_IS_5_4 ""
#define SVO_SURE_IS_5_6 ""
#define SVO_SURE_IS_6_1 ""
#define SVO_SURE_IS_6_2 ""
#define SVO_SURE_IS_6_3 ""
#define SVO_SURE_IS_6_4 ""
#define SVO_SURE_IS_6_5 ""
#define SVO_STR_(x) #x
#define SVO_STR(x) SVO_STR_(x)

namespace svo {

struct StepStats { unsigned steps = 0, lanes = 0, stalls = 0, chased = 0, sure = 0; };    // (-DSVO_STACK_TIMING) wave-steps, marching lanes summed over them, lanes that sat a BRANCH out, lanes that took a level inside the step

struct StepUniform {            // wave-uniform inputs (SGPRs)
    float csize, eps, eps2;
    int cap_twig;
    const uint32_t *wide;
    const uint64_t *mask;
    int descend_shift;
};

// The statement's variants (step_asm_body.inc is included once per variant):
//   addressing   32-bit offsets against scalar bases / 64-bit addresses (entry: index * 4 + the lane's wide-tree address; mask:
//                brick index * 8 + the pool's address; the carry-out lands in VCC, which is dead at all three places; q64 is free
//                there: the brick test is its only other user);
//   semantics    the CPU march (src/Traverse.cpp) / its GLSL twin (shaders/Chunkmarch.glsl): the escape distance's guard
//                `d < EPS ? BIGEPS : d` (:113; three instructions, q7 is free behind the per-axis maxima) and a LEAF hit at t
//                instead of t - EPS (:266).  The entry condition, the containment re-check and the constants live outside the step.
#define SVO_STEP_ADDR32_ENTRY "v_lshl_add_u32 %[q1], %[q1], 2, %[wb]\n\t" "global_load_dword %[w], %[q1], %[wide]\n\t"
#define SVO_STEP_ADDR32_MASKOFF "v_lshlrev_b32 %[q6], 3, %[q6]\n\t" "s_nop 0\n\t"
#define SVO_STEP_ADDR32_MASK "global_load_dwordx2 %[bm], %[q6], %[maskp]\n\t"
#define SVO_STEP_ADDR32_MASK_HALVES "global_load_dword %[r3], %[q6], %[maskp]\n\t" "global_load_dword %[w], %[q6], %[maskp] offset:4\n\t"
#define SVO_STEP_ADDR64_ENTRY "v_mad_u64_u32 %[q64], vcc, %[q1], 4, %[wb]\n\t" "global_load_dword %[w], %[q64], off\n\t"
#define SVO_STEP_ADDR64_MASKOFF "s_nop 1\n\t"
#define SVO_STEP_ADDR64_MASK "v_mad_u64_u32 %[q64], vcc, %[q6], 8, %[maskp]\n\t" "global_load_dwordx2 %[bm], %[q64], off\n\t"
#define SVO_STEP_ADDR64_MASK_HALVES "global_load_dword %[r3], %[q64], off\n\t" "global_load_dword %[w], %[q64], off offset:4\n\t"
#define SVO_STEP_CPU_LEAF "v_subrev_f32 %[q1], %[eps], %[t]\n\t"
#define SVO_STEP_CPU_GUARD ""
#define SVO_STEP_GLSL_LEAF "v_mov_b32 %[q1], %[t]\n\t"
#define SVO_STEP_GLSL_GUARD "v_cmp_gt_f32 vcc, %[eps], %[r1]\n\t" "v_mov_b32 %[q7], 0x3d800000\n\t" "s_nop 0\n\t" "v_cndmask_b32 %[r1], %[r1], %[q7], vcc\n\t"

#define SVO_STEP_FN march_steps_asm
#define SVO_STEP_WIDE_T uint32_t
#define SVO_STEP_LOAD_ENTRY SVO_STEP_ADDR32_ENTRY
#define SVO_STEP_MASK_OFFSET SVO_STEP_ADDR32_MASKOFF
#define SVO_STEP_LOAD_MASK SVO_STEP_ADDR32_MASK
#define SVO_STEP_LOAD_MASK_HALVES SVO_STEP_ADDR32_MASK_HALVES
#define SVO_STEP_LEAF_DISTANCE SVO_STEP_CPU_LEAF
#define SVO_STEP_ESCAPE_GUARD SVO_STEP_CPU_GUARD
#include "step_asm_body.inc"
#undef SVO_STEP_FN
#undef SVO_STEP_LEAF_DISTANCE
#undef SVO_STEP_ESCAPE_GUARD
#define SVO_STEP_FN march_steps_asm_glsl
#define SVO_STEP_LEAF_DISTANCE SVO_STEP_GLSL_LEAF
#define SVO_STEP_ESCAPE_GUARD SVO_STEP_GLSL_GUARD
#include "step_asm_body.inc"
#undef SVO_STEP_FN
#undef SVO_STEP_WIDE_T
#undef SVO_STEP_LOAD_ENTRY
#undef SVO_STEP_MASK_OFFSET
#undef SVO_STEP_LOAD_MASK
#undef SVO_STEP_LOAD_MASK_HALVES
#define SVO_STEP_FN march_steps_asm_big_glsl
#define SVO_STEP_WIDE_T unsigned long long
#define SVO_STEP_LOAD_ENTRY SVO_STEP_ADDR64_ENTRY
#define SVO_STEP_MASK_OFFSET SVO_STEP_ADDR64_MASKOFF
#define SVO_STEP_LOAD_MASK SVO_STEP_ADDR64_MASK
#define SVO_STEP_LOAD_MASK_HALVES SVO_STEP_ADDR64_MASK_HALVES
#include "step_asm_body.inc"
#undef SVO_STEP_FN
#undef SVO_STEP_LEAF_DISTANCE
#undef SVO_STEP_ESCAPE_GUARD
#define SVO_STEP_FN march_steps_asm_big
#define SVO_STEP_LEAF_DISTANCE SVO_STEP_CPU_LEAF
#define SVO_STEP_ESCAPE_GUARD SVO_STEP_CPU_GUARD
#include "step_asm_body.inc"
#undef SVO_STEP_FN
#undef SVO_STEP_WIDE_T
#undef SVO_STEP_LOAD_ENTRY
#undef SVO_STEP_MASK_OFFSET
#undef SVO_STEP_LOAD_MASK
#undef SVO_STEP_LOAD_MASK_HALVES
#undef SVO_STEP_LEAF_DISTANCE
#undef SVO_STEP_ESCAPE_GUARD

} // namespace svo
