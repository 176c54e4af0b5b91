// rt_skip_asm.hpp -- the two traversal loops of k_render_skip (f32) written by hand in gfx950 assembly.
//
// Why: a 1080p frame is bounded by its heaviest wave, and a lone wave retires about one instruction per 4-5 cycles,
// so the only lever is the instruction count of a traversal step.  hipcc's step is ~75 instructions (IEEE sqrt
// expansion, an INF select chain, ballot materialisation, 64-bit address arithmetic, a 4-instruction loop exit);
// the hand-written step is ~45 with the same arithmetic, operation for operation:
//      b    = (a0*dx + a1*dy) + a2*dz              primitive.rs:57   (5 VALU, node fields as SGPR operands)
//      disc = (b*b - a3) + a4                       primitive.rs:58   (3 VALU)
//      root = correctly rounded sqrt(disc)          v_sqrt_f32 + two exact FMA residuals (== sqrt_rn_lean, which is
//                                                   checked against the IEEE sqrt on all 2^32 inputs)
//      t2 = b + root, t1 = b - root, d = t1 > 0 ? t1 : t2            primitive.rs:65-71
//      go = active && disc >= 0 && t2 >= 0 && d < hit.distance       (the negation of `d >= hit.distance`, group.rs:73 /
//                                                                      primitive.rs:79, for the NaN-free values a
//                                                                      validated scene produces)
// When no active lane has disc >= 0 the whole root block is skipped (every such lane "misses").
//
// Hazards follow what hipcc itself emits for gfx950: a VALU-written SGPR/VCC is not read as a v_cndmask mask within
// the next two instructions, a v_sqrt_f32 result is not consumed by the next instruction, every s_load is followed
// by s_waitcnt lgkmcnt(0) before its registers are read.  Fixed SGPRs s40-s61 are declared clobbered.
//
// The C++ loops in rt_skip.hpp remain the reference implementation (f64, and every launch that counts tests);
// tools/ab.py checks this variant's frames against them, and the parity tests compare it with the CPU restatement.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

// Step prologue / epilogue flavours.  RT_PF_* add a speculative fetch of BOTH possible successors (node i+1 and node
// `skip`, into s[64:79]) at the top of the step and select one at the bottom: 8 more scalar instructions per step, but
// the dependent scalar-load latency leaves the wave's critical path.  It pays for a lone long-running wave (the frame's
// critical path at 1080p) and costs throughput under load.
#define RT_TOP_PLAIN ""
#define RT_NEXT_PLAIN                                       \
    "s_lshl_b32 s50, s49, 5\n\t"                            \
    "s_load_dwordx8 s[40:47], %[base], s50\n\t"             \
    "s_mov_b32 s48, s49\n\t"                                \
    "s_waitcnt lgkmcnt(0)\n\t"
#define RT_TOP_PF                                           \
    "s_add_u32 s51, s48, 1\n\t"                             \
    "s_lshl_b32 s50, s51, 5\n\t"                            \
    "s_load_dwordx8 s[64:71], %[base], s50\n\t"             \
    "s_lshl_b32 s50, s45, 5\n\t"                            \
    "s_load_dwordx8 s[72:79], %[base], s50\n\t"
#define RT_NEXT_PF                                          \
    "s_cmp_eq_u32 s49, s51\n\t"                             \
    "s_mov_b32 s48, s49\n\t"                                \
    "s_waitcnt lgkmcnt(0)\n\t"                              \
    "s_cselect_b64 s[40:41], s[64:65], s[72:73]\n\t"        \
    "s_cselect_b64 s[42:43], s[66:67], s[74:75]\n\t"        \
    "s_cselect_b64 s[44:45], s[68:69], s[76:77]\n\t"        \
    "s_cselect_b64 s[46:47], s[70:71], s[78:79]\n\t"
#define RT_SKIP_CLOBBERS                                                                                                       \
    "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", \
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74",   \
        "s75", "s76", "s77", "s78", "s79"


// Node-update flavours.  V1 selects with v_cndmask (a VGPR copy of the SGPR operand + one select per register); V2 narrows
// EXEC to the lanes that change and moves (one VALU op fewer per BOUND and per ITEM step), and leaves `resume` alone when the
// wave jumps over a subtree nobody enters: the lanes that culled it are awake again at `skip` either way (resume <= i < skip).
#define RT_BOUND_UPD_V1                                                   \
        "s_andn2_b64 s[56:57], s[52:53], vcc\n\t"                         \
        "v_mov_b32_e32 %[t5], s45\n\t"                                    \
        "s_cmp_eq_u64 vcc, 0\n\t"                                         \
        "s_cselect_b32 s49, s45, s49\n\t"                                 \
        "v_cndmask_b32_e64 %[resume], %[resume], %[t5], s[56:57]\n\t"
#define RT_BOUND_UPD_V2                                                   \
        "s_cmp_eq_u64 vcc, 0\n\t"                                         \
        "s_cselect_b32 s49, s45, s49\n\t"                                 \
        "s_andn2_b64 exec, s[52:53], vcc\n\t"  /* lanes that may not enter */ \
        "v_mov_b32_e32 %[resume], s45\n\t"                                \
        "s_mov_b64 exec, s[62:63]\n\t"
#define RT_ITEM_UPD_V1                                                    \
        "v_mov_b32_e32 %[t5], s46\n\t"                                    \
        "v_cndmask_b32_e32 %[best], %[best], %[t4], vcc\n\t"              \
        "v_cndmask_b32_e32 %[bitem], %[bitem], %[t5], vcc\n"
#define RT_ITEM_UPD_V2                                                    \
        "s_mov_b64 exec, vcc\n\t"                                         \
        "v_mov_b32_e32 %[best], %[t4]\n\t"                                \
        "v_mov_b32_e32 %[bitem], s46\n\t"                                 \
        "s_mov_b64 exec, s[62:63]\n"
#define RT_MISS_BOUND_V1                                                  \
        "v_mov_b32_e32 %[t5], s45\n\t"                                    \
        "s_mov_b32 s49, s45\n\t"                                          \
        "v_cndmask_b32_e64 %[resume], %[resume], %[t5], s[52:53]\n\t"
#define RT_MISS_BOUND_V2                                                  \
        "s_mov_b32 s49, s45\n\t"

#define RT_PRIMARY_ASM(TOP, NEXT, BOUND_UPD, ITEM_UPD, MISS_BOUND) \
        "s_mov_b32 s48, 0\n\t" \
        "s_mov_b64 s[62:63], exec\n\t" \
        "s_load_dwordx8 s[40:47], %[base], 0x0\n\t" \
        "s_waitcnt lgkmcnt(0)\n" \
        "1:\n\t" \
        TOP \
  /* ---- b, disc ---- */ \
        "v_mul_f32_e32 %[t0], s40, %[dx]\n\t" \
        "v_mul_f32_e32 %[t1], s41, %[dy]\n\t" \
        "v_mul_f32_e32 %[t2], s42, %[dz]\n\t" \
        "v_add_f32_e32 %[t0], %[t0], %[t1]\n\t" \
        "v_add_f32_e32 %[b], %[t0], %[t2]\n\t" \
        "v_mul_f32_e32 %[t0], %[b], %[b]\n\t" \
        "v_subrev_f32_e32 %[t0], s43, %[t0]\n\t" \
        "v_add_f32_e32 %[disc], s44, %[t0]\n\t" \
        "v_cmp_ge_u32_e64 s[52:53], s48, %[resume]\n\t"  /* active = i >= resume */ \
        "v_cmp_le_f32_e32 vcc, 0, %[disc]\n\t"  /* disc >= 0 */ \
        "s_and_b64 vcc, vcc, s[52:53]\n\t"  /* lanes that need the exact distance */ \
        "s_cbranch_vccz 3f\n\t"  /* no active lane can hit this node */ \
  /* ---- correctly rounded root ---- */ \
        "v_sqrt_f32_e32 %[root], %[disc]\n\t" \
        "v_cmp_lt_f32_e64 s[60:61], |%[disc]|, %[tiny]\n\t" \
        "s_and_b64 s[56:57], s[60:61], vcc\n\t" \
        "s_cbranch_scc1 9f\n\t"  /* some needed lane below 2^-96: scaled path */ \
        "v_add_u32_e32 %[t0], -1, %[root]\n\t" \
        "v_add_u32_e32 %[t1], 1, %[root]\n\t" \
        "v_fma_f32 %[t3], -%[t0], %[root], %[disc]\n\t" \
        "v_fma_f32 %[t4], -%[t1], %[root], %[disc]\n\t" \
        "v_cmp_ge_f32_e64 s[56:57], 0, %[t3]\n\t" \
        "v_cmp_lt_f32_e64 s[58:59], 0, %[t4]\n\t" \
        "s_nop 0\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]\n" \
        "8:\n\t" \
  /* ---- distance, decision ---- */ \
        "v_add_f32_e32 %[t3], %[b], %[root]\n\t"  /* t2 */ \
        "v_sub_f32_e32 %[t4], %[b], %[root]\n\t"  /* t1 */ \
        "v_cmp_lt_f32_e64 s[56:57], 0, %[t4]\n\t"  /* t1 > 0 */ \
        "v_cmp_le_f32_e64 s[58:59], 0, %[t3]\n\t"  /* t2 >= 0 */ \
        "s_and_b64 vcc, vcc, s[58:59]\n\t" \
        "v_cndmask_b32_e64 %[t4], %[t3], %[t4], s[56:57]\n\t"  /* d = t1 > 0 ? t1 : t2 */ \
        "v_cmp_lt_f32_e64 s[56:57], %[t4], %[best]\n\t"  /* d < hit.distance */ \
        "s_and_b64 vcc, vcc, s[56:57]\n"  /* go */ \
        "4:\n\t" \
        "s_add_u32 s49, s48, 1\n\t" \
        "s_cmp_eq_u32 s45, 0\n\t" \
        "s_cbranch_scc1 5f\n\t" \
  /* ---- BOUND (group.rs:73): lanes that may not enter sleep until `skip`; jump if nobody enters ---- */ \
        BOUND_UPD \
        "s_cmp_ge_u32 s49, %[n]\n\t"  /* own copy of the loop tail: one taken branch per BOUND step */ \
        "s_cbranch_scc1 7f\n\t" \
        NEXT \
        "s_branch 1b\n" \
        "5:\n\t" \
  /* ---- ITEM (primitive.rs:78-83) ---- */ \
        ITEM_UPD \
        "6:\n\t" \
        "s_cmp_ge_u32 s49, %[n]\n\t" \
        "s_cbranch_scc1 7f\n\t" \
        NEXT \
        "s_branch 1b\n" \
        "3:\n\t"  /* nobody can hit: an ITEM changes nothing; a BOUND puts every active lane to sleep and the wave jumps */ \
        "s_add_u32 s49, s48, 1\n\t" \
        "s_cmp_eq_u32 s45, 0\n\t" \
        "s_cbranch_scc1 6b\n\t" \
        MISS_BOUND \
        "s_branch 6b\n" \
        "9:\n\t"  /* root with the 2^32 / 2^-16 scaling for tiny lanes */ \
        "v_mul_f32_e32 %[t0], 0x4f800000, %[disc]\n\t" \
        "v_cndmask_b32_e64 %[t5], %[disc], %[t0], s[60:61]\n\t" \
        "v_sqrt_f32_e32 %[root], %[t5]\n\t" \
        "s_nop 0\n\t" \
        "v_add_u32_e32 %[t0], -1, %[root]\n\t" \
        "v_add_u32_e32 %[t1], 1, %[root]\n\t" \
        "v_fma_f32 %[t3], -%[t0], %[root], %[t5]\n\t" \
        "v_fma_f32 %[t4], -%[t1], %[root], %[t5]\n\t" \
        "v_cmp_ge_f32_e64 s[56:57], 0, %[t3]\n\t" \
        "v_cmp_lt_f32_e64 s[58:59], 0, %[t4]\n\t" \
        "s_nop 1\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]\n\t" \
        "v_mul_f32_e32 %[t0], 0x37800000, %[root]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[60:61]\n\t" \
        "s_branch 8b\n" \
        "7:\n\t"  /* every exit: speculative loads must have landed before their registers are free again */ \
        "s_waitcnt lgkmcnt(0)\n\t"

// Primary-ray traversal: s.group.intersect(&mut h, r) for all 64 rays of the wave.  nodes: Node<float>[n + 1].
// resume: 0 for lanes with a ray, 0xFFFFFFFF for lanes without.  Returns hit.distance / item index per lane.
template <bool PF, bool V2>
__device__ __forceinline__ void skip_primary_asm(const void *nodes, unsigned n, float dx, float dy, float dz, unsigned resume,
                                                 float &best_out, unsigned &item_out)
{
    float best = __builtin_huge_valf();
    unsigned bitem = 0;
    float t0, t1, t2, t3, t4, t5, b, disc, root;
    const float tiny = 0x1p-96f;
#define RT_PRIMARY_OPERANDS                                                                                                    \
    : [best] "+v"(best), [bitem] "+v"(bitem), [resume] "+v"(resume), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),          \
      [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [b] "=&v"(b), [disc] "=&v"(disc), [root] "=&v"(root)                   \
    : [base] "s"(nodes), [n] "s"(n), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [tiny] "s"(tiny)                               \
    : RT_SKIP_CLOBBERS
    if constexpr (PF && V2) asm volatile(RT_PRIMARY_ASM(RT_TOP_PF, RT_NEXT_PF, RT_BOUND_UPD_V2, RT_ITEM_UPD_V2, RT_MISS_BOUND_V2) RT_PRIMARY_OPERANDS);
    else if constexpr (PF) asm volatile(RT_PRIMARY_ASM(RT_TOP_PF, RT_NEXT_PF, RT_BOUND_UPD_V1, RT_ITEM_UPD_V1, RT_MISS_BOUND_V1) RT_PRIMARY_OPERANDS);
    else if constexpr (V2) asm volatile(RT_PRIMARY_ASM(RT_TOP_PLAIN, RT_NEXT_PLAIN, RT_BOUND_UPD_V2, RT_ITEM_UPD_V2, RT_MISS_BOUND_V2) RT_PRIMARY_OPERANDS);
    else asm volatile(RT_PRIMARY_ASM(RT_TOP_PLAIN, RT_NEXT_PLAIN, RT_BOUND_UPD_V1, RT_ITEM_UPD_V1, RT_MISS_BOUND_V1) RT_PRIMARY_OPERANDS);
#undef RT_PRIMARY_OPERANDS
    best_out = best;
    item_out = bitem;
}

#define RT_SHADOW_ASM(TOP, NEXT, BOUND_UPD, MISS_BOUND) \
        "s_mov_b32 s48, %[start]\n\t" \
        "s_mov_b64 s[62:63], exec\n\t" \
        "s_lshl_b32 s50, s48, 5\n\t" \
        "s_load_dwordx8 s[40:47], %[base], s50\n\t" \
        "s_waitcnt lgkmcnt(0)\n" \
        "1:\n\t" \
        TOP \
        "v_sub_f32_e32 %[vx], s40, %[ox]\n\t"  /* v = centre - origin   primitive.rs:56 */ \
        "v_sub_f32_e32 %[vy], s41, %[oy]\n\t" \
        "v_sub_f32_e32 %[vz], s42, %[oz]\n\t" \
        "v_mul_f32_e32 %[t0], %[lx], %[vx]\n\t" \
        "v_mul_f32_e32 %[t1], %[ly], %[vy]\n\t" \
        "v_mul_f32_e32 %[t2], %[lz], %[vz]\n\t" \
        "v_add_f32_e32 %[t0], %[t0], %[t1]\n\t" \
        "v_add_f32_e32 %[b], %[t0], %[t2]\n\t"  /* b = dot(v, dir)       primitive.rs:57 */ \
        "v_mul_f32_e32 %[t3], %[vx], %[vx]\n\t" \
        "v_mul_f32_e32 %[t4], %[vy], %[vy]\n\t" \
        "v_mul_f32_e32 %[t5], %[vz], %[vz]\n\t" \
        "v_add_f32_e32 %[t3], %[t3], %[t4]\n\t" \
        "v_add_f32_e32 %[t3], %[t3], %[t5]\n\t"  /* dot(v, v) */ \
        "v_mul_f32_e32 %[t0], %[b], %[b]\n\t" \
        "v_sub_f32_e32 %[t0], %[t0], %[t3]\n\t" \
        "v_add_f32_e32 %[disc], s43, %[t0]\n\t"  /* disc = (b*b - vv) + rr   primitive.rs:58 */ \
        "v_cmp_ge_u32_e64 s[52:53], s48, %[resume]\n\t"  /* active = i >= resume */ \
        "v_cmp_le_f32_e32 vcc, 0, %[disc]\n\t" \
        "s_and_b64 vcc, vcc, s[52:53]\n\t"  /* candidates: active, disc >= 0 */ \
        "s_cbranch_vccz 3f\n\t"  /* no active lane can hit this node */ \
        "v_cmp_gt_f32_e64 s[54:55], 0, %[b]\n\t"  /* b < 0: t2 may still be negative */ \
        "s_and_b64 s[54:55], s[54:55], vcc\n\t" \
        "s_cbranch_scc0 4f\n\t"  /* nobody needs the root: hit = candidates */ \
        "v_sqrt_f32_e32 %[root], %[disc]\n\t" \
        "v_cmp_lt_f32_e64 s[60:61], |%[disc]|, %[tiny]\n\t" \
        "s_and_b64 s[56:57], s[60:61], s[54:55]\n\t" \
        "s_cbranch_scc1 9f\n\t" \
        "v_add_u32_e32 %[t0], -1, %[root]\n\t" \
        "v_add_u32_e32 %[t1], 1, %[root]\n\t" \
        "v_fma_f32 %[t3], -%[t0], %[root], %[disc]\n\t" \
        "v_fma_f32 %[t4], -%[t1], %[root], %[disc]\n\t" \
        "v_cmp_ge_f32_e64 s[56:57], 0, %[t3]\n\t" \
        "v_cmp_lt_f32_e64 s[58:59], 0, %[t4]\n\t" \
        "s_nop 0\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]\n" \
        "8:\n\t" \
        "v_add_f32_e32 %[t3], %[b], %[root]\n\t"  /* t2 */ \
        "v_cmp_gt_f32_e64 s[56:57], 0, %[t3]\n\t"  /* t2 < 0 */ \
        "s_and_b64 s[56:57], s[56:57], s[54:55]\n\t"  /* root lanes that miss after all */ \
        "s_andn2_b64 vcc, vcc, s[56:57]\n" \
        "4:\n\t"  /* vcc = lanes whose ray hits this node */ \
        "s_add_u32 s49, s48, 1\n\t" \
        "s_cmp_eq_u32 s45, 0\n\t" \
        "s_cbranch_scc1 5f\n\t" \
  /* ---- BOUND: a lane that misses the bound sleeps until `skip`; jump if nobody enters ---- */ \
        BOUND_UPD \
        "s_cmp_ge_u32 s49, %[n]\n\t"  /* own copy of the loop tail: one taken branch per BOUND step */ \
        "s_cbranch_scc1 10f\n\t" \
        NEXT \
        "s_branch 1b\n" \
        "5:\n\t" \
  /* ---- ITEM: any hit ends those rays; hand them to the caller ---- */ \
        "s_cmp_eq_u64 vcc, 0\n\t" \
        "s_cbranch_scc1 6f\n\t" \
        "v_cndmask_b32_e64 %[fin], 0, 1, vcc\n\t" \
        "s_mov_b32 %[stop], s48\n\t" \
        "s_branch 7f\n" \
        "6:\n\t" \
        "s_cmp_ge_u32 s49, %[n]\n\t" \
        "s_cbranch_scc1 10f\n\t" \
        NEXT \
        "s_branch 1b\n" \
        "3:\n\t"  /* nobody can hit: an ITEM changes nothing; a BOUND puts every active lane to sleep and the wave jumps */ \
        "s_add_u32 s49, s48, 1\n\t" \
        "s_cmp_eq_u32 s45, 0\n\t" \
        "s_cbranch_scc1 6b\n\t" \
        MISS_BOUND \
        "s_branch 6b\n" \
        "9:\n\t"  /* root with the 2^32 / 2^-16 scaling for tiny lanes */ \
        "v_mul_f32_e32 %[t0], 0x4f800000, %[disc]\n\t" \
        "v_cndmask_b32_e64 %[t5], %[disc], %[t0], s[60:61]\n\t" \
        "v_sqrt_f32_e32 %[root], %[t5]\n\t" \
        "s_nop 0\n\t" \
        "v_add_u32_e32 %[t0], -1, %[root]\n\t" \
        "v_add_u32_e32 %[t1], 1, %[root]\n\t" \
        "v_fma_f32 %[t3], -%[t0], %[root], %[t5]\n\t" \
        "v_fma_f32 %[t4], -%[t1], %[root], %[t5]\n\t" \
        "v_cmp_ge_f32_e64 s[56:57], 0, %[t3]\n\t" \
        "v_cmp_lt_f32_e64 s[58:59], 0, %[t4]\n\t" \
        "s_nop 1\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]\n\t" \
        "v_mul_f32_e32 %[t0], 0x37800000, %[root]\n\t" \
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[60:61]\n\t" \
        "s_branch 8b\n" \
        "10:\n\t" \
        "s_mov_b32 %[stop], s49\n" \
        "7:\n\t"  /* every exit: speculative loads must have landed before their registers are free again */ \
        "s_waitcnt lgkmcnt(0)\n\t"

// Shadow-ray traversal (any hit, render.rs:202-208) from node `start` until the stream ends or some lane's ray hits an
// ITEM: the caller retires those lanes, finds the next node any lane still wants and calls again (at most 64 times per
// wave).  hit.distance is INF throughout, so a node is "hit" iff disc >= 0 and t2 = b + root >= 0 -- and t2 >= 0 is
// certain when b >= 0, so the root is only formed when some candidate lane has b < 0.
// Returns the index it stopped at (>= n: stream finished); fin = 1 in the lanes that hit the ITEM at that index.
template <bool PF, bool V2>
__device__ __forceinline__ unsigned skip_shadow_asm(const void *nodes, unsigned n, unsigned start, float ox, float oy, float oz, float lx,
                                                   float ly, float lz, unsigned &resume_io, unsigned &fin_out)
{
    unsigned resume = resume_io, fin = 0, stop;
    float t0, t1, t2, t3, t4, t5, vx, vy, vz, b, disc, root;
    const float tiny = 0x1p-96f;
#define RT_SHADOW_OPERANDS                                                                                                     \
    : [resume] "+v"(resume), [fin] "+v"(fin), [stop] "=&s"(stop), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3), \
      [t4] "=&v"(t4), [t5] "=&v"(t5), [vx] "=&v"(vx), [vy] "=&v"(vy), [vz] "=&v"(vz), [b] "=&v"(b), [disc] "=&v"(disc),       \
      [root] "=&v"(root)                                                                                                       \
    : [base] "s"(nodes), [n] "s"(n), [start] "s"(start), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "s"(lx), [ly] "s"(ly), \
      [lz] "s"(lz), [tiny] "s"(tiny)                                                                                           \
    : RT_SKIP_CLOBBERS
    if constexpr (PF && V2) asm volatile(RT_SHADOW_ASM(RT_TOP_PF, RT_NEXT_PF, RT_BOUND_UPD_V2, RT_MISS_BOUND_V2) RT_SHADOW_OPERANDS);
    else if constexpr (PF) asm volatile(RT_SHADOW_ASM(RT_TOP_PF, RT_NEXT_PF, RT_BOUND_UPD_V1, RT_MISS_BOUND_V1) RT_SHADOW_OPERANDS);
    else if constexpr (V2) asm volatile(RT_SHADOW_ASM(RT_TOP_PLAIN, RT_NEXT_PLAIN, RT_BOUND_UPD_V2, RT_MISS_BOUND_V2) RT_SHADOW_OPERANDS);
    else asm volatile(RT_SHADOW_ASM(RT_TOP_PLAIN, RT_NEXT_PLAIN, RT_BOUND_UPD_V1, RT_MISS_BOUND_V1) RT_SHADOW_OPERANDS);
#undef RT_SHADOW_OPERANDS
    resume_io = resume;
    fin_out = fin;
    return stop;
}

}  // namespace rt
