#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_skip_rot.hpp: the two traversal loops of k_render_skip (f32) in gfx950 assembly,
"rotating" flavour.

Same arithmetic, operation for operation, as the C++ loops of rt_skip.hpp (and as the first, hand-written assembly loops,
which this file's output was validated against frame for frame before they were retired); what is generated is the
bookkeeping around it.  A lone wave retires about one instruction
per 6 cycles whatever its type, and a 1080p frame is as long as its longest wave, so every scalar instruction of a step
counts:

  * node positions are BYTE offsets into the stream (i, resume, skip): no shifts before the scalar loads;
  * both successors of a node (i + 32 and skip) are fetched at the top of its step into two of THREE register banks, and
    the step ends by branching into the copy of the loop body whose "current node" bank is the one that holds the
    successor it chose -- no select instructions.  With banks (0, 1, 2) three copies suffice:
        copy A: current 0, next -> 1, skip -> 2      next: B   skip: C
        copy B: current 1, next -> 0, skip -> 2      next: A   skip: C
        copy C: current 2, next -> 0, skip -> 1      next: A   skip: B
  * the commonest step (a BOUND no live lane can hit) falls straight through to its jump: 22 instructions, one taken
    branch (the hand-written loop: 32 and three);
  * a step knows its node's type from the first instruction on (BOUND and ITEM steps are separate bodies), so an ITEM step
    fetches one successor only;
  * FUSED flavour, for scenes in which every group's first child is a sphere concentric with the group's bound (the
    reference's pyramid, group.rs:37-41): a BOUND step goes on to test that sphere for the lanes that enter -- v, b and
    b*b - vv are the same bits, only rr differs -- and the walk continues two nodes on.  One step and eight VALU
    operations fewer per entered group, same tests, same order, same values.

Run:  python3 tools/gen_skip_asm.py   (writes the header; the build does not need this script)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_skip_rot.hpp")

BANKS = {0: 40, 1: 64, 2: 72}                        # first SGPR of each 8-dword node bank
COPIES = {"A": (0, 1, 2), "B": (1, 0, 2), "C": (2, 0, 1)}          # current, next, skip
NEXT_COPY = {"A": "B", "B": "A", "C": "A"}
SKIP_COPY = {"A": "C", "B": "C", "C": "B"}


def bank(b):
    f = BANKS[b]
    return {"w": ["s%d" % (f + k) for k in range(8)], "range": "s[%d:%d]" % (f, f + 7)}


class Asm:
    def __init__(self):
        self.lines = []

    def op(self, text, comment=None):
        self.lines.append(("\t", text, comment))

    def label(self, name):
        self.lines.append(("", name + ":", None))

    def render(self, indent="        "):
        out = []
        for i, (tab, text, comment) in enumerate(self.lines):
            end = "\\n" if tab == "" else "\\n"
            s = '%s"%s%s%s"' % (indent, "" if tab == "" else "\\t", text, end)
            if comment:
                s += "  /* %s */" % comment
            out.append(s)
        return "\n".join(out)


ROOT_CORRECT = [
    ("v_add_u32_e32 %[t0], -1, %[root]", None),
    ("v_add_u32_e32 %[t1], 1, %[root]", None),
    ("v_fma_f32 %[t3], -%[t0], %[root], {x}", "exact residuals of the two neighbours"),
    ("v_fma_f32 %[t4], -%[t1], %[root], {x}", None),
    ("v_cmp_ge_f32_e64 s[56:57], 0, %[t3]", None),
    ("v_cmp_lt_f32_e64 s[58:59], 0, %[t4]", None),
    ("s_nop {nop}", None),
    ("v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]", None),
    ("v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]", None),
]


def emit_root(a, L, need_mask, done_label, tiny_label):
    """Correctly rounded sqrt(disc) into %[root] (== sqrt_rn_lean).  need_mask: SGPR pair of the lanes whose root is used."""
    a.op("v_sqrt_f32_e32 %[root], %[disc]")
    a.op("v_cmp_lt_f32_e64 s[60:61], |%[disc]|, %[tiny]")
    a.op("s_and_b64 s[56:57], s[60:61], %s" % need_mask)
    a.op("s_cbranch_scc1 %s" % tiny_label, "some needed lane below 2^-96: scaled path")
    for t, c in ROOT_CORRECT:
        a.op(t.format(x="%[disc]", nop=0), c)
    a.label(done_label)


def emit_tiny(a, tiny_label, done_label):
    a.label(tiny_label)
    a.op("v_mul_f32_e32 %[t0], 0x4f800000, %[disc]", "root with the 2^32 / 2^-16 scaling for tiny lanes")
    a.op("v_cndmask_b32_e64 %[t5], %[disc], %[t0], s[60:61]")
    a.op("v_sqrt_f32_e32 %[root], %[t5]")
    a.op("s_nop 0")
    for t, c in ROOT_CORRECT:
        a.op(t.format(x="%[t5]", nop=1), None)
    a.op("v_mul_f32_e32 %[t0], 0x37800000, %[root]")
    a.op("v_cndmask_b32_e64 %[root], %[root], %[t0], s[60:61]")
    a.op("s_branch %s" % done_label)


def emit_transitions(a, name, C, lab):
    """skip / next (with end check); `next` uses s51 set at the top of the step."""
    a.label(lab("skip"))
    a.op("s_mov_b32 s48, %s" % C["w"][7], "jump over the subtree")
    a.op("s_cmp_ge_u32 s48, %[n]")
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % SKIP_COPY[name])
    a.label(lab("next"))
    emit_next(a, name)


def emit_next(a, name):
    a.op("s_cmp_ge_u32 s51, %[n]")
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.op("s_mov_b32 s48, s51")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % NEXT_COPY[name])


def primary_terms(a, w):
    a.op("v_mul_f32_e32 %%[t0], %s, %%[dx]" % w[0], "b = (vx*dx + vy*dy) + vz*dz   primitive.rs:57")
    a.op("v_mul_f32_e32 %%[t1], %s, %%[dy]" % w[1])
    a.op("v_mul_f32_e32 %%[t2], %s, %%[dz]" % w[2])
    a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
    a.op("v_add_f32_e32 %[b], %[t0], %[t2]")
    a.op("v_mul_f32_e32 %[t0], %[b], %[b]", "disc = (b*b - vv) + rr   primitive.rs:58")
    a.op("v_subrev_f32_e32 %%[q], %s, %%[t0]" % w[3])
    a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % w[4])


def primary_distance(a):
    """vcc (lanes with disc >= 0) -> vcc = go: t2 >= 0 and d < hit.distance; d left in t4."""
    a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
    a.op("v_sub_f32_e32 %[t4], %[b], %[root]", "t1")
    a.op("v_cmp_lt_f32_e64 s[56:57], 0, %[t4]", "t1 > 0")
    a.op("v_cmp_le_f32_e64 s[58:59], 0, %[t3]", "t2 >= 0")
    a.op("s_and_b64 vcc, vcc, s[58:59]")
    a.op("v_cndmask_b32_e64 %[t4], %[t3], %[t4], s[56:57]", "d = t1 > 0 ? t1 : t2")
    a.op("v_cmp_lt_f32_e64 s[56:57], %[t4], %[best]", "d < hit.distance")
    a.op("s_and_b64 vcc, vcc, s[56:57]", "go")


def primary_item_update(a, item_reg):
    a.op("s_mov_b64 exec, vcc", "primitive.rs:80-83")
    a.op("v_mov_b32_e32 %[best], %[t4]")
    a.op("v_mov_b32_e32 %%[bitem], %s" % item_reg)
    a.op("s_mov_b64 exec, s[62:63]")


def candidates(a, hit_label):
    a.op("v_cmp_ge_u32_e64 s[52:53], s48, %[resume]", "active = i >= resume")
    a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")
    a.op("s_and_b64 vcc, vcc, s[52:53]", "live lanes whose line meets the sphere")
    a.op("s_cbranch_vccnz %s" % hit_label)


def primary_copy(a, name, fused):
    c, n, s = COPIES[name]
    C, N, S = bank(c), bank(n), bank(s)
    w = C["w"]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    a.op("s_cmp_eq_u32 %s, 0" % w[7])
    a.op("s_cbranch_scc1 %s" % lab("item"))
    # ---------------- BOUND step (group.rs:73) ----------------
    a.op("s_add_u32 s51, s48, %d" % (64 if fused else 32), "the walk goes on behind the group's own sphere" if fused else None)
    a.op("s_load_dwordx8 %s, %%[base], s51" % N["range"], "both successors, while this node is processed")
    a.op("s_load_dwordx8 %s, %%[base], %s" % (S["range"], w[7]))
    primary_terms(a, w)
    candidates(a, lab("bhit"))
    emit_transitions(a, name, C, lab)      # nobody can hit the bound: jump (the lanes that culled it are awake again at `skip`)
    a.label(lab("bhit"))
    emit_root(a, lab, "vcc", lab("brooted"), lab("btiny"))
    primary_distance(a)
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("skip"), "nobody enters")
    a.op("s_andn2_b64 exec, s[52:53], vcc", "lanes that may not enter sleep until `skip`")
    a.op("v_mov_b32_e32 %%[resume], %s" % w[7])
    a.op("s_mov_b64 exec, s[62:63]")
    if fused:
        # the group's own sphere, for the lanes that entered: same centre, so v, b and b*b - vv are the values just formed
        a.op("s_mov_b64 s[52:53], vcc", "the lanes that are live at the next node")
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % w[5], "disc = (b*b - vv) + rr of the sphere")
        a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")
        a.op("s_and_b64 vcc, vcc, s[52:53]")
        a.op("s_cbranch_vccz %s" % lab("next"))
        emit_root(a, lab, "vcc", lab("frooted"), lab("ftiny"))
        primary_distance(a)
        primary_item_update(a, w[6])
    a.op("s_branch %s" % lab("next"))
    emit_tiny(a, lab("btiny"), lab("brooted"))
    if fused:
        emit_tiny(a, lab("ftiny"), lab("frooted"))
    # ---------------- ITEM step (primitive.rs:77-84) ----------------
    a.label(lab("item"))
    a.op("s_add_u32 s51, s48, 32")
    a.op("s_load_dwordx8 %s, %%[base], s51" % N["range"])
    primary_terms(a, w)
    candidates(a, lab("ihit"))
    emit_next(a, name)                      # nobody can hit: an ITEM changes nothing
    a.label(lab("ihit"))
    emit_root(a, lab, "vcc", lab("irooted"), lab("itiny"))
    primary_distance(a)
    primary_item_update(a, w[6])
    a.op("s_branch %s" % lab("next"))
    emit_tiny(a, lab("itiny"), lab("irooted"))


def shadow_terms(a, w):
    a.op("v_sub_f32_e32 %%[vx], %s, %%[ox]" % w[0], "v = centre - origin   primitive.rs:56")
    a.op("v_sub_f32_e32 %%[vy], %s, %%[oy]" % w[1])
    a.op("v_sub_f32_e32 %%[vz], %s, %%[oz]" % w[2])
    a.op("v_mul_f32_e32 %[t0], %[lx], %[vx]")
    a.op("v_mul_f32_e32 %[t1], %[ly], %[vy]")
    a.op("v_mul_f32_e32 %[t2], %[lz], %[vz]")
    a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
    a.op("v_add_f32_e32 %[b], %[t0], %[t2]", "b = dot(v, dir)   primitive.rs:57")
    a.op("v_mul_f32_e32 %[t3], %[vx], %[vx]")
    a.op("v_mul_f32_e32 %[t4], %[vy], %[vy]")
    a.op("v_mul_f32_e32 %[t5], %[vz], %[vz]")
    a.op("v_add_f32_e32 %[t3], %[t3], %[t4]")
    a.op("v_add_f32_e32 %[t3], %[t3], %[t5]", "dot(v, v)")
    a.op("v_mul_f32_e32 %[t0], %[b], %[b]")
    a.op("v_sub_f32_e32 %[q], %[t0], %[t3]")
    a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % w[3], "disc = (b*b - vv) + rr   primitive.rs:58")


def shadow_decide(a, lab, tag):
    """vcc (live lanes with disc >= 0) -> vcc = lanes whose ray hits the sphere (t2 >= 0; certain when b >= 0)."""
    a.op("v_cmp_gt_f32_e64 s[54:55], 0, %[b]", "b < 0: t2 may still be negative")
    a.op("s_and_b64 s[54:55], s[54:55], vcc")
    a.op("s_cbranch_scc0 %s" % lab(tag + "decided"), "nobody needs the root: hit = candidates")
    emit_root(a, lab, "s[54:55]", lab(tag + "rooted"), lab(tag + "tiny"))
    a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
    a.op("v_cmp_gt_f32_e64 s[56:57], 0, %[t3]", "t2 < 0")
    a.op("s_and_b64 s[56:57], s[56:57], s[54:55]", "root lanes that miss after all")
    a.op("s_andn2_b64 vcc, vcc, s[56:57]")
    a.label(lab(tag + "decided"))


def shadow_copy(a, name, fused):
    c, n, s = COPIES[name]
    C, N, S = bank(c), bank(n), bank(s)
    w = C["w"]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    a.op("s_cmp_eq_u32 %s, 0" % w[7])
    a.op("s_cbranch_scc1 %s" % lab("item"))
    # ---------------- BOUND step: hit.distance is INF, so a bound culls iff the ray misses it ----------------
    a.op("s_add_u32 s51, s48, %d" % (64 if fused else 32))
    a.op("s_load_dwordx8 %s, %%[base], s51" % N["range"], "both successors, while this node is processed")
    a.op("s_load_dwordx8 %s, %%[base], %s" % (S["range"], w[7]))
    shadow_terms(a, w)
    candidates(a, lab("bhit"))
    emit_transitions(a, name, C, lab)
    a.label(lab("bhit"))
    shadow_decide(a, lab, "b")
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("skip"))
    a.op("s_andn2_b64 exec, s[52:53], vcc", "a lane that misses the bound sleeps until `skip`")
    a.op("v_mov_b32_e32 %%[resume], %s" % w[7])
    a.op("s_mov_b64 exec, s[62:63]")
    if fused:
        a.op("s_mov_b64 s[52:53], vcc", "the lanes that are live at the next node")
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % w[5], "disc = (b*b - vv) + rr of the group's own sphere")
        a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")
        a.op("s_and_b64 vcc, vcc, s[52:53]")
        a.op("s_cbranch_vccz %s" % lab("next"))
        shadow_decide(a, lab, "f")
        a.op("s_cmp_eq_u64 vcc, 0")
        a.op("s_cbranch_scc1 %s" % lab("next"))
        a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc", "any hit ends those rays; hand them to the caller")
        a.op("s_add_u32 %[stop], s48, 32", "they hit the ITEM behind this BOUND")
        a.op("s_branch .Lrt_out_%=")
    else:
        a.op("s_branch %s" % lab("next"))
    emit_tiny(a, lab("btiny"), lab("brooted"))
    if fused:
        emit_tiny(a, lab("ftiny"), lab("frooted"))
    # ---------------- ITEM step ----------------
    a.label(lab("item"))
    a.op("s_add_u32 s51, s48, 32")
    a.op("s_load_dwordx8 %s, %%[base], s51" % N["range"])
    shadow_terms(a, w)
    candidates(a, lab("ihit"))
    emit_next(a, name)
    a.label(lab("ihit"))
    shadow_decide(a, lab, "i")
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("next"))
    a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc", "any hit ends those rays; hand them to the caller")
    a.op("s_mov_b32 %[stop], s48")
    a.op("s_branch .Lrt_out_%=")
    emit_tiny(a, lab("itiny"), lab("irooted"))


HEADER = """// rt_skip_rot.hpp -- GENERATED by tools/gen_skip_asm.py; edit the generator, not this file.
//
// The two traversal loops of k_render_skip (f32) in gfx950 assembly.  Each is the arithmetic of the C++ loop beside it in
// rt_skip.hpp (the reference implementation: f64, every launch that counts tests), operation for operation:
//      b    = (vx*dx + vy*dy) + vz*dz              primitive.rs:57   (node terms as SGPR operands)
//      disc = (b*b - vv) + rr                      primitive.rs:58
//      root = correctly rounded sqrt(disc)         v_sqrt_f32 + two exact FMA residuals (== sqrt_rn_lean, which is checked
//                                                  against the IEEE sqrt on all 2^32 inputs)
//      t2 = b + root, t1 = b - root, d = t1 > 0 ? t1 : t2            primitive.rs:65-71
//      go = live && disc >= 0 && t2 >= 0 && d < hit.distance         (the negation of `d >= hit.distance`, group.rs:73 /
//                                                                      primitive.rs:79, for the NaN-free values a
//                                                                      validated scene produces)
// The root is only formed when some live lane has disc >= 0 (shadow rays: and b < 0, since t2 >= 0 is certain otherwise).
//
// Bookkeeping (why the loops are generated): a lone wave retires about one instruction per 6 cycles whatever its type and
// a 1080p frame is as long as its longest wave, so every scalar instruction of a step counts.  Node positions are byte
// offsets; both successors of a node (the next one and `skip`) are fetched at the top of its step into two of three
// scalar register banks, and the step ends by branching into the copy of the loop body (A, B, C) whose current-node bank
// already holds the chosen successor -- no selects, no shifts, one taken branch for the commonest step.  BOUND and ITEM steps
// are separate bodies.  The *_fused flavour serves scenes in which every BOUND is followed by an ITEM with the same centre
// (the reference's pyramid): the BOUND step goes on to test that sphere for the lanes that enter (v, b, b*b - vv are the
// same bits; only rr differs) and continues two nodes on.
//
// Hazards follow what hipcc itself emits for gfx950: a VALU-written SGPR pair is not read as a v_cndmask mask within the
// next two instructions, a v_sqrt_f32 result is not consumed by the next instruction, s_waitcnt lgkmcnt(0) before loaded
// registers are read and at every exit (the speculative loads must have landed before their registers are free again).
//
// Node<float> words: 0-4 geometry terms, 5 rr of the group's own sphere (fused scenes, BOUND nodes), 6 item, 7 skip as a byte
// offset (0: ITEM).  Fixed SGPRs: s[40:47] / s[64:71] / s[72:79] node banks, s48 current byte offset, s51 next,
// s[52:61] masks, s[62:63] EXEC at entry.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

#define RT_ROT_CLOBBERS                                                                                                        \\
    "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", \\
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72",   \\
        "s73", "s74", "s75", "s76", "s77", "s78", "s79"

"""

PRIMARY_FN = """// Primary-ray traversal: s.group.intersect(&mut h, r) for all 64 rays of the wave.  nodes: Node<float>[n + 2];
// n_bytes = n * 32.  resume: 0 for lanes with a ray, 0xFFFFFFFF for lanes without.  Returns hit.distance / item per lane.
__device__ __forceinline__ void %(name)s(const void *nodes, unsigned n_bytes, float dx, float dy, float dz, unsigned resume,
                                                 float &best_out, unsigned &item_out)
{
    float best = __builtin_huge_valf();
    unsigned bitem = 0;
    float t0, t1, t2, t3, t4, t5, b, q, disc, root;
    const float tiny = 0x1p-96f;
    asm volatile(
%(body)s
        : [best] "+v"(best), [bitem] "+v"(bitem), [resume] "+v"(resume), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [b] "=&v"(b), [q] "=&v"(q), [disc] "=&v"(disc), [root] "=&v"(root)
        : [base] "s"(nodes), [n] "s"(n_bytes), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [tiny] "s"(tiny)
        : RT_ROT_CLOBBERS);
    best_out = best;
    item_out = bitem;
}

"""

SHADOW_FN = """// Shadow-ray traversal (any hit, render.rs:202-208) from byte offset `start` until the stream ends or some lane's ray hits
// an ITEM: the caller retires those lanes, finds the next node any lane still wants and calls again.  Returns the byte
// offset it stopped at (>= n_bytes: stream finished); fin = 1 in the lanes that hit the ITEM there.  resume in bytes.
// hit.distance is INF throughout, so a node is "hit" iff disc >= 0 and t2 = b + root >= 0.
__device__ __forceinline__ unsigned %(name)s(const void *nodes, unsigned n_bytes, unsigned start, float ox, float oy, float oz,
                                                   float lx, float ly, float lz, unsigned &resume_io, unsigned &fin_out)
{
    unsigned resume = resume_io, fin = 0, stop;
    float t0, t1, t2, t3, t4, t5, vx, vy, vz, b, q, disc, root;
    const float tiny = 0x1p-96f;
    asm volatile(
%(body)s
        : [resume] "+v"(resume), [fin] "+v"(fin), [stop] "=&s"(stop), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [t4] "=&v"(t4), [t5] "=&v"(t5), [vx] "=&v"(vx), [vy] "=&v"(vy), [vz] "=&v"(vz), [b] "=&v"(b), [q] "=&v"(q),
          [disc] "=&v"(disc), [root] "=&v"(root)
        : [base] "s"(nodes), [n] "s"(n_bytes), [start] "s"(start), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "s"(lx), [ly] "s"(ly),
          [lz] "s"(lz), [tiny] "s"(tiny)
        : RT_ROT_CLOBBERS);
    resume_io = resume;
    fin_out = fin;
    return stop;
}

"""


def primary(fused):
    a = Asm()
    a.op("s_mov_b32 s48, 0")
    a.op("s_mov_b64 s[62:63], exec")
    a.op("s_load_dwordx8 s[40:47], %[base], 0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        primary_copy(a, name, fused)
    a.label(".Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def shadow(fused):
    a = Asm()
    a.op("s_mov_b32 s48, %[start]")
    a.op("s_mov_b64 s[62:63], exec")
    a.op("s_load_dwordx8 s[40:47], %[base], s48")
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        shadow_copy(a, name, fused)
    a.label(".Lrt_exit_%=")
    a.op("s_mov_b32 %[stop], %[n]", "stream finished")
    a.label(".Lrt_out_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def main():
    text = HEADER
    for fused in (False, True):
        sfx = "_fused" if fused else ""
        text += PRIMARY_FN % {"name": "skip_primary_rot" + sfx, "body": primary(fused)}
        text += SHADOW_FN % {"name": "skip_shadow_rot" + sfx, "body": shadow(fused)}
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
