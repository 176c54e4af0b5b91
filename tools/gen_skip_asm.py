#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_skip_rot.hpp: the two traversal loops of k_render_skip (f32) in gfx950 assembly,
"rotating" flavour.

Same arithmetic, operation for operation, as the hand-written loops of rt_skip_asm.hpp (which this file's output was
validated against, frame for frame); what changes is the bookkeeping around it.  A lone wave retires about one instruction
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
    branch (the hand-written loop: 32 and three).

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


def emit_top(a, N, S, C):
    a.op("s_add_u32 s51, s48, 32")
    a.op("s_load_dwordx8 %s, %%[base], s51" % N["range"], "both successors, while this node is processed")
    a.op("s_load_dwordx8 %s, %%[base], %s" % (S["range"], C["w"][7]))


def emit_transitions(a, name, C, lab):
    """next (with / without end check) and skip."""
    a.label(lab("skip"))
    a.op("s_mov_b32 s48, %s" % C["w"][7], "jump over the subtree")
    a.op("s_cmp_ge_u32 s48, %[n]")
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % SKIP_COPY[name])
    a.label(lab("next"))
    a.op("s_cmp_ge_u32 s51, %[n]")
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.label(lab("next_nc"))
    a.op("s_mov_b32 s48, s51")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % NEXT_COPY[name])


def primary_copy(a, name):
    c, n, s = COPIES[name]
    C, N, S = bank(c), bank(n), bank(s)
    w = C["w"]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    emit_top(a, N, S, C)
    a.op("v_mul_f32_e32 %%[t0], %s, %%[dx]" % w[0], "b = (vx*dx + vy*dy) + vz*dz   primitive.rs:57")
    a.op("v_mul_f32_e32 %%[t1], %s, %%[dy]" % w[1])
    a.op("v_mul_f32_e32 %%[t2], %s, %%[dz]" % w[2])
    a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
    a.op("v_add_f32_e32 %[b], %[t0], %[t2]")
    a.op("v_mul_f32_e32 %[t0], %[b], %[b]", "disc = (b*b - vv) + rr   primitive.rs:58")
    a.op("v_subrev_f32_e32 %%[t0], %s, %%[t0]" % w[3])
    a.op("v_add_f32_e32 %%[disc], %s, %%[t0]" % w[4])
    a.op("v_cmp_ge_u32_e64 s[52:53], s48, %[resume]", "active = i >= resume")
    a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")
    a.op("s_and_b64 vcc, vcc, s[52:53]", "lanes that need the exact distance")
    a.op("s_cbranch_vccnz %s" % lab("hit"))
    # nobody can hit: an ITEM changes nothing; a BOUND is jumped over (the lanes that culled it are awake again at `skip`)
    a.op("s_cmp_eq_u32 %s, 0" % w[7])
    a.op("s_cbranch_scc1 %s" % lab("next"))
    emit_transitions(a, name, C, lab)
    a.label(lab("hit"))
    emit_root(a, lab, "vcc", lab("rooted"), lab("tiny"))
    a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
    a.op("v_sub_f32_e32 %[t4], %[b], %[root]", "t1")
    a.op("v_cmp_lt_f32_e64 s[56:57], 0, %[t4]", "t1 > 0")
    a.op("v_cmp_le_f32_e64 s[58:59], 0, %[t3]", "t2 >= 0")
    a.op("s_and_b64 vcc, vcc, s[58:59]")
    a.op("v_cndmask_b32_e64 %[t4], %[t3], %[t4], s[56:57]", "d = t1 > 0 ? t1 : t2")
    a.op("v_cmp_lt_f32_e64 s[56:57], %[t4], %[best]", "d < hit.distance")
    a.op("s_and_b64 vcc, vcc, s[56:57]", "go")
    a.op("s_cmp_eq_u32 %s, 0" % w[7])
    a.op("s_cbranch_scc1 %s" % lab("item"))
    a.op("s_cmp_eq_u64 vcc, 0", "BOUND (group.rs:73)")
    a.op("s_cbranch_scc1 %s" % lab("skip"), "nobody enters")
    a.op("s_andn2_b64 exec, s[52:53], vcc", "lanes that may not enter sleep until `skip`")
    a.op("v_mov_b32_e32 %%[resume], %s" % w[7])
    a.op("s_mov_b64 exec, s[62:63]")
    a.op("s_branch %s" % lab("next_nc"), "an entered subtree is not empty: no end check")
    a.label(lab("item"))
    a.op("s_mov_b64 exec, vcc", "ITEM (primitive.rs:78-83)")
    a.op("v_mov_b32_e32 %[best], %[t4]")
    a.op("v_mov_b32_e32 %%[bitem], %s" % w[6])
    a.op("s_mov_b64 exec, s[62:63]")
    a.op("s_branch %s" % lab("next"))
    emit_tiny(a, lab("tiny"), lab("rooted"))


def shadow_copy(a, name):
    c, n, s = COPIES[name]
    C, N, S = bank(c), bank(n), bank(s)
    w = C["w"]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    emit_top(a, N, S, C)
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
    a.op("v_sub_f32_e32 %[t0], %[t0], %[t3]")
    a.op("v_add_f32_e32 %%[disc], %s, %%[t0]" % w[3], "disc = (b*b - vv) + rr   primitive.rs:58")
    a.op("v_cmp_ge_u32_e64 s[52:53], s48, %[resume]", "active = i >= resume")
    a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")
    a.op("s_and_b64 vcc, vcc, s[52:53]", "candidates: active, disc >= 0")
    a.op("s_cbranch_vccnz %s" % lab("hit"))
    a.op("s_cmp_eq_u32 %s, 0" % w[7])
    a.op("s_cbranch_scc1 %s" % lab("next"))
    emit_transitions(a, name, C, lab)
    a.label(lab("hit"))
    a.op("v_cmp_gt_f32_e64 s[54:55], 0, %[b]", "b < 0: t2 may still be negative")
    a.op("s_and_b64 s[54:55], s[54:55], vcc")
    a.op("s_cbranch_scc0 %s" % lab("decided"), "nobody needs the root: hit = candidates")
    emit_root(a, lab, "s[54:55]", lab("rooted"), lab("tiny"))
    a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
    a.op("v_cmp_gt_f32_e64 s[56:57], 0, %[t3]", "t2 < 0")
    a.op("s_and_b64 s[56:57], s[56:57], s[54:55]", "root lanes that miss after all")
    a.op("s_andn2_b64 vcc, vcc, s[56:57]")
    a.label(lab("decided"))
    a.op("s_cmp_eq_u32 %s, 0" % w[7], "vcc = lanes whose ray hits this node")
    a.op("s_cbranch_scc1 %s" % lab("item"))
    a.op("s_cmp_eq_u64 vcc, 0", "BOUND: a lane that misses the bound sleeps until `skip`")
    a.op("s_cbranch_scc1 %s" % lab("skip"))
    a.op("s_andn2_b64 exec, s[52:53], vcc")
    a.op("v_mov_b32_e32 %%[resume], %s" % w[7])
    a.op("s_mov_b64 exec, s[62:63]")
    a.op("s_branch %s" % lab("next_nc"))
    a.label(lab("item"))
    a.op("s_cmp_eq_u64 vcc, 0", "ITEM: any hit ends those rays; hand them to the caller")
    a.op("s_cbranch_scc1 %s" % lab("next"))
    a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc")
    a.op("s_mov_b32 %[stop], s48")
    a.op("s_branch .Lrt_out_%=")
    emit_tiny(a, lab("tiny"), lab("rooted"))


HEADER = '''// rt_skip_rot.hpp -- GENERATED by tools/gen_skip_asm.py; edit the generator, not this file.
//
// The two traversal loops of k_render_skip (f32) in gfx950 assembly, rotating flavour: node positions are byte offsets,
// both successors of a node are prefetched into two of three scalar register banks, and a step ends by branching into the
// copy of the loop body whose current-node bank already holds the chosen successor.  The arithmetic is, operation for
// operation, that of rt_skip_asm.hpp (see there for the derivation, the hazards observed and the reference lines).
//
// Node<float> words: 0-4 geometry terms, 5 skip (index, unused here), 6 item, 7 skip as a byte offset (0: ITEM).
// Fixed SGPRs: s[40:47] / s[64:71] / s[72:79] node banks, s48 current byte offset, s51 next, s[52:61] masks, s[62:63] EXEC.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

#define RT_ROT_CLOBBERS                                                                                                        \\
    "memory", "vcc", "scc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", \\
        "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72",   \\
        "s73", "s74", "s75", "s76", "s77", "s78", "s79"

'''

PRIMARY_FN = '''// Primary-ray traversal: s.group.intersect(&mut h, r) for all 64 rays of the wave.  nodes: Node<float>[n + 1];
// n_bytes = n * 32.  resume: 0 for lanes with a ray, 0xFFFFFFFF for lanes without.  Returns hit.distance / item per lane.
__device__ __forceinline__ void skip_primary_rot(const void *nodes, unsigned n_bytes, float dx, float dy, float dz, unsigned resume,
                                                 float &best_out, unsigned &item_out)
{
    float best = __builtin_huge_valf();
    unsigned bitem = 0;
    float t0, t1, t2, t3, t4, t5, b, disc, root;
    const float tiny = 0x1p-96f;
    asm volatile(
%s
        : [best] "+v"(best), [bitem] "+v"(bitem), [resume] "+v"(resume), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),
          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), [b] "=&v"(b), [disc] "=&v"(disc), [root] "=&v"(root)
        : [base] "s"(nodes), [n] "s"(n_bytes), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [tiny] "s"(tiny)
        : RT_ROT_CLOBBERS);
    best_out = best;
    item_out = bitem;
}

'''

SHADOW_FN = '''// Shadow-ray traversal (any hit, render.rs:202-208) from byte offset `start` until the stream ends or some lane's ray hits
// an ITEM: the caller retires those lanes, finds the next node any lane still wants and calls again.  Returns the byte
// offset it stopped at (>= n_bytes: stream finished); fin = 1 in the lanes that hit the ITEM there.  resume in bytes.
__device__ __forceinline__ unsigned skip_shadow_rot(const void *nodes, unsigned n_bytes, unsigned start, float ox, float oy, float oz,
                                                   float lx, float ly, float lz, unsigned &resume_io, unsigned &fin_out)
{
    unsigned resume = resume_io, fin = 0, stop;
    float t0, t1, t2, t3, t4, t5, vx, vy, vz, b, disc, root;
    const float tiny = 0x1p-96f;
    asm volatile(
%s
        : [resume] "+v"(resume), [fin] "+v"(fin), [stop] "=&s"(stop), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [t4] "=&v"(t4), [t5] "=&v"(t5), [vx] "=&v"(vx), [vy] "=&v"(vy), [vz] "=&v"(vz), [b] "=&v"(b), [disc] "=&v"(disc),
          [root] "=&v"(root)
        : [base] "s"(nodes), [n] "s"(n_bytes), [start] "s"(start), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "s"(lx), [ly] "s"(ly),
          [lz] "s"(lz), [tiny] "s"(tiny)
        : RT_ROT_CLOBBERS);
    resume_io = resume;
    fin_out = fin;
    return stop;
}

'''


def primary():
    a = Asm()
    a.op("s_mov_b32 s48, 0")
    a.op("s_mov_b64 s[62:63], exec")
    a.op("s_load_dwordx8 s[40:47], %[base], 0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        primary_copy(a, name)
    a.label(".Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)", "the speculative loads must have landed before their registers are free again")
    return a.render()


def shadow():
    a = Asm()
    a.op("s_mov_b32 s48, %[start]")
    a.op("s_mov_b64 s[62:63], exec")
    a.op("s_load_dwordx8 s[40:47], %[base], s48")
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        shadow_copy(a, name)
    a.label(".Lrt_exit_%=")
    a.op("s_mov_b32 %[stop], %[n]", "stream finished")
    a.label(".Lrt_out_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def main():
    text = HEADER + PRIMARY_FN % primary() + SHADOW_FN % shadow() + "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
