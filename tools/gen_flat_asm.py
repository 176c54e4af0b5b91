#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_flat_rot.hpp: the inner loops of the scalar-fed flat scan (rt_flat_sc.hpp) in gfx950
assembly, f32.

A linear scan is wave-uniform: all 64 rays of a wave test the same item at the same moment, so the items are scalars.  A group
of THREE items is one 64-byte record = one s_load_dwordx16 into one of two SGPR banks; the next group's load is issued before
the current group's arithmetic, whose operands are the bank's SGPRs in plain 4-byte VOP2 instructions (1.46 cycles per wave at 8
waves per SIMD, profiles/r02_valu_issue_probe.json; a packed v_pk_* is 2.55 for twice the lanes, LDS reads not counted).  The
three discriminants are reduced with ONE v_max3_f32 and one branch rejects the group; the exact path (root, t2, t1, d, strict
`<` against hit.distance, in item order) runs only when some lane's line meets one of the three spheres.

Per six items the loop issues 2 scalar loads + 48 (primary) / 96 (shadow) VOP2 + 2 v_max3 + 2 compares + 2 branches + 2 waits
+ 3 loop instructions: 10.2 / 18.2 instructions per item and wave of rays.  hipcc's rendering of the same scan (groups of four,
C++) spent 12.5 per item on the primary pass, 4.8 of them scalar.

Arithmetic, operation for operation (primitive.rs:55-72; each + - * rounded once, no FMA outside the exact root):
    primary   b = (vx*dx + vy*dy) + vz*dz ; disc = (b*b - vv) + rr            (v = c - eye, vv, rr pre-formed per item)
    shadow    v = c - o ; b = (v.x*l.x + v.y*l.y) + v.z*l.z ; vv = (v.x*v.x + v.y*v.y) + v.z*v.z ; disc = (b*b - vv) + rr
    exact     disc >= 0 ; root = correctly rounded sqrt(disc) (== sqrt_rn_lean) ; t2 = b + root >= 0 ; t1 = b - root ;
              d = t1 > 0 ? t1 : t2 ; primary: d < hit.distance -> hit.distance = d, item = index ; shadow: any hit retires the ray

Run:  python3 tools/gen_flat_asm.py   (writes the header; the build does not need this script)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_flat_rot.hpp")

BANK = {"A": 36, "B": 52}            # s[36:51], s[52:67]
OFF, IDX, EXS, TINY = "s68", "s69", "s[70:71]", "s[72:73]"
STRIDE = 64


class Asm:
    def __init__(self):
        self.lines = []

    def op(self, text, comment=None):
        self.lines.append(("\t", text, comment))

    def label(self, name):
        self.lines.append(("", name + ":", None))

    def render(self, indent="        "):
        out = []
        for tab, text, comment in self.lines:
            s = '%s"%s%s\\n"' % (indent, "" if tab == "" else "\\t", text)
            if comment:
                s += "  /* %s */" % comment
            out.append(s)
        return "\n".join(out)


def sreg(bank, field, k):
    """SGPR of item k's field (0..4) in a bank: fields are stored [field][item]."""
    return "s%d" % (BANK[bank] + 3 * field + k)


def load(a, bank, off, comment=None):
    a.op("s_load_dwordx16 s[%d:%d], %%[base], %s" % (BANK[bank], BANK[bank] + 15, off), comment)


def refine(a, x):
    a.op("v_mul_f32_e32 %%[root], %s, %%[t0]" % x, "g = x*y")
    a.op("v_mul_f32_e32 %[t0], 0.5, %[t0]", "h = y/2")
    a.op("v_fma_f32 %%[t1], -%%[root], %%[root], %s" % x, "r = x - g*g")
    a.op("v_fma_f32 %[root], %[t1], %[t0], %[root]", "g + r*h")


def exact_root(a, disc, tag):
    """Correctly rounded sqrt(disc) into %[root] for the lanes in EXEC (== sqrt_rn_lean)."""
    a.op("v_rsq_f32_e32 %%[t0], %s" % disc)
    a.op("v_cmp_lt_f32_e64 %s, |%s|, %%[tiny]" % (TINY, disc))
    a.op("s_cmp_lg_u64 %s, 0" % TINY)
    a.op("s_cbranch_scc1 .Lfl_tiny_%s_%%=" % tag, "a lane below 2^-96 (zero included): scaled path")
    refine(a, disc)
    a.label(".Lfl_rooted_%s_%%=" % tag)


def exact_tiny(a, disc, tag):
    a.label(".Lfl_tiny_%s_%%=" % tag)
    a.op("v_mul_f32_e32 %%[t0], 0x4f800000, %s" % disc, "root with the 2^32 / 2^-16 scaling for tiny lanes")
    a.op("v_cndmask_b32_e64 %%[t2], %s, %%[t0], %s" % (disc, TINY))
    a.op("v_rsq_f32_e32 %[t0], %[t2]")
    a.op("v_cmp_eq_f32_e32 vcc, 0, %[t2]", "sqrt(+-0) = +-0 (rsq would make it 0 * inf)")
    refine(a, "%[t2]")
    a.op("v_cndmask_b32_e32 %[root], %[root], %[t2], vcc")
    a.op("v_mul_f32_e32 %[t0], 0x37800000, %[root]")
    a.op("v_cndmask_b32_e64 %%[root], %%[root], %%[t0], %s" % TINY)
    a.op("s_branch .Lfl_rooted_%s_%%=" % tag)


def primary_group(a, bank):
    for k in range(3):
        b, d = "%%[b%d]" % k, "%%[d%d]" % k
        a.op("v_mul_f32_e32 %%[t0], %s, %%[dx]" % sreg(bank, 0, k), "item %d: b = (vx*dx + vy*dy) + vz*dz" % k if k == 0 else None)
        a.op("v_mul_f32_e32 %%[t1], %s, %%[dy]" % sreg(bank, 1, k))
        a.op("v_mul_f32_e32 %%[t2], %s, %%[dz]" % sreg(bank, 2, k))
        a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
        a.op("v_add_f32_e32 %s, %%[t0], %%[t2]" % b)
        a.op("v_mul_f32_e32 %%[t0], %s, %s" % (b, b), "disc = (b*b - vv) + rr" if k == 0 else None)
        a.op("v_subrev_f32_e32 %%[t0], %s, %%[t0]" % sreg(bank, 3, k))
        a.op("v_add_f32_e32 %s, %s, %%[t0]" % (d, sreg(bank, 4, k)))
    a.op("v_max3_f32 %[t0], %[d0], %[d1], %[d2]")
    a.op("v_cmp_le_f32_e32 vcc, 0, %[t0]")
    a.op("s_cbranch_vccnz .Lfl_slow_%s_%%=" % bank, "some lane's line meets one of the three spheres")
    a.label(".Lfl_cont_%s_%%=" % bank)


def primary_slow(a, bank):
    a.label(".Lfl_slow_%s_%%=" % bank)
    a.op("s_lshr_b32 %s, %s, 6" % (IDX, OFF), "index of the group's first item: 3 * (offset / 64)%s" % (" + 3" if bank == "B" else ""))
    a.op("s_mul_i32 %s, %s, 3" % (IDX, IDX))
    if bank == "B":
        a.op("s_add_u32 %s, %s, 3" % (IDX, IDX))
    for k in range(3):
        tag = "%s%d" % (bank, k)
        b, d = "%%[b%d]" % k, "%%[d%d]" % k
        a.op("v_cmp_le_f32_e32 vcc, 0, %s" % d, "item %d, in item order (primitive.rs:79: the first one keeps a tie)" % k)
        a.op("s_and_saveexec_b64 %s, vcc" % EXS)
        a.op("s_cbranch_execz .Lfl_next_%s_%%=" % tag)
        exact_root(a, d, tag)
        a.op("v_add_f32_e32 %%[t0], %s, %%[root]" % b, "t2")
        a.op("v_sub_f32_e32 %%[t1], %s, %%[root]" % b, "t1")
        a.op("v_cmp_lt_f32_e32 vcc, 0, %[t1]")
        a.op("v_cndmask_b32_e32 %[t1], %[t0], %[t1], vcc", "d = t1 > 0 ? t1 : t2")
        a.op("v_cmpx_le_f32_e32 0, %[t0]", "t2 >= 0")
        a.op("v_cmpx_lt_f32_e32 %[t1], %[best]", "d < hit.distance")
        a.op("v_mov_b32_e32 %[best], %[t1]", "primitive.rs:80-83")
        a.op("v_mov_b32_e32 %%[bitem], %s" % IDX)
        a.label(".Lfl_next_%s_%%=" % tag)
        a.op("s_mov_b64 exec, %s" % EXS)
        if k < 2:
            a.op("s_add_u32 %s, %s, 1" % (IDX, IDX))
    a.op("s_branch .Lfl_cont_%s_%%=" % bank)
    for k in range(3):
        exact_tiny(a, "%%[d%d]" % k, "%s%d" % (bank, k))


def shadow_group(a, bank):
    for k in range(3):
        b, d = "%%[b%d]" % k, "%%[d%d]" % k
        a.op("v_sub_f32_e32 %%[vx], %s, %%[ox]" % sreg(bank, 0, k), "item %d: v = centre - origin" % k if k == 0 else None)
        a.op("v_sub_f32_e32 %%[vy], %s, %%[oy]" % sreg(bank, 1, k))
        a.op("v_sub_f32_e32 %%[vz], %s, %%[oz]" % sreg(bank, 2, k))
        a.op("v_mul_f32_e32 %[t0], %[lx], %[vx]")
        a.op("v_mul_f32_e32 %[t1], %[ly], %[vy]")
        a.op("v_mul_f32_e32 %[t2], %[lz], %[vz]")
        a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
        a.op("v_add_f32_e32 %s, %%[t0], %%[t2]" % b, "b = dot(v, dir)" if k == 0 else None)
        a.op("v_mul_f32_e32 %[t0], %[vx], %[vx]")
        a.op("v_mul_f32_e32 %[t1], %[vy], %[vy]")
        a.op("v_mul_f32_e32 %[t2], %[vz], %[vz]")
        a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
        a.op("v_add_f32_e32 %[t0], %[t0], %[t2]", "dot(v, v)" if k == 0 else None)
        a.op("v_mul_f32_e32 %%[t1], %s, %s" % (b, b))
        a.op("v_sub_f32_e32 %[t0], %[t1], %[t0]")
        a.op("v_add_f32_e32 %s, %s, %%[t0]" % (d, sreg(bank, 3, k)), "disc = (b*b - vv) + rr" if k == 0 else None)
    a.op("v_max3_f32 %[t0], %[d0], %[d1], %[d2]")
    a.op("v_cmp_le_f32_e32 vcc, 0, %[t0]")
    a.op("s_cbranch_vccnz .Lfl_slow_%s_%%=" % bank)
    a.label(".Lfl_cont_%s_%%=" % bank)


def shadow_slow(a, bank):
    """EXEC = the rays still pending.  A ray that hits any of the three items retires: flagged and taken out of EXEC."""
    a.label(".Lfl_slow_%s_%%=" % bank)
    for k in range(3):
        tag = "%s%d" % (bank, k)
        b, d = "%%[b%d]" % k, "%%[d%d]" % k
        a.op("v_cmp_le_f32_e32 vcc, 0, %s" % d, "item %d: disc >= 0 among the pending rays" % k)
        a.op("s_and_saveexec_b64 %s, vcc" % EXS)
        a.op("s_cbranch_execz .Lfl_next_%s_%%=" % tag)
        exact_root(a, d, tag)
        a.op("v_add_f32_e32 %%[t0], %s, %%[root]" % b, "t2")
        a.op("v_cmp_le_f32_e32 vcc, 0, %[t0]", "t2 >= 0: the ray is occluded (render.rs:208 only asks has_missed())")
        a.op("v_cndmask_b32_e64 %[occ], %[occ], 1, vcc")
        a.op("s_andn2_b64 %s, %s, vcc" % (EXS, EXS), "retired")
        a.label(".Lfl_next_%s_%%=" % tag)
        a.op("s_mov_b64 exec, %s" % EXS)
        a.op("s_cbranch_execz .Lfl_exit_%=", "every ray of the wave is settled")
    a.op("s_branch .Lfl_cont_%s_%%=" % bank)
    for k in range(3):
        exact_tiny(a, "%%[d%d]" % k, "%s%d" % (bank, k))


def loop(a, group, slow):
    """Two groups per iteration, double-buffered: bank A holds the current group on entry."""
    a.label(".Lfl_loop_%=")
    a.op("s_add_u32 %s, %s, %d" % (IDX, OFF, STRIDE))
    load(a, "B", IDX, "the next group, while this one is tested")
    group(a, "A")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_add_u32 %s, %s, %d" % (IDX, OFF, 2 * STRIDE))
    load(a, "A", IDX)
    group(a, "B")
    a.op("s_add_u32 %s, %s, %d" % (OFF, OFF, 2 * STRIDE))
    a.op("s_cmp_lt_u32 %s, %%[end]" % OFF)
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_cbranch_scc1 .Lfl_loop_%=")
    a.op("s_branch .Lfl_exit_%=")
    slow(a, "A")
    slow(a, "B")


HEADER = """// rt_flat_rot.hpp -- GENERATED by tools/gen_flat_asm.py; edit the generator, not this file.
//
// The inner loops of the scalar-fed flat scan (rt_flat_sc.hpp) in gfx950 assembly, f32.  A group of three items is one 64-byte
// record = one s_load_dwordx16 into one of two SGPR banks (s[36:51], s[52:67]); the next group's load is issued before the
// current group's arithmetic, whose operands are the bank's SGPRs in plain VOP2 instructions.  One v_max3_f32 and one branch
// reject a group; the exact path (root == sqrt_rn_lean, t2, t1, d, strict `<`, item order) runs only when some lane's line meets
// one of the three spheres.  s68 byte offset of the current group pair, s69 scratch / item index, s[70:71] saved EXEC,
// s[72:73] tiny mask: the kernels stay at 80 SGPRs (8 waves per SIMD).
//
// Group record (rt_flat_sc.hpp, FGroup): primary {vx[3], vy[3], vz[3], vv[3], rr[3], pad}; shadow {cx[3], cy[3], cz[3], rr[3], pad[4]}.
// The arrays end in pad groups (rr = -inf: never a candidate) so that the load issued one pair ahead stays inside them.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

"""

PRIMARY = """// Nearest hit of all groups [0, n_bytes / 64) for the wave's 64 primary rays (n_bytes: a multiple of 128).  Lanes without a ray
// scan along (their result is ignored).  Returns hit.distance and the index of the winning item per lane.
__device__ __forceinline__ void flat_primary_scan(const void *groups, unsigned n_bytes, float dx, float dy, float dz, float &best_out,
                                                  unsigned &item_out)
{
    float best = __builtin_huge_valf();
    unsigned bitem = 0;
    float t0, t1, t2, root, b0, b1, b2, d0, d1, d2;
    const float tiny = 0x1p-96f;
    asm volatile(
%(body)s
        : [best] "+v"(best), [bitem] "+v"(bitem), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [root] "=&v"(root),
          [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2)
        : [base] "s"(groups), [end] "s"(n_bytes), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz), [tiny] "s"(tiny)
        : %(clobbers)s);
    best_out = best;
    item_out = bitem;
}

"""

SHADOW = """// Any hit over the groups [begin_bytes / 64, end_bytes / 64) (multiples of 128) for the lanes with pending != 0; returns 1 in the
// lanes whose ray is occluded.  The wave leaves as soon as every pending ray is settled.
__device__ __forceinline__ unsigned flat_shadow_scan(const void *groups, unsigned begin_bytes, unsigned end_bytes, float ox, float oy, float oz,
                                                     float lx, float ly, float lz, unsigned pending)
{
    unsigned occ = 0;
    float t0, t1, t2, root, vx, vy, vz, b0, b1, b2, d0, d1, d2;
    const float tiny = 0x1p-96f;
    asm volatile(
%(body)s
        : [occ] "+v"(occ), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [root] "=&v"(root), [vx] "=&v"(vx), [vy] "=&v"(vy), [vz] "=&v"(vz),
          [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2)
        : [base] "s"(groups), [begin] "s"(begin_bytes), [end] "s"(end_bytes), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "s"(lx),
          [ly] "s"(ly), [lz] "s"(lz), [pend] "v"(pending), [tiny] "s"(tiny)
        : %(clobbers)s);
    return occ;
}

"""


def clobbers():
    regs = ['"s%d"' % r for r in range(36, 74)]
    lines, cur = [], '"memory", "vcc", "scc"'
    for r in regs:
        if len(cur) + len(r) + 2 > 118:
            lines.append(cur + ",")
            cur = "          " + r
        else:
            cur += ", " + r
    lines.append(cur)
    return "\n".join(lines)


def primary():
    a = Asm()
    a.op("s_mov_b32 %s, 0" % OFF)
    load(a, "A", "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    loop(a, primary_group, primary_slow)
    a.label(".Lfl_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)", "the load issued one pair ahead must have landed before its registers are free again")
    return a.render()


def shadow_body():
    a = Asm()
    a.op("s_mov_b64 %[saved], exec")
    a.op("v_cmp_ne_u32_e32 vcc, 0, %[pend]")
    a.op("s_and_b64 exec, exec, vcc", "EXEC = the rays still pending, for the whole scan")
    a.op("s_cbranch_execz .Lfl_done_%=")
    a.op("s_mov_b32 %s, %%[begin]" % OFF)
    load(a, "A", OFF)
    a.op("s_waitcnt lgkmcnt(0)")
    loop(a, shadow_group, shadow_slow)
    a.label(".Lfl_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    a.label(".Lfl_done_%=")
    a.op("s_mov_b64 exec, %[saved]")
    return a.render()


def main():
    text = HEADER
    text += PRIMARY % {"body": primary(), "clobbers": clobbers()}
    sh = SHADOW.replace("unsigned occ = 0;", "unsigned occ = 0;\n    unsigned long long saved;")
    sh = sh.replace(': [occ] "+v"(occ),', ': [occ] "+v"(occ), [saved] "=&s"(saved),')
    text += sh % {"body": shadow_body(), "clobbers": clobbers()}
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
