#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_skip_rot.hpp: the traversal loops of k_render_skip in gfx950 assembly (f32 and f64,
plain and fused flavours).

Same arithmetic, operation for operation, as the C++ loops of rt_skip.hpp (and as the first, hand-written assembly loops,
which this file's f32 output was validated against frame for frame before they were retired); what is generated is the
bookkeeping around it.  A lone wave retires about one instruction per 6 cycles whatever its type, and a 1080p frame is as
long as its longest wave, so every scalar instruction of a step counts:

  * node positions are BYTE offsets into the stream (i, resume, skip): no shifts before the scalar loads;
  * both successors of a node (the next one and skip) are fetched at the top of its step into two of THREE register banks,
    and the step ends by branching into the copy of the loop body whose "current node" bank is the one that holds the
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

f32 forms the correctly rounded root as v_sqrt_f32 + two exact FMA residuals (== sqrt_rn_lean, checked against the IEEE sqrt on
all 2^32 inputs).  f64 replays, instruction for instruction, the expansion hipcc emits for the IEEE-correct __builtin_sqrt
(scale by 2^256 below 2^-767, v_rsq_f64, two Goldschmidt/Newton steps in FMA, scale back, pass +-0 and +inf through).

Run:  python3 tools/gen_skip_asm.py   (writes the header; the build does not need this script)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_skip_rot.hpp")

COPIES = {"A": (0, 1, 2), "B": (1, 0, 2), "C": (2, 0, 1)}          # current, next, skip
NEXT_COPY = {"A": "B", "B": "A", "C": "A"}
SKIP_COPY = {"A": "C", "B": "C", "C": "B"}


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


def sp(first, n=2):
    return "s[%d:%d]" % (first, first + n - 1)


class Prec:
    """Register plan and the precision-dependent instruction sequences."""

    def bank(self, b):
        return self.bank_first[b]

    def load(self, a, b, off, comment=None):
        a.op("%s %s, %%[base], %s" % (self.load_op, sp(self.bank(b), self.bank_dwords), off), comment)


class F32(Prec):
    name, ctype, stride = "f32", "float", 32
    bank_first, bank_dwords, load_op = (40, 64, 72), 8, "s_load_dwordx8"
    I, NX = "s48", "s51"
    ACT, M54, M56, M58, TINY, EX = sp(52), sp(54), sp(56), sp(58), sp(60), sp(62)
    clobber_lo, clobber_hi = 40, 79

    def fld(self, b, k):                             # geometry term k = 0..4
        return "s%d" % (self.bank(b) + k)

    def item(self, b):
        return "s%d" % (self.bank(b) + 5)

    def skip(self, b):
        return "s%d" % (self.bank(b) + 6)

    def own(self, b):
        return "s%d" % (self.bank(b) + 7)

    def cand_cmp(self, a):
        a.op("v_cmp_le_f32_e32 vcc, 0, %[disc]")

    def primary_terms(self, a, c):
        a.op("v_mul_f32_e32 %%[t0], %s, %%[dx]" % self.fld(c, 0), "b = (vx*dx + vy*dy) + vz*dz   primitive.rs:57")
        a.op("v_mul_f32_e32 %%[t1], %s, %%[dy]" % self.fld(c, 1))
        a.op("v_mul_f32_e32 %%[t2], %s, %%[dz]" % self.fld(c, 2))
        a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
        a.op("v_add_f32_e32 %[b], %[t0], %[t2]")
        a.op("v_mul_f32_e32 %[t0], %[b], %[b]", "disc = (b*b - vv) + rr   primitive.rs:58")
        a.op("v_subrev_f32_e32 %%[q], %s, %%[t0]" % self.fld(c, 3))
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % self.fld(c, 4))

    def shadow_terms(self, a, c):
        a.op("v_sub_f32_e32 %%[vx], %s, %%[ox]" % self.fld(c, 0), "v = centre - origin   primitive.rs:56")
        a.op("v_sub_f32_e32 %%[vy], %s, %%[oy]" % self.fld(c, 1))
        a.op("v_sub_f32_e32 %%[vz], %s, %%[oz]" % self.fld(c, 2))
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
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % self.fld(c, 3), "disc = (b*b - vv) + rr   primitive.rs:58")

    def fused_disc(self, a, c):
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % self.own(c), "disc = (b*b - vv) + rr of the group's own sphere")

    CORRECT = [
        "v_add_u32_e32 %[t0], -1, %[root]",
        "v_add_u32_e32 %[t1], 1, %[root]",
        "v_fma_f32 %[t3], -%[t0], %[root], {x}",
        "v_fma_f32 %[t4], -%[t1], %[root], {x}",
        "v_cmp_ge_f32_e64 s[56:57], 0, %[t3]",
        "v_cmp_lt_f32_e64 s[58:59], 0, %[t4]",
        "s_nop {nop}",
        "v_cndmask_b32_e64 %[root], %[root], %[t0], s[56:57]",
        "v_cndmask_b32_e64 %[root], %[root], %[t1], s[58:59]",
    ]

    def root(self, a, need_mask, done_label, tiny_label):
        """Correctly rounded sqrt(disc) into %[root] (== sqrt_rn_lean).  need_mask: the lanes whose root is used."""
        a.op("v_sqrt_f32_e32 %[root], %[disc]")
        a.op("v_cmp_lt_f32_e64 s[60:61], |%[disc]|, %[tiny]")
        a.op("s_and_b64 s[56:57], s[60:61], %s" % need_mask)
        a.op("s_cbranch_scc1 %s" % tiny_label, "some needed lane below 2^-96: scaled path")
        for t in self.CORRECT:
            a.op(t.format(x="%[disc]", nop=0))
        a.label(done_label)

    def tiny(self, a, tiny_label, done_label):
        a.label(tiny_label)
        a.op("v_mul_f32_e32 %[t0], 0x4f800000, %[disc]", "root with the 2^32 / 2^-16 scaling for tiny lanes")
        a.op("v_cndmask_b32_e64 %[t5], %[disc], %[t0], s[60:61]")
        a.op("v_sqrt_f32_e32 %[root], %[t5]")
        a.op("s_nop 0")
        for t in self.CORRECT:
            a.op(t.format(x="%[t5]", nop=1))
        a.op("v_mul_f32_e32 %[t0], 0x37800000, %[root]")
        a.op("v_cndmask_b32_e64 %[root], %[root], %[t0], s[60:61]")
        a.op("s_branch %s" % done_label)

    def primary_distance(self, a):
        """vcc (live lanes with disc >= 0) -> vcc = go: t2 >= 0 and d < hit.distance; d left in t4."""
        a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
        a.op("v_sub_f32_e32 %[t4], %[b], %[root]", "t1")
        a.op("v_cmp_lt_f32_e64 s[56:57], 0, %[t4]", "t1 > 0")
        a.op("v_cmp_le_f32_e64 s[58:59], 0, %[t3]", "t2 >= 0")
        a.op("s_and_b64 vcc, vcc, s[58:59]")
        a.op("v_cndmask_b32_e64 %[t4], %[t3], %[t4], s[56:57]", "d = t1 > 0 ? t1 : t2")
        a.op("v_cmp_lt_f32_e64 s[56:57], %[t4], %[best]", "d < hit.distance")
        a.op("s_and_b64 vcc, vcc, s[56:57]", "go")

    def item_update(self, a, c):
        a.op("s_mov_b64 exec, vcc", "primitive.rs:80-83")
        a.op("v_mov_b32_e32 %[best], %[t4]")
        a.op("v_mov_b32_e32 %%[bitem], %s" % self.item(c))
        a.op("s_mov_b64 exec, %s" % self.EX)

    def shadow_t2_negative(self, a):
        a.op("v_cmp_gt_f32_e64 %s, 0, %%[b]" % self.M54, "b < 0: t2 may still be negative")

    def shadow_t2(self, a):
        a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
        a.op("v_cmp_gt_f32_e64 %s, 0, %%[t3]" % self.M56, "t2 < 0")

    primary_decl = "float t0, t1, t2, t3, t4, t5, b, q, disc, root;\n    const float tiny = 0x1p-96f;"
    primary_out = ('[t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),\n          [t3] "=&v"(t3), [t4] "=&v"(t4), [t5] "=&v"(t5), '
                   '[b] "=&v"(b), [q] "=&v"(q), [disc] "=&v"(disc), [root] "=&v"(root)')
    extra_in = ', [tiny] "s"(tiny)'
    shadow_decl = "float t0, t1, t2, t3, t4, t5, vx, vy, vz, b, q, disc, root;\n    const float tiny = 0x1p-96f;"
    shadow_out = ('[t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),\n          [t4] "=&v"(t4), [t5] "=&v"(t5), [vx] "=&v"(vx), '
                  '[vy] "=&v"(vy), [vz] "=&v"(vz), [b] "=&v"(b), [q] "=&v"(q),\n          [disc] "=&v"(disc), [root] "=&v"(root)')
    inf = "__builtin_huge_valf()"


class F64(Prec):
    name, ctype, stride = "f64", "double", 64
    bank_first, bank_dwords, load_op = (36, 52, 68), 16, "s_load_dwordx16"
    I, NX = "s84", "s85"
    ACT, M54, M56, M58, TINY, EX = sp(86), sp(88), sp(90), sp(92), sp(94), sp(96)
    clobber_lo, clobber_hi = 36, 97

    def fld(self, b, k):
        return sp(self.bank(b) + 2 * k)

    def item(self, b):
        return "s%d" % (self.bank(b) + 10)

    def skip(self, b):
        return "s%d" % (self.bank(b) + 11)

    def own(self, b):
        return sp(self.bank(b) + 12)

    def cand_cmp(self, a):
        a.op("v_cmp_le_f64_e32 vcc, 0, %[disc]")

    def primary_terms(self, a, c):
        a.op("v_mul_f64 %%[t0], %s, %%[dx]" % self.fld(c, 0), "b = (vx*dx + vy*dy) + vz*dz   primitive.rs:57")
        a.op("v_mul_f64 %%[t1], %s, %%[dy]" % self.fld(c, 1))
        a.op("v_mul_f64 %%[t2], %s, %%[dz]" % self.fld(c, 2))
        a.op("v_add_f64 %[t0], %[t0], %[t1]")
        a.op("v_add_f64 %[b], %[t0], %[t2]")
        a.op("v_mul_f64 %[t0], %[b], %[b]", "disc = (b*b - vv) + rr   primitive.rs:58")
        a.op("v_add_f64 %%[q], %%[t0], -%s" % self.fld(c, 3))
        a.op("v_add_f64 %%[disc], %s, %%[q]" % self.fld(c, 4))

    def shadow_terms(self, a, c):
        a.op("v_add_f64 %%[vx], %s, -%%[ox]" % self.fld(c, 0), "v = centre - origin   primitive.rs:56")
        a.op("v_add_f64 %%[vy], %s, -%%[oy]" % self.fld(c, 1))
        a.op("v_add_f64 %%[vz], %s, -%%[oz]" % self.fld(c, 2))
        a.op("v_mul_f64 %[t0], %[lx], %[vx]")
        a.op("v_mul_f64 %[t1], %[ly], %[vy]")
        a.op("v_mul_f64 %[t2], %[lz], %[vz]")
        a.op("v_add_f64 %[t0], %[t0], %[t1]")
        a.op("v_add_f64 %[b], %[t0], %[t2]", "b = dot(v, dir)   primitive.rs:57")
        a.op("v_mul_f64 %[t3], %[vx], %[vx]")
        a.op("v_mul_f64 %[t4], %[vy], %[vy]")
        a.op("v_mul_f64 %[t5], %[vz], %[vz]")
        a.op("v_add_f64 %[t3], %[t3], %[t4]")
        a.op("v_add_f64 %[t3], %[t3], %[t5]", "dot(v, v)")
        a.op("v_mul_f64 %[t0], %[b], %[b]")
        a.op("v_add_f64 %[q], %[t0], -%[t3]")
        a.op("v_add_f64 %%[disc], %s, %%[q]" % self.fld(c, 3), "disc = (b*b - vv) + rr   primitive.rs:58")

    def fused_disc(self, a, c):
        a.op("v_add_f64 %%[disc], %s, %%[q]" % self.own(c), "disc = (b*b - vv) + rr of the group's own sphere")

    def root(self, a, need_mask, done_label, tiny_label):
        """hipcc's IEEE-correct f64 sqrt expansion, instruction for instruction (x = t0, y/h = t1, s = t2, r/d = t4)."""
        a.op("v_cmp_gt_f64_e64 %s, %%[scalec], %%[disc]" % self.TINY, "below 2^-767: scale by 2^256")
        a.op("s_nop 1")
        a.op("v_cndmask_b32_e64 %%[e], 0, %%[c256], %s" % self.TINY)
        a.op("v_ldexp_f64 %[t0], %[disc], %[e]", "x")
        a.op("v_rsq_f64_e32 %[t1], %[t0]", "y")
        a.op("s_nop 0")
        a.op("v_mul_f64 %[t2], %[t0], %[t1]", "s0 = x*y")
        a.op("v_mul_f64 %[t1], %[t1], 0.5", "h0 = y/2")
        a.op("v_fma_f64 %[t4], -%[t1], %[t2], 0.5", "r0")
        a.op("v_fma_f64 %[t2], %[t2], %[t4], %[t2]", "s1")
        a.op("v_fma_f64 %[t1], %[t1], %[t4], %[t1]", "h1")
        a.op("v_fma_f64 %[t4], -%[t2], %[t2], %[t0]", "d0")
        a.op("v_fma_f64 %[t2], %[t4], %[t1], %[t2]", "s2")
        a.op("v_fma_f64 %[t4], -%[t2], %[t2], %[t0]", "d1")
        a.op("v_fma_f64 %[root], %[t4], %[t1], %[t2]")
        a.op("v_cndmask_b32_e64 %%[e], 0, %%[cm128], %s" % self.TINY)
        a.op("v_ldexp_f64 %[root], %[root], %[e]")
        a.op("v_cmp_class_f64_e64 %s, %%[t0], %%[cclass]" % self.M56, "+-0 and +inf are their own roots")
        a.op("s_mov_b64 exec, %s" % self.M56)
        a.op("v_mov_b64 %[root], %[t0]")
        a.op("s_mov_b64 exec, %s" % self.EX)
        a.label(done_label)

    def tiny(self, a, tiny_label, done_label):
        pass                                         # the f64 expansion scales without a branch

    def primary_distance(self, a):
        """vcc (live lanes with disc >= 0) -> vcc = go; d left in t3."""
        a.op("v_add_f64 %[t3], %[b], %[root]", "t2")
        a.op("v_add_f64 %[t4], %[b], -%[root]", "t1")
        a.op("v_cmp_lt_f64_e64 %s, 0, %%[t4]" % self.M56, "t1 > 0")
        a.op("v_cmp_le_f64_e64 %s, 0, %%[t3]" % self.M58, "t2 >= 0")
        a.op("s_and_b64 vcc, vcc, %s" % self.M58)
        a.op("s_mov_b64 exec, %s" % self.M56, "d = t1 > 0 ? t1 : t2")
        a.op("v_mov_b64 %[t3], %[t4]")
        a.op("s_mov_b64 exec, %s" % self.EX)
        a.op("v_cmp_lt_f64_e64 %s, %%[t3], %%[best]" % self.M56, "d < hit.distance")
        a.op("s_and_b64 vcc, vcc, %s" % self.M56, "go")

    def item_update(self, a, c):
        a.op("s_mov_b64 exec, vcc", "primitive.rs:80-83")
        a.op("v_mov_b64 %[best], %[t3]")
        a.op("v_mov_b32_e32 %%[bitem], %s" % self.item(c))
        a.op("s_mov_b64 exec, %s" % self.EX)

    def shadow_t2_negative(self, a):
        a.op("v_cmp_gt_f64_e64 %s, 0, %%[b]" % self.M54, "b < 0: t2 may still be negative")

    def shadow_t2(self, a):
        a.op("v_add_f64 %[t3], %[b], %[root]", "t2")
        a.op("v_cmp_gt_f64_e64 %s, 0, %%[t3]" % self.M56, "t2 < 0")

    consts = ("const double scalec = 0x1p-767;\n    const unsigned c256 = 256u, cm128 = 0xffffff80u, cclass = 0x260u;      "
              "// ldexp exponents; class mask: +inf | +0 | -0")
    primary_decl = "double t0, t1, t2, t3, t4, b, q, disc, root;\n    unsigned e;\n    " + consts
    primary_out = ('[t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2),\n          [t3] "=&v"(t3), [t4] "=&v"(t4), [e] "=&v"(e), '
                   '[b] "=&v"(b), [q] "=&v"(q), [disc] "=&v"(disc), [root] "=&v"(root)')
    extra_in = ', [scalec] "s"(scalec), [c256] "v"(c256), [cm128] "v"(cm128), [cclass] "v"(cclass)'
    shadow_decl = "double t0, t1, t2, t3, t4, t5, vx, vy, vz, b, q, disc, root;\n    unsigned e;\n    " + consts
    shadow_out = ('[t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),\n          [t4] "=&v"(t4), [t5] "=&v"(t5), [e] "=&v"(e), '
                  '[vx] "=&v"(vx), [vy] "=&v"(vy), [vz] "=&v"(vz), [b] "=&v"(b), [q] "=&v"(q),\n          [disc] "=&v"(disc), [root] "=&v"(root)')
    inf = "__builtin_huge_val()"


def emit_next(a, P, name):
    a.op("s_cmp_ge_u32 %s, %%[n]" % P.NX)
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.op("s_mov_b32 %s, %s" % (P.I, P.NX))
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % NEXT_COPY[name])


def emit_transitions(a, P, name, c, lab):
    """skip / next (with end check); `next` uses the position computed at the top of the step."""
    a.label(lab("skip"))
    a.op("s_mov_b32 %s, %s" % (P.I, P.skip(c)), "jump over the subtree")
    a.op("s_cmp_ge_u32 %s, %%[n]" % P.I)
    a.op("s_cbranch_scc1 .Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch .Lrt_%s_top_%%=" % SKIP_COPY[name])
    a.label(lab("next"))
    emit_next(a, P, name)


def live_mask(a, P):
    """Issued before the terms: it depends on nothing they compute, and the chain disc -> compare -> branch stays short."""
    a.op("v_cmp_ge_u32_e64 %s, %s, %%[resume]" % (P.ACT, P.I), "active = i >= resume")


def candidates(a, P, hit_label):
    P.cand_cmp(a)
    a.op("s_and_b64 vcc, vcc, %s" % P.ACT, "live lanes whose line meets the sphere")
    a.op("s_cbranch_vccnz %s" % hit_label)


def bound_top(a, P, c, n, s, fused):
    a.op("s_add_u32 %s, %s, %d" % (P.NX, P.I, 2 * P.stride if fused else P.stride),
         "the walk goes on behind the group's own sphere" if fused else None)
    P.load(a, n, P.NX, "both successors, while this node is processed")
    P.load(a, s, P.skip(c))


def sleep_culled(a, P, c):
    a.op("s_andn2_b64 exec, %s, vcc" % P.ACT, "lanes that may not enter sleep until `skip`")
    a.op("v_mov_b32_e32 %%[resume], %s" % P.skip(c))
    a.op("s_mov_b64 exec, %s" % P.EX)


def primary_copy(a, P, name, fused):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    a.op("s_cmp_eq_u32 %s, 0" % P.skip(c))
    a.op("s_cbranch_scc1 %s" % lab("item"))
    # ---------------- BOUND step (group.rs:73) ----------------
    bound_top(a, P, c, n, s, fused)
    live_mask(a, P)
    P.primary_terms(a, c)
    candidates(a, P, lab("bhit"))
    emit_transitions(a, P, name, c, lab)   # nobody can hit the bound: jump (the lanes that culled it are awake again at `skip`)
    a.label(lab("bhit"))
    P.root(a, "vcc", lab("brooted"), lab("btiny"))
    P.primary_distance(a)
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("skip"), "nobody enters")
    sleep_culled(a, P, c)
    if fused:
        # the group's own sphere, for the lanes that entered: same centre, so v, b and b*b - vv are the values just formed
        a.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        P.fused_disc(a, c)
        P.cand_cmp(a)
        a.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        a.op("s_cbranch_vccz %s" % lab("next"))
        P.root(a, "vcc", lab("frooted"), lab("ftiny"))
        P.primary_distance(a)
        P.item_update(a, c)
    a.op("s_branch %s" % lab("next"))
    P.tiny(a, lab("btiny"), lab("brooted"))
    if fused:
        P.tiny(a, lab("ftiny"), lab("frooted"))
    # ---------------- ITEM step (primitive.rs:77-84) ----------------
    a.label(lab("item"))
    a.op("s_add_u32 %s, %s, %d" % (P.NX, P.I, P.stride))
    P.load(a, n, P.NX)
    live_mask(a, P)
    P.primary_terms(a, c)
    candidates(a, P, lab("ihit"))
    emit_next(a, P, name)                  # nobody can hit: an ITEM changes nothing
    a.label(lab("ihit"))
    P.root(a, "vcc", lab("irooted"), lab("itiny"))
    P.primary_distance(a)
    P.item_update(a, c)
    a.op("s_branch %s" % lab("next"))
    P.tiny(a, lab("itiny"), lab("irooted"))


def shadow_decide(a, P, lab, tag):
    """vcc (live lanes with disc >= 0) -> vcc = lanes whose ray hits the sphere (t2 >= 0; certain when b >= 0)."""
    P.shadow_t2_negative(a)
    a.op("s_and_b64 %s, %s, vcc" % (P.M54, P.M54))
    a.op("s_cbranch_scc0 %s" % lab(tag + "decided"), "nobody needs the root: hit = candidates")
    P.root(a, P.M54, lab(tag + "rooted"), lab(tag + "tiny"))
    P.shadow_t2(a)
    a.op("s_and_b64 %s, %s, %s" % (P.M56, P.M56, P.M54), "root lanes that miss after all")
    a.op("s_andn2_b64 vcc, vcc, %s" % P.M56)
    a.label(lab(tag + "decided"))


def shadow_copy(a, P, name, fused):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    a.label(lab("top"))
    a.op("s_cmp_eq_u32 %s, 0" % P.skip(c))
    a.op("s_cbranch_scc1 %s" % lab("item"))
    # ---------------- BOUND step: hit.distance is INF, so a bound culls iff the ray misses it ----------------
    bound_top(a, P, c, n, s, fused)
    live_mask(a, P)
    P.shadow_terms(a, c)
    candidates(a, P, lab("bhit"))
    emit_transitions(a, P, name, c, lab)
    a.label(lab("bhit"))
    shadow_decide(a, P, lab, "b")
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("skip"))
    sleep_culled(a, P, c)
    if fused:
        a.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        P.fused_disc(a, c)
        P.cand_cmp(a)
        a.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        a.op("s_cbranch_vccz %s" % lab("next"))
        shadow_decide(a, P, lab, "f")
        a.op("s_cmp_eq_u64 vcc, 0")
        a.op("s_cbranch_scc1 %s" % lab("next"))
        a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc", "any hit ends those rays; hand them to the caller")
        a.op("s_add_u32 %%[stop], %s, %d" % (P.I, P.stride), "they hit the ITEM behind this BOUND")
        a.op("s_branch .Lrt_out_%=")
    else:
        a.op("s_branch %s" % lab("next"))
    P.tiny(a, lab("btiny"), lab("brooted"))
    if fused:
        P.tiny(a, lab("ftiny"), lab("frooted"))
    # ---------------- ITEM step ----------------
    a.label(lab("item"))
    a.op("s_add_u32 %s, %s, %d" % (P.NX, P.I, P.stride))
    P.load(a, n, P.NX)
    live_mask(a, P)
    P.shadow_terms(a, c)
    candidates(a, P, lab("ihit"))
    emit_next(a, P, name)
    a.label(lab("ihit"))
    shadow_decide(a, P, lab, "i")
    a.op("s_cmp_eq_u64 vcc, 0")
    a.op("s_cbranch_scc1 %s" % lab("next"))
    a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc", "any hit ends those rays; hand them to the caller")
    a.op("s_mov_b32 %%[stop], %s" % P.I)
    a.op("s_branch .Lrt_out_%=")
    P.tiny(a, lab("itiny"), lab("irooted"))


HEADER = """// rt_skip_rot.hpp -- GENERATED by tools/gen_skip_asm.py; edit the generator, not this file.
//
// The traversal loops of k_render_skip in gfx950 assembly (f32 and f64, plain and fused).  Each is the arithmetic of the C++
// loop beside it in rt_skip.hpp (the reference implementation: every launch that counts tests), operation for operation:
//      b    = (vx*dx + vy*dy) + vz*dz              primitive.rs:57   (node terms as SGPR operands)
//      disc = (b*b - vv) + rr                      primitive.rs:58
//      root = correctly rounded sqrt(disc)         f32: v_sqrt_f32 + two exact FMA residuals (== sqrt_rn_lean, which is checked
//                                                  against the IEEE sqrt on all 2^32 inputs); f64: hipcc's own IEEE-correct
//                                                  expansion of __builtin_sqrt, instruction for instruction
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
// next two instructions, a transcendental result (v_sqrt_f32, v_rsq_f64) is not consumed by the next instruction,
// s_waitcnt lgkmcnt(0) before loaded registers are read and at every exit (the speculative loads must have landed before
// their registers are free again).  64-bit selects narrow EXEC and use v_mov_b64.
//
// Node<T> (rt_skip.hpp): five geometry terms, item, skip as a byte offset (0: ITEM), rr of the group's own sphere (fused
// scenes, BOUND nodes).  Fixed SGPRs, f32: s[40:47] / s[64:71] / s[72:79] node banks, s48 position, s51 next, s[52:61]
// masks, s[62:63] EXEC at entry; f64: s[36:51] / s[52:67] / s[68:83], s84, s85, s[86:95], s[96:97].
#pragma once
#include "rt_kernels.hpp"

namespace rt {

"""

PRIMARY_FN = """// Primary-ray traversal: s.group.intersect(&mut h, r) for all 64 rays of the wave.  nodes: Node<%(ctype)s>[n + 2];
// n_bytes = n * %(stride)d.  resume: 0 for lanes with a ray, 0xFFFFFFFF for lanes without.  Returns hit.distance / item per lane.
__device__ __forceinline__ void %(name)s(const void *nodes, unsigned n_bytes, %(ctype)s dx, %(ctype)s dy, %(ctype)s dz, unsigned resume,
                                                 %(ctype)s &best_out, unsigned &item_out)
{
    %(ctype)s best = %(inf)s;
    unsigned bitem = 0;
    %(decl)s
    asm volatile(
%(body)s
        : [best] "+v"(best), [bitem] "+v"(bitem), [resume] "+v"(resume), %(out)s
        : [base] "s"(nodes), [n] "s"(n_bytes), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz)%(extra_in)s
        : %(clobbers)s);
    best_out = best;
    item_out = bitem;
}

"""

SHADOW_FN = """// Shadow-ray traversal (any hit, render.rs:202-208) from byte offset `start` until the stream ends or some lane's ray hits
// an ITEM: the caller retires those lanes, finds the next node any lane still wants and calls again.  Returns the byte
// offset it stopped at (>= n_bytes: stream finished); fin = 1 in the lanes that hit the ITEM there.  resume in bytes.
// hit.distance is INF throughout, so a node is "hit" iff disc >= 0 and t2 = b + root >= 0.
__device__ __forceinline__ unsigned %(name)s(const void *nodes, unsigned n_bytes, unsigned start, %(ctype)s ox, %(ctype)s oy, %(ctype)s oz,
                                                   %(ctype)s lx, %(ctype)s ly, %(ctype)s lz, unsigned &resume_io, unsigned &fin_out)
{
    unsigned resume = resume_io, fin = 0, stop;
    %(decl)s
    asm volatile(
%(body)s
        : [resume] "+v"(resume), [fin] "+v"(fin), [stop] "=&s"(stop), %(out)s
        : [base] "s"(nodes), [n] "s"(n_bytes), [start] "s"(start), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "s"(lx), [ly] "s"(ly),
          [lz] "s"(lz)%(extra_in)s
        : %(clobbers)s);
    resume_io = resume;
    fin_out = fin;
    return stop;
}

"""


def clobbers(P):
    regs = ['"s%d"' % r for r in range(P.clobber_lo, P.clobber_hi + 1)]
    lines, cur = [], '"memory", "vcc", "scc"'
    for r in regs:
        if len(cur) + len(r) + 2 > 118:
            lines.append(cur + ",")
            cur = "          " + r
        else:
            cur += ", " + r
    lines.append(cur)
    return "\n".join(lines)


def primary(P, fused):
    a = Asm()
    a.op("s_mov_b32 %s, 0" % P.I)
    a.op("s_mov_b64 %s, exec" % P.EX)
    P.load(a, 0, "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        primary_copy(a, P, name, fused)
    a.label(".Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def shadow(P, fused):
    a = Asm()
    a.op("s_mov_b32 %s, %%[start]" % P.I)
    a.op("s_mov_b64 %s, exec" % P.EX)
    P.load(a, 0, P.I)
    a.op("s_waitcnt lgkmcnt(0)")
    for name in "ABC":
        shadow_copy(a, P, name, fused)
    a.label(".Lrt_exit_%=")
    a.op("s_mov_b32 %[stop], %[n]", "stream finished")
    a.label(".Lrt_out_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def main():
    text = HEADER
    for P in (F32(), F64()):
        for fused in (False, True):
            sfx = "_fused" if fused else ""
            common = {"ctype": P.ctype, "stride": P.stride, "inf": P.inf, "extra_in": P.extra_in, "clobbers": clobbers(P)}
            text += PRIMARY_FN % dict(common, name="skip_primary_rot" + sfx, body=primary(P, fused), decl=P.primary_decl, out=P.primary_out)
            text += SHADOW_FN % dict(common, name="skip_shadow_rot" + sfx, body=shadow(P, fused), decl=P.shadow_decl, out=P.shadow_out)
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
