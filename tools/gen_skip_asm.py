#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_skip_rot.hpp: the traversal loops of k_render_skip in gfx950 assembly (f32 and f64,
plain and fused flavours).

Same arithmetic, operation for operation, as the C++ loops of rt_skip.hpp; what is generated is the bookkeeping around it.
The probe (tools/valu_issue_probe.hip, profiles/r02_valu_issue_probe.json) shows that scalar instructions compete with the
vector ones for a SIMD's issue slots (a step of 10 VALU + 12 SALU costs 35 cycles per SIMD at 8 waves, the 10 VALU alone
about 16), and a frame is as long as its longest wave, so every instruction of a step counts -- and so does every scalar-
cache line: the 16 KB scalar cache serves 64 waves, and a speculative fetch that is not walked evicts a line that is:

  * a position is a BYTE offset into the stream and every node is one stride long (fused scenes walk a COMPACTED stream:
    the sphere a group's bound is built around lives inside the BOUND node), so the only position the loop keeps is NX, the
    offset of the node after the current one; `active = i >= resume` is formed as NX > resume;
  * the likely successor of a node (`skip`: three of four BOUND tests end there, and an ITEM's `skip` is the node behind
    it) is fetched at the top of its step, the first child of a group when somebody enters it, into two of THREE register
    banks, and the step ends in the copy of the loop body whose "current node" bank is the one that holds the successor it
    chose -- no select instructions.  With banks (0, 1, 2) three copies suffice:
        copy A: current 0, next -> 1, skip -> 2      next: B   skip: C
        copy B: current 1, next -> 0, skip -> 2      next: A   skip: C
        copy C: current 2, next -> 0, skip -> 1      next: A   skip: B
    The main paths are laid out A, C, B so that A's and C's `skip` transitions fall through;
  * the commonest step (a node no live lane can hit) does not ask what kind of node it is: an ITEM's `skip` is simply the node
    behind it, so "nobody hits -> go to skip" is right for both.  The kind (flag bits in the item word) is only looked at
    when somebody hits;
  * the stream ends in an END node that every lane hits (disc = +inf) and every lane is awake at (a lane without a ray sleeps
    until END, not for ever), so the walk needs no end-of-stream test: the commonest step is 15 or 16 instructions (10 VALU, one
    scalar load, s_and, branch, s_add, s_waitcnt[, s_branch]); round 1's was 22;
  * FUSED flavour, for scenes in which every group's first child is a sphere concentric with the group's bound (the
    reference's pyramid, group.rs:37-41): a BOUND step goes on to test that sphere for the lanes that enter -- v, b and
    b*b - vv are the same bits, only rr differs.  One step and eight VALU operations fewer per entered group, same tests,
    same order, same values;
  * the f32 loops keep to s[36:73]: with the six registers the hardware adds the kernel stays at 80 SGPRs, the most a wave may
    have at 8 waves per SIMD (86 meant 7).

f32 forms the correctly rounded root as v_rsq_f32 + one exact FMA residual + one FMA correction (== sqrt_rn_lean, checked
against the IEEE sqrt on all 2^32 inputs): 5 instructions; round 1's v_sqrt_f32 + two residuals + two selects took 10.  f64 replays, instruction for instruction, the expansion hipcc emits for the IEEE-correct __builtin_sqrt
(scale by 2^256 below 2^-767, v_rsq_f64, two Goldschmidt/Newton steps in FMA, scale back, pass +-0 and +inf through).

Run:  python3 tools/gen_skip_asm.py   (writes the header; the build does not need this script)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_skip_rot.hpp")

COPIES = {"A": (0, 1, 2), "B": (1, 0, 2), "C": (2, 0, 1)}          # current, next, skip
NEXT_COPY = {"A": "B", "B": "A", "C": "A"}
SKIP_COPY = {"A": "C", "B": "C", "C": "B"}
LAYOUT = "ACB"                                                     # main paths in this order: A.skip and C.skip fall through
FLAG_LIMIT = "0x3fffffff"                                          # item word above this: ITEM (bit 31) or END (bit 30)


class Asm:
    def __init__(self):
        self.lines = []

    def op(self, text, comment=None):
        self.lines.append(("\t", text, comment))

    def label(self, name):
        self.lines.append(("", name + ":", None))

    def extend(self, other):
        self.lines.extend(other.lines)

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


# The shadow bound's in-plane distance on packed math: the lane's (q1, q2) sit in an aligned register pair, the node's (w1, w2) are the first
# two words of its FNodeS record -- both differences, then both squares, are ONE v_pk_*_f32 each (a packed instruction issues in the 4 cycles
# of a plain one, profiles/r02_valu_issue_probe.json).  The pairs are physical registers because inline assembly cannot name the halves of
# a 64-bit operand: QQ is an input ("{v[40:41]}"), TT a clobber.
QQ, TT = (30, 31), (32, 33)
PACKED_CLOBBERS = ['"v%d"' % r for r in TT]
QQ_IN = '[qq] "{v[%d:%d]}"(qq)' % QQ
QQ_DECL = "\n    const rt_v2f qq = { q1, q2 };"


def packed_p2(a, bank, dst):
    """dst = RN(RN((w1 - q1)^2) + RN((w2 - q2)^2)) -- shadow_p2() in rt_skip.hpp, which the counting launches evaluate."""
    assert bank % 2 == 0
    tt = "v[%d:%d]" % TT
    a.op("v_pk_add_f32 %s, s[%d:%d], v[%d:%d] neg_lo:[0,1] neg_hi:[0,1]" % ((tt, bank, bank + 1) + QQ),
         "P2 = |w - q|^2 in the plane perpendicular to the light: (w1 - q1, w2 - q2) ...")
    a.op("v_pk_mul_f32 %s, %s, %s" % (tt, tt, tt), "... their squares ...")
    a.op("v_add_f32_e32 %s, v%d, v%d" % ((dst,) + TT), "... and the sum")


class Prec:
    """Register plan and the precision-dependent instruction sequences."""
    filt = False
    shortcut = False                 # the primary BOUND step's root-free decision (F32F.bound_shortcut)
    sure_enter = False               # the f64 primary BOUND step that needs no exact record at all (F64F.sure_enter_path)
    primary_extra_args = ""
    primary_extra_in = ""
    primary_only = False
    shadow_extra_in = ""
    shadow_extra_out = ""
    shadow_extra_decl = ""
    shadow_vclobbers = []            # vector registers the shadow loops name directly (packed_p2)
    ftmp = "%[t0]"                   # an f32 scratch register of the two-sided shadow bound
    shadow_subst = ()                # named operands that are such registers

    def kind_test(self, a, c, lab):
        a.op("s_cmp_gt_u32 %s, %s" % (self.item(c), FLAG_LIMIT), "an ITEM or the END node?  (flag bits of the item word)")
        a.op("s_cbranch_scc1 %s" % lab("flagged"))

    def own_item_update(self, a, c):
        self.item_update(a, c)

    def bank(self, b):
        return self.bank_first[b]

    def load(self, a, b, off, comment=None):
        a.op("%s %s, %%[base], %s" % (self.load_op, sp(self.bank(b), self.bank_dwords), off), comment)


class F32(Prec):
    name, ctype, stride = "f32", "float", 32
    bank_first, bank_dwords, load_op = (36, 44, 52), 8, "s_load_dwordx8"
    NX = "s60"
    ACT, M54, M56, M58, TINY, EX = sp(62), sp(64), sp(66), sp(68), sp(70), sp(72)
    clobber_lo, clobber_hi = 36, 73

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

    def refine(self, a, x):
        """t0 = y ~ 1/sqrt(x) (v_rsq_f32, 1 ulp)  ->  %[root] = the correctly rounded sqrt(x): g = x*y, h = y/2, r = x - g*g (exact,
        one FMA), root = g + r*h (one FMA).  Equal to the IEEE root for every finite x >= 2^-96 (checked on the device against
        all of them: rt_selftest_sqrt runs the same sequence, sqrt_rn_lean in rt_math.hpp)."""
        a.op("v_mul_f32_e32 %%[root], %s, %%[t0]" % x, "g = x*y")
        a.op("v_mul_f32_e32 %[t0], 0.5, %[t0]", "h = y/2")
        a.op("v_fma_f32 %%[t1], -%%[root], %%[root], %s" % x, "r = x - g*g")
        a.op("v_fma_f32 %[root], %[t1], %[t0], %[root]", "g + r*h")

    def root(self, a, need_mask, done_label, tiny_label):
        """Correctly rounded sqrt(disc) into %[root] (== sqrt_rn_lean).  need_mask: the lanes whose root is used."""
        a.op("v_rsq_f32_e32 %[t0], %[disc]")
        a.op("v_cmp_lt_f32_e64 %s, |%%[disc]|, %%[tiny]" % self.TINY)
        a.op("s_and_b64 %s, %s, %s" % (self.M56, self.TINY, need_mask))
        a.op("s_cbranch_scc1 %s" % tiny_label, "some needed lane below 2^-96 (zero included): scaled path")
        self.refine(a, "%[disc]")
        a.label(done_label)

    def tiny(self, a, tiny_label, done_label):
        a.label(tiny_label)
        a.op("v_mul_f32_e32 %[t0], 0x5f800000, %[disc]", "root with the 2^64 / 2^-32 scaling for tiny lanes (the scaled operand stays >= 2^-85: its residual is never subnormal)")
        a.op("v_cndmask_b32_e64 %%[t5], %%[disc], %%[t0], %s" % self.TINY)
        a.op("v_rsq_f32_e32 %[t0], %[t5]")
        a.op("v_cmp_eq_f32_e64 %s, 0, %%[t5]" % self.M58, "sqrt(+-0) = +-0 (rsq would make it 0 * inf)")
        self.refine(a, "%[t5]")
        a.op("v_cndmask_b32_e64 %%[root], %%[root], %%[t5], %s" % self.M58)
        a.op("v_mul_f32_e32 %[t0], 0x2f800000, %[root]")
        a.op("v_cndmask_b32_e64 %%[root], %%[root], %%[t0], %s" % self.TINY)
        a.op("s_branch %s" % done_label)

    def primary_distance(self, a):
        """vcc (live lanes with disc >= 0) -> vcc = go: t2 >= 0 and d < hit.distance; d left in t4."""
        a.op("v_add_f32_e32 %[t3], %[b], %[root]", "t2")
        a.op("v_sub_f32_e32 %[t4], %[b], %[root]", "t1")
        a.op("v_cmp_lt_f32_e64 %s, 0, %%[t4]" % self.M56, "t1 > 0")
        a.op("v_cmp_le_f32_e64 %s, 0, %%[t3]" % self.M58, "t2 >= 0")
        a.op("s_and_b64 vcc, vcc, %s" % self.M58)
        a.op("v_cndmask_b32_e64 %%[t4], %%[t3], %%[t4], %s" % self.M56, "d = t1 > 0 ? t1 : t2")
        a.op("v_cmp_lt_f32_e64 %s, %%[t4], %%[best]" % self.M56, "d < hit.distance")
        a.op("s_and_b64 vcc, vcc, %s" % self.M56, "go")

    def item_update(self, a, c):
        # by EXEC, not by two v_cndmask_b32_e32 in a row: back to back they stall for 23 cycles each on VCC (an isolated one costs a
        # VOP2's 2.2, an e64 one with its mask in another SGPR pair 4.2; tools/valu_issue_probe.hip)
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
    NX = "s85"
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


class F32F(F32):
    """f32 with the conservative FILTER in front of every test: the walk reads FNode streams (rt_skip.hpp) --
    primary {vx, vy, vz, vv, rr, T, skip_off, tag}, shadow {cx, cy, cz, rr, w1, w2, skip_off, tag}; tag = 0 (BOUND), rr of the
    group's own sphere (fused BOUND), item | bit 31 (ITEM), bits 31 + 30 (END).  A step first asks a bound of the test that is
    proven to hold whenever the reference's test can return a finite distance (primary: fma-chain b' >= T; shadow: squared distance
    of the sphere's centre from the ray in the plane perpendicular to the light <= rr * k1 + K0); only when some live lane passes
    does it form the reference's own discriminant, operation for operation, exactly as the unfiltered loops do."""
    name = "f32"
    filt = True
    shortcut = True

    def item(self, b):
        return "s%d" % (self.bank(b) + 7)            # the tag word: an ITEM's index | bit 31

    def thr(self, b):
        return "s%d" % (self.bank(b) + 5)

    def primary_filter(self, a, c):
        a.op("v_mul_f32_e32 %%[t0], %s, %%[dx]" % self.fld(c, 0), "filter: b' = fma(vz, dz, fma(vy, dy, vx*dx)) >= T ?")
        a.op("v_fma_f32 %%[disc], %s, %%[dy], %%[t0]" % self.fld(c, 1))
        a.op("v_fma_f32 %%[disc], %s, %%[dz], %%[disc]" % self.fld(c, 2))
        a.op("v_cmp_le_f32_e32 vcc, %s, %%[disc]" % self.thr(c))

    def primary_terms_after_filter(self, a, c):
        # t0 = vx*dx is the filter's first product: the same bits the reference forms (primitive.rs:57)
        a.op("v_mul_f32_e32 %%[t1], %s, %%[dy]" % self.fld(c, 1), "b = (vx*dx + vy*dy) + vz*dz   primitive.rs:57")
        a.op("v_mul_f32_e32 %%[t2], %s, %%[dz]" % self.fld(c, 2))
        a.op("v_add_f32_e32 %[t0], %[t0], %[t1]")
        a.op("v_add_f32_e32 %[b], %[t0], %[t2]")
        a.op("v_mul_f32_e32 %[t0], %[b], %[b]", "disc = (b*b - vv) + rr   primitive.rs:58")
        a.op("v_subrev_f32_e32 %%[q], %s, %%[t0]" % self.fld(c, 3))
        a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % self.fld(c, 4))

    # ---- shadow stream (two-sided bound): {w1, w2, cl, R2o | ITEM flag, R2i, R2o_own | END flag, R2i_own, skip_off}
    def s_w1(self, b): return "s%d" % (self.bank(b) + 0)
    def s_w2(self, b): return "s%d" % (self.bank(b) + 1)
    def s_cl(self, b): return "s%d" % (self.bank(b) + 2)
    def s_r2o(self, b): return "s%d" % (self.bank(b) + 3)
    def s_r2i(self, b): return "s%d" % (self.bank(b) + 4)
    def s_r2o_own(self, b): return "s%d" % (self.bank(b) + 5)
    def s_r2i_own(self, b): return "s%d" % (self.bank(b) + 6)
    def s_skip(self, b): return "s%d" % (self.bank(b) + 7)

    def shadow_filter(self, a, c):
        packed_p2(a, self.bank(c), "%[p2]")
        a.op("v_cmp_ngt_f32_e64 vcc, %%[p2], |%s|" % self.s_r2o(c), "not beyond the outer bound (a NaN -- a ray the bounds do not cover -- passes)")

    def kind_test(self, a, c, lab):
        a.op("s_bitcmp1_b32 %s, 31" % self.item(c), "an ITEM or the END node?  (flag bits of the tag word)")
        a.op("s_cbranch_scc1 %s" % lab("flagged"))

    def bound_shortcut(self, a, c, lab):
        """Primary BOUND step, vcc = live lanes with disc >= 0: decide `d < hit.distance` (group.rs:73) WITHOUT the root where the reference's
        own values already settle it ("the BOUND step without its root").  Only for a node the eye is clearly outside of
        (T > -inf: then a finite distance needs b > 0 and is t1 = RN(b - s) with 0 <= s = RN(sqrt(disc)) < b):
            b <= 0                         -> the sphere is behind the eye: t2 < 0, the test returns INF          (no)
            b <  hit.distance              -> RN(b - s) <= b < hit.distance                                        (yes)
            disc * K <= RN(w)^2, w = RN(b - hit.distance) >= 0, K = 1 + 2^-20
                                           -> s <= sqrt(disc) (1 + 2^-24) <= b - hit.distance, so RN(b - s) >= hit.distance   (no)
        A lane none of these settles sends the whole step through the root (lab bexact); otherwise vcc = the lanes that enter (lab bdecided)."""
        a.op("s_cmp_eq_u32 %s, 0xff800000" % self.thr(c), "T = -inf: the eye is not clearly outside this sphere -- the reference's arithmetic decides")
        a.op("s_cbranch_scc1 %s" % lab("bexact"))
        a.op("v_cmp_lt_f32_e64 %s, 0, %%[b]" % self.M54, "b > 0")
        a.op("v_sub_f32_e32 %[t3], %[b], %[best]", "w = b - hit.distance  (-inf while nothing was hit)")
        a.op("v_cmp_gt_f32_e64 %s, 0, %%[t3]" % self.M56, "b < hit.distance: enters")
        a.op("v_mul_f32_e32 %[t4], %[t3], %[t3]", "w^2")
        a.op("v_mul_f32_e32 %[t5], 0x3f800008, %[disc]", "disc (1 + 2^-20)")
        a.op("v_cmp_le_f32_e64 %s, %%[t5], %%[t4]" % self.M58, "the root cannot reach down to b - hit.distance: culled")
        a.op("s_and_b64 vcc, vcc, %s" % self.M54, "candidates in front of the eye")
        a.op("s_or_b64 %s, %s, %s" % (self.M58, self.M58, self.M56), "settled lanes")
        a.op("s_andn2_b64 %s, vcc, %s" % (self.M58, self.M58), "candidates nothing above settles")
        a.op("s_cbranch_scc1 %s" % lab("bexact"), "(vcc lost only lanes with b <= 0, which the root path rejects as well)")
        a.op("s_and_b64 vcc, vcc, %s" % self.M56, "go")
        a.op("s_branch %s" % lab("bdecided"))

    def own_item_update(self, a, c):
        # the fused BOUND's tag is its own sphere's rr: the hit records WHERE it happened (NX = this node's offset + stride, bit 31
        # clear) and the kernel looks the item up in the stream's own_item table afterwards
        a.op("s_mov_b64 exec, vcc", "primitive.rs:80-83")
        a.op("v_mov_b32_e32 %[best], %[t4]")
        a.op("v_mov_b32_e32 %%[bitem], %s" % self.NX)
        a.op("s_mov_b64 exec, %s" % self.EX)

    shadow_extra_in = ', ' + QQ_IN + ', [ol] "v"(ol), [a0] "s"(a0), [k1] "s"(k1), [kc] "v"(kc), [base2] "s"(exact)'
    shadow_extra_out = ', [p2] "=&v"(p2), [av] "=&v"(av), [inn] "=&v"(inn)'
    shadow_extra_decl = "\n    float p2, av, inn;" + QQ_DECL
    shadow_vclobbers = PACKED_CLOBBERS
    # the shadow loops' temporaries t0 / t1 ARE the halves of TT (two registers fewer at the statement: the lane-cooperative flavour of the
    # kernel sits at the 64-register limit of 8 waves per SIMD)
    # ... and b*b - vv / the discriminant of the reference's arithmetic (shadow_exact) take the registers of v.y / v.x, dead by then
    shadow_decl = F32.shadow_decl.replace("float t0, t1, ", "float ").replace("b, q, disc, root", "b, root")
    shadow_out = F32.shadow_out.replace('[t0] "=&v"(t0), [t1] "=&v"(t1), ', "").replace(' [q] "=&v"(q),\n          [disc] "=&v"(disc),', "")
    shadow_subst = (("%[t0]", "v%d" % TT[0]), ("%[t1]", "v%d" % TT[1]), ("%[q]", "%[vy]"), ("%[disc]", "%[vx]"))


class F64F(F64):
    """f64 PRIMARY walk behind the f32 filter (round 4; the shadow walk of f64 scenes stays the plain F64 loop).  The walk reads the scene's
    FNode stream -- f32 roundings of {vx, vy, vz}, a threshold T for the f32 fma-chain b' on an f32 rounding of the ray direction, skip_off
    and tag as in F32F -- one 32-byte record per step; only when some live lane has b' >= T is the node's own 64-byte Node<double> record
    fetched (same index, twice the offset) and the reference's f64 test run, instruction for instruction as in F64.  Positions (NX, resume,
    skip_off) are byte offsets into the FNODE stream.  T: rt_skip.hpp primary_filter_threshold64 -- a finite f64 distance needs
    b >= sqrt(vv - rr) (1 - 1e-15), and b' is within 5.3 eps32 sqrt(vv) of b (two f32 roundings of the operands, three of the chain)."""
    name = "f64"
    filt = True
    primary_only = True
    stride = 32
    fbank_first = (36, 44, 52)                       # three filter banks of 8; the exact record lives in s[60:75]
    EXACT = 60
    # Round 5: s76 scratch, s77 NX, masks s[78:89] -- the filtered f64 loops end at s89 (96 scalar registers with the hardware's six: seven
    # waves per SIMD where the plain F64 loops' s[36:97] allow six; the f64 walk waits for its node records like the f32 one)
    TMP = "s76"
    NX = "s77"
    ACT, M54, M56, M58, TINY, EX = sp(78), sp(80), sp(82), sp(84), sp(86), sp(88)
    clobber_lo, clobber_hi = 36, 89
    fn_suffix = ""
    reserved = (88, 89)                              # k_render_skip_f64: amdgpu_num_sgpr(96) + amdgpu_waves_per_eu(7) leave the compiler s[0:87]
    load_op = "s_load_dwordx8"

    def fbank(self, b):
        return self.fbank_first[b]

    def load(self, a, b, off, comment=None):
        a.op("s_load_dwordx8 %s, %%[base], %s" % (sp(self.fbank(b), 8), off), comment)

    def fld(self, b, k):                             # geometry terms: always the exact record of the CURRENT node
        return sp(self.EXACT + 2 * k)

    def own(self, b):
        return sp(self.EXACT + 12)

    def item(self, b):
        return "s%d" % (self.fbank(b) + 7)           # tag word of the filter record

    def skip(self, b):
        return "s%d" % (self.fbank(b) + 6)

    def thr(self, b):
        return "s%d" % (self.fbank(b) + 5)

    def primary_filter(self, a, c):
        a.op("v_mul_f32_e32 %%[tf0], s%d, %%[dxf]" % (self.fbank(c) + 0), "filter (f32): b' = fma(vz, dz, fma(vy, dy, vx*dx)) >= T ?")
        a.op("v_fma_f32 %%[tf1], s%d, %%[dyf], %%[tf0]" % (self.fbank(c) + 1))
        a.op("v_fma_f32 %%[tf1], s%d, %%[dzf], %%[tf1]" % (self.fbank(c) + 2))
        a.op("v_cmp_le_f32_e32 vcc, %s, %%[tf1]" % self.thr(c))

    def fetch_exact(self, a):
        tmp = self.TMP                               # (free: the filter banks end at s59, the exact record at s75, NX and the masks start at s77)
        a.op("s_sub_u32 %s, %s, %d" % (tmp, self.NX, self.stride), "this node's offset in the filter stream ...")
        a.op("s_lshl_b32 %s, %s, 1" % (tmp, tmp), "... and in the Node<double> stream")
        a.op("s_load_dwordx16 %s, %%[base2], %s" % (sp(self.EXACT, 16), tmp), "its exact record")

    def primary_terms_after_filter(self, a, c):
        self.fetch_exact(a)
        a.op("s_waitcnt lgkmcnt(0)")
        self.primary_terms(a, c)

    def kind_test(self, a, c, lab):
        a.op("s_bitcmp1_b32 %s, 31" % self.item(c), "an ITEM or the END node?  (flag bits of the tag word)")
        a.op("s_cbranch_scc1 %s" % lab("flagged"))

    shortcut = True

    def bound_shortcut(self, a, c, lab):
        """F32F.bound_shortcut in f64 (same three facts; K = 1 + 2^-20 again -- far more than f64 needs, but a literal an f64 instruction can
        carry: its high word): saves the 17-instruction f64 root and the distance on the steps it settles."""
        a.op("s_cmp_eq_u32 %s, 0xff800000" % self.thr(c), "T = -inf: the eye is not clearly outside this sphere -- the reference's arithmetic decides")
        a.op("s_cbranch_scc1 %s" % lab("bexact"))
        a.op("v_cmp_lt_f64_e64 %s, 0, %%[b]" % self.M54, "b > 0")
        a.op("v_add_f64 %[t3], %[b], -%[best]", "w = b - hit.distance  (-inf while nothing was hit)")
        a.op("v_cmp_gt_f64_e64 %s, 0, %%[t3]" % self.M56, "b < hit.distance: enters")
        a.op("v_mul_f64 %[t4], %[t3], %[t3]", "w^2")
        a.op("v_mul_f64 %[t0], %[kk], %[disc]", "disc (1 + 2^-20)")
        a.op("v_cmp_le_f64_e64 %s, %%[t0], %%[t4]" % self.M58, "the root cannot reach down to b - hit.distance: culled")
        a.op("s_and_b64 vcc, vcc, %s" % self.M54, "candidates in front of the eye")
        a.op("s_or_b64 %s, %s, %s" % (self.M58, self.M58, self.M56), "settled lanes")
        a.op("s_andn2_b64 %s, vcc, %s" % (self.M58, self.M58), "candidates nothing above settles")
        a.op("s_cbranch_scc1 %s" % lab("bexact"), "(vcc lost only lanes with b <= 0, which the root path rejects as well)")
        a.op("s_and_b64 vcc, vcc, %s" % self.M56, "go")
        a.op("s_branch %s" % lab("bdecided"))

    sure_enter = True

    def t_in(self, b):
        return "s%d" % (self.fbank(b) + 3)           # FNode::a3 of an f64 scene: b' >= T_in proves a finite distance t1 > 0 (rt_skip.hpp)

    def t_own(self, b):
        return "s%d" % (self.fbank(b) + 4)           # FNode::a4: b' < T_own proves the group's own sphere returns INF (+inf: it has none)

    def sure_enter_path(self, a, c, n, name, lab):
        """Round 5.  An entered BOUND step of the f64 walk fetched the node's Node<double> record BEHIND the filter's verdict (a dependent
        scalar load more than the f32 walk, whose filter record is the exact one) and ran eight f64 operations and the root-free decision
        on it.  Most such steps need neither: when every candidate lane has b' >= T_in (the f64 test returns a finite t1 > 0, so d <= b),
        b' (1 + 2^-11) < hit.distance (b <= b' + 6 eps |v| <= b' (1 + 2^-11) for b' >= T_in >= 2^-9 |v|: d <= b < hit.distance, the lane
        enters) and b' < T_own (the group's own sphere returns INF), the step is decided by the filter record alone.  Any candidate lane
        that is not sure of all three sends the wave down the exact path.  Counting launches hold both verdicts against the f64 test."""
        a.op("s_bitcmp1_b32 %s, 31" % self.item(c), "an ITEM or the END node: the exact path")
        a.op("s_cbranch_scc1 %s" % lab("exact"))
        a.op("v_cmp_le_f32_e64 %s, %s, %%[tf1]" % (self.M54, self.t_in(c)), "b' >= T_in: a finite distance for sure")
        a.op("s_andn2_b64 %s, vcc, %s" % (self.M56, self.M54), "candidates that are not")
        a.op("s_cbranch_scc1 %s" % lab("exact"))
        a.op("v_mul_f32_e32 %[tf0], 0x3f801000, %[tf1]", "b' (1 + 2^-11) >= b >= d")
        a.op("v_cvt_f64_f32_e32 %[t0], %[tf0]")
        a.op("v_cmp_gt_f64_e64 %s, %%[best], %%[t0]" % self.M56, "< hit.distance: enters")
        a.op("s_andn2_b64 %s, vcc, %s" % (self.M56, self.M56), "candidates that may not")
        a.op("s_cbranch_scc1 %s" % lab("exact"))
        a.op("v_cmp_le_f32_e64 %s, %s, %%[tf1]" % (self.M56, self.t_own(c)), "b' >= T_own: the own sphere may be hit")
        a.op("s_and_b64 %s, %s, vcc" % (self.M56, self.M56))
        a.op("s_cbranch_scc1 %s" % lab("exact"))
        enter_group(a, self, c, n)
        sleep_culled(a, self, c)
        a.op("s_branch %s" % lab("next"), "every candidate enters, nobody can hit the own sphere: no exact record needed")
        a.label(lab("exact"))

    def own_item_update(self, a, c):
        a.op("s_mov_b64 exec, vcc", "primitive.rs:80-83")
        a.op("v_mov_b64 %[best], %[t3]")
        a.op("v_mov_b32_e32 %%[bitem], %s" % self.NX, "WHERE it happened: the kernel finds the item in the stream's own_item table")
        a.op("s_mov_b64 exec, %s" % self.EX)

    primary_decl = F64.primary_decl + "\n    float tf0, tf1;\n    const double kk = 0x1.00001p+0;      // 1 + 2^-20 (bound_shortcut)"
    primary_out = F64.primary_out + ', [tf0] "=&v"(tf0), [tf1] "=&v"(tf1)'
    primary_extra_args = ", float dxf, float dyf, float dzf, const void *exact"
    primary_extra_in = ', [dxf] "v"(dxf), [dyf] "v"(dyf), [dzf] "v"(dzf), [base2] "s"(exact), [kk] "s"(kk)'


class F64FS(F64F):
    """The SHADOW walk of f64 scenes behind the f32 walk's TWO-SIDED bound (round 4): the walk reads FNodeS records {w1, w2, cl, R2o | ITEM /
    END flag, R2i, R2o_own | END flag, R2i_own, skip_off} -- the centre in the plane perpendicular to the light and along it, formed in double
    and rounded once, and the f32 walk's bounds with rr rounded up (outer) / down (inner): their margins cover an f32 reference's roundings,
    a superset of what the f64 reference needs -- and decides a step like shadow_copy_filt: beyond the outer bound a miss, inside the inner
    one (and in front) a hit, behind the origin a miss; only a lane between the bounds fetches the node's Node<double> record and runs the
    reference's sixteen f64 operations (and the f64 root where b < 0) for every lane.  Counting launches hold every verdict against the
    f64 test (rt_skip.hpp, shadow_filter_verdict)."""
    primary_only = False

    def item(self, b):
        return "s%d" % (self.EXACT + 10)             # the exact record's item word

    def skip(self, b):
        return "s%d" % (self.fbank(b) + 7)           # FNodeS::skip_off

    def s_skip(self, b):
        return self.skip(b)

    def kind_test(self, a, c, lab):
        F64.kind_test(self, a, c, lab)

    # FNodeS: {w1, w2, cl, R2o | ITEM flag, R2i, R2o_own | END flag, R2i_own, skip_off}
    def s_cl(self, b): return "s%d" % (self.fbank(b) + 2)
    def s_r2o(self, b): return "s%d" % (self.fbank(b) + 3)
    def s_r2i(self, b): return "s%d" % (self.fbank(b) + 4)
    def s_r2o_own(self, b): return "s%d" % (self.fbank(b) + 5)
    def s_r2i_own(self, b): return "s%d" % (self.fbank(b) + 6)
    ftmp = "%[tf0]"

    def shadow_filter(self, a, c):
        packed_p2(a, self.fbank(c), "%[p2]")
        a.op("v_cmp_ngt_f32_e64 vcc, %%[p2], |%s|" % self.s_r2o(c), "not beyond the outer bound (a NaN -- a ray the bounds do not cover -- passes)")

    def shadow_terms(self, a, c):
        tmp = self.TMP
        a.op("s_sub_u32 %s, %s, %d" % (tmp, self.NX, self.stride), "this node's offset in the filter stream ...")
        a.op("s_lshl_b32 %s, %s, 1" % (tmp, tmp), "... and in the Node<double> stream")
        a.op("s_load_dwordx16 %s, %%[base2], %s" % (sp(self.EXACT, 16), tmp), "its exact record")
        a.op("s_waitcnt lgkmcnt(0)")
        F64.shadow_terms(self, a, c)

    # (the bound's scratch register and P2 + a^2 live in the halves of TT, free between two steps' P2: two vector registers fewer, and the f64
    # kernel's 81 become 79 -- six waves per SIMD instead of five)
    shadow_decl = F64.shadow_decl + "\n    float p2, av;" + QQ_DECL
    shadow_out = F64.shadow_out + ', [p2] "=&v"(p2), [av] "=&v"(av)'
    shadow_subst = (("%[tf0]", "v%d" % TT[0]), ("%[inn]", "v%d" % TT[1]))
    shadow_extra_in = ', ' + QQ_IN + ', [ol] "v"(ol), [a0] "s"(a0), [k1] "s"(k1), [kc] "v"(kc), [base2] "s"(exact)'
    shadow_vclobbers = PACKED_CLOBBERS


class F64F_LO(F64F):
    """Round 6: the same loops in s[20:73] -- for k_render_skip_fast64_coop (rt_skip_fast64.hpp), which keeps so little across the loops that
    s[0:19] are enough for it: .sgpr_count 80, EIGHT waves per SIMD where every other f64 kernel runs seven (MI355X_MICROARCH.md,
    "Residency").  Nothing is reserved under amdgpu_num_sgpr(82): every register of the window is declared."""
    fbank_first = (20, 28, 36)
    EXACT = 44
    TMP = "s60"
    NX = "s61"
    ACT, M54, M56, M58, TINY, EX = sp(62), sp(64), sp(66), sp(68), sp(70), sp(72)
    clobber_lo, clobber_hi = 20, 73
    fn_suffix = "_lo"
    reserved = (32,)                                 # (the stack pointer of a kernel that has no stack: reserved in every kernel)
    mark_name = "f64-lo"
    lreg = "s"
    # the kernel has s[0:19] for itself AND for the statements' scalar operands: the constants travel in vector registers here
    extra_in = F64.extra_in.replace('[scalec] "s"(scalec)', '[scalec] "v"(scalec)')


class F64FS_LO(F64FS):
    fbank_first = F64F_LO.fbank_first
    EXACT = F64F_LO.EXACT
    TMP, NX = F64F_LO.TMP, F64F_LO.NX
    ACT, M54, M56, M58, TINY, EX = F64F_LO.ACT, F64F_LO.M54, F64F_LO.M56, F64F_LO.M58, F64F_LO.TINY, F64F_LO.EX
    clobber_lo, clobber_hi = F64F_LO.clobber_lo, F64F_LO.clobber_hi
    fn_suffix = "_lo"
    reserved = (32,)
    mark_name = "f64-lo"
    extra_in = F64F_LO.extra_in
    shadow_extra_in = F64FS.shadow_extra_in.replace('[a0] "s"(a0)', '[a0] "v"(a0)').replace('[k1] "s"(k1)', '[k1] "v"(k1)')


def top_of(name):
    return ".Lrt_%s_top_%%=" % name


def emit_skip(a, P, name, c, lab, skip_reg=None):
    """`skip` transition: the successor is the node behind the subtree (an ITEM's own successor)."""
    a.label(lab("skip"))
    a.op("s_add_u32 %s, %s, %d" % (P.NX, skip_reg or P.skip(c), P.stride), "jump over the subtree")
    a.op("s_waitcnt lgkmcnt(0)")
    nxt = SKIP_COPY[name]
    if LAYOUT.index(nxt) != LAYOUT.index(name) + 1:                # A -> C and C -> B fall through
        a.op("s_branch %s" % top_of(nxt))


def emit_next(a, P, name):
    a.op("s_add_u32 %s, %s, %d" % (P.NX, P.NX, P.stride))
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch %s" % top_of(NEXT_COPY[name]))


def node_skip(P, c, terms):
    return P.s_skip(c) if (P.filt and terms == P.shadow_terms) else P.skip(c)


def step_top(a, P, c, n, s, terms):
    """The part of a step every node shares: fetch the likely successor, form the live mask and the discriminant."""
    # Only `skip` is fetched ahead: three of four BOUND tests end there, and an ITEM's `skip` IS its successor.  The first
    # child of a group is fetched when somebody enters (enter_group): fetched at the top of every BOUND step it doubled the
    # scalar-cache misses (most of those lines are never walked), which cost more than the late fetch on the entered quarter.
    P.load(a, s, node_skip(P, c, terms), "the likely successor, while this node is processed")
    if not lazy_wake(P):
        a.op("v_cmp_gt_u32_e64 %s, %s, %%[resume]" % (P.ACT, P.NX), "active = i >= resume  (NX = i + stride)")
    if P.filt:
        # Round 5: the filter is asked for EVERY lane, asleep or not, and who is awake only where somebody passes (hit_entry): a lane that sleeps
        # missed an enclosing bound and passes the filter of a node inside it in 0.2 % of the quiet steps (1080p, a CPU replay), so the
        # commonest step loses its compare with `resume` and the s_and -- two of its ten instructions.  Lanes without a ray, and shadow
        # lanes that have retired, carry a ray no bound lets through (a NaN direction / an in-plane origin at infinity: rt_skip.hpp).
        (P.primary_filter if terms == P.primary_terms else P.shadow_filter)(a, c)
        for _ in range(int(os.environ.get("RT_GEN_PAD_VALU", "0")) if P.ctype == "float" else 0):   # design-time probe: what one more instruction per step costs
            a.op("v_max_f32_e32 %[t2], %[t2], %[t2]")
        for _ in range(int(os.environ.get("RT_GEN_PAD_SALU", "0")) if P.ctype == "float" else 0):
            a.op("s_mov_b32 %s, %s" % (P.TINY.split(":")[0].replace("s[", "s"), P.NX))
        if not lazy_wake(P):
            a.op("s_and_b64 vcc, vcc, %s" % P.ACT, "live lanes the bound cannot rule out")
        return
    terms(a, c)
    P.cand_cmp(a)
    a.op("s_and_b64 vcc, vcc, %s" % P.ACT, "live lanes whose line meets the sphere")


def enter_group(a, P, c, n):
    P.load(a, n, P.NX, "somebody enters: fetch the group's first child")


def sleep_culled(a, P, c):
    a.op("s_andn2_b64 exec, %s, vcc" % P.ACT, "lanes that may not enter sleep until `skip`")
    a.op("v_mov_b32_e32 %%[resume], %s" % P.skip(c))
    a.op("s_mov_b64 exec, %s" % P.EX)


def kind_test(a, P, c, lab):
    P.kind_test(a, c, lab)


def lazy_wake(P):
    """The f32 filtered loops ask who is awake only where some lane passes the bound (step_top).  The f64 loops keep the compare at the
    top of the step: their steps are bound by f64 arithmetic, and the poisoned ray costs their kernels a spill."""
    return P.filt and P.ctype == "float"


def hit_entry(a, P, shadow=False):
    """Lazy loops: some lane passed the bound.  Now: who is awake?  (The primary path ands ACT into its exact test; the shadow path
    has no other test before it reads vcc.)"""
    if not lazy_wake(P):
        return
    a.op("v_cmp_gt_u32_e64 %s, %s, %%[resume]" % (P.ACT, P.NX), "active = i >= resume  (NX = i + stride)")
    if shadow:
        a.op("s_and_b64 vcc, vcc, %s" % P.ACT, "live lanes inside the outer bound")


def exact_after_filter(a, P, c, lab, terms):
    """Filtered loops: some lane passed the bound -- now the reference's own discriminant, for every lane."""
    if not P.filt:
        return
    (P.primary_terms_after_filter if terms == P.primary_terms else terms)(a, c)
    P.cand_cmp(a)
    a.op("s_and_b64 vcc, vcc, %s" % P.ACT, "live lanes whose line meets the sphere")
    a.op("s_cbranch_vccz %s" % lab("skip"), "the bound let it through, the test does not: nobody can hit the node")


def primary_copy(P, name, fused):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    m, k = Asm(), Asm()
    m.label(lab("top"))
    step_top(m, P, c, n, s, P.primary_terms)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, P, name, c, lab)          # nobody can hit the node: a BOUND is jumped over, an ITEM changes nothing
    # ---------------- somebody's line meets the sphere ----------------
    k.label(lab("hit"))
    hit_entry(k, P)
    if P.sure_enter and P.filt:
        P.sure_enter_path(k, c, n, name, lab)
    exact_after_filter(k, P, c, lab, P.primary_terms)
    kind_test(k, P, c, lab)
    # BOUND (group.rs:73)
    if P.shortcut:
        P.bound_shortcut(k, c, lab)
        k.label(lab("bexact"))
    P.root(k, "vcc", lab("brooted"), lab("btiny"))
    P.primary_distance(k)
    if P.shortcut:
        k.label(lab("bdecided"))
    k.op("s_cbranch_vccz %s" % lab("skip"), "nobody enters (the lanes that culled it are awake again at `skip`)")
    enter_group(k, P, c, n)
    sleep_culled(k, P, c)
    if fused:
        # the group's own sphere, for the lanes that entered: same centre, so v, b and b*b - vv are the values just formed
        k.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        P.fused_disc(k, c)
        P.cand_cmp(k)
        k.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        k.op("s_cbranch_vccz %s" % lab("next"))
        P.root(k, "vcc", lab("frooted"), lab("ftiny"))
        P.primary_distance(k)
        P.own_item_update(k, c)
    k.label(lab("next"))
    emit_next(k, P, name)
    P.tiny(k, lab("btiny"), lab("brooted"))
    if fused:
        P.tiny(k, lab("ftiny"), lab("frooted"))
    # ITEM (primitive.rs:77-84) or END
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 30" % P.item(c))
    k.op("s_cbranch_scc1 .Lrt_exit_%=", "END: every lane is awake here and hits it")
    P.root(k, "vcc", lab("irooted"), lab("itiny"))
    P.primary_distance(k)
    P.item_update(k, c)
    k.op("s_branch %s" % lab("skip"), "an ITEM's `skip` is the node behind it")
    P.tiny(k, lab("itiny"), lab("irooted"))
    return m, k


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


def shadow_copy(P, name, fused):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    m, k = Asm(), Asm()
    m.label(lab("top"))
    step_top(m, P, c, n, s, P.shadow_terms)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, P, name, c, lab)
    k.label(lab("hit"))
    exact_after_filter(k, P, c, lab, P.shadow_terms)
    kind_test(k, P, c, lab)
    # BOUND: hit.distance is INF, so a bound culls iff the ray misses it
    shadow_decide(k, P, lab, "b")
    k.op("s_cbranch_vccz %s" % lab("skip"))
    enter_group(k, P, c, n)
    sleep_culled(k, P, c)
    if fused:
        k.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        P.fused_disc(k, c)
        P.cand_cmp(k)
        k.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        k.op("s_cbranch_vccz %s" % lab("next"))
        shadow_decide(k, P, lab, "f")
        k.op("s_cbranch_vccz %s" % lab("next"))
        k.op("s_branch .Lrt_fin_%=", "any hit ends those rays; hand them to the caller")
    k.label(lab("next"))
    emit_next(k, P, name)
    P.tiny(k, lab("btiny"), lab("brooted"))
    if fused:
        P.tiny(k, lab("ftiny"), lab("frooted"))
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 30" % P.item(c))
    k.op("s_cbranch_scc1 .Lrt_exit_%=", "END: every lane is awake here and hits it")
    shadow_decide(k, P, lab, "i")
    k.op("s_cbranch_vccz %s" % lab("skip"), "an ITEM's `skip` is the node behind it")
    k.op("s_branch .Lrt_fin_%=")
    P.tiny(k, lab("itiny"), lab("irooted"))
    return m, k


def shadow_exact(k, P, src, lab, tag, rr_reg):
    """The reference's own test (primitive.rs:55-72) for every lane, node terms from bank `src` (a Node<float> record of the exact stream):
    vcc = live lanes (P.ACT) whose ray hits the sphere with squared radius rr_reg."""
    a = k
    a.op("v_sub_f32_e32 %%[vx], %s, %%[ox]" % P.fld(src, 0), "v = centre - origin   primitive.rs:56")
    a.op("v_sub_f32_e32 %%[vy], %s, %%[oy]" % P.fld(src, 1))
    a.op("v_sub_f32_e32 %%[vz], %s, %%[oz]" % P.fld(src, 2))
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
    a.op("v_add_f32_e32 %%[disc], %s, %%[q]" % rr_reg, "disc = (b*b - vv) + rr   primitive.rs:58")
    P.cand_cmp(a)
    a.op("s_and_b64 vcc, vcc, %s" % P.ACT)
    shadow_decide(a, P, lab, tag)


def shadow_two_sided(a, P, r2i, exact_label, tag):
    a.op("v_cmp_le_f32_e64 %s, %%[inn], %s" % (P.M56, r2i), "origin inside the sphere, by a margin")
    a.op("s_or_b64 %s, %s, %s" % (P.M56, P.M56, P.M58), "... or b >= 0, by a margin")
    a.op("v_fma_f32 %s, %%[inn], %%[k1], %%[p2]" % P.ftmp, "P2 + k1 |centre - origin|^2: the reference's rounding of disc grows with the distance")
    a.op("v_cmp_le_f32_e64 %s, %s, %s" % (P.M54, P.ftmp, r2i), "inside the inner bound: disc >= 0, by a margin")
    a.op("s_and_b64 %s, %s, %s" % (P.M54, P.M54, P.M56), "sure hits")
    a.op("s_andn2_b64 %s, vcc, %s" % (P.M56, P.M54), "lanes between the bounds: the reference's test decides")
    a.op("s_cbranch_scc1 %s" % exact_label)


def shadow_second_chance(k, P, r2o, exact2, none_label, some_label):
    """Lanes between the bounds before the reference's arithmetic is fetched: most of them have the sphere BEHIND them -- b < 0 by a
    margin while the origin is clearly outside the sphere (P2 + a^2 >= R2o (1 + 4 tau)): disc < 0, or the root is smaller than |b| and
    t2 = b + root < 0 -- the reference's test says miss.  vcc = lanes inside the outer bound, M56 = those not settled yet."""
    k.op("v_mul_f32_e32 %s, %%[kc], %%[inn]" % P.ftmp, "(P2 + a^2) / (1 + 4 tau)")
    k.op("v_cmp_le_f32_e64 %s, |%s|, %s" % (P.TINY, r2o, P.ftmp), "the origin is clearly outside the sphere")
    k.op("v_cmp_le_f32_e64 %s, %%[av], -%%[a0]" % P.M54, "b < 0, by a margin")
    k.op("s_and_b64 %s, %s, %s" % (P.TINY, P.TINY, P.M54), "sure misses")
    k.op("s_andn2_b64 %s, %s, %s" % (P.M56, P.M56, P.TINY), "still between the bounds")
    k.op("s_cbranch_scc1 %s" % exact2)
    k.op("s_andn2_b64 vcc, vcc, %s" % P.TINY, "every lane is settled: the hits are the lanes inside the outer bound that are not sure misses")
    k.op("s_cbranch_vccz %s" % none_label)
    k.op("s_branch %s" % some_label)


def shadow_copy_filt(P, name, fused):
    """Two-sided flavour of shadow_copy (f32 filtered streams)."""
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    m, k = Asm(), Asm()
    m.label(lab("top"))
    step_top(m, P, c, n, s, P.shadow_terms)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, P, name, c, lab, P.s_skip(c))
    # ---------------- some live lane is inside the outer bound ----------------
    k.label(lab("hit"))
    if lazy_wake(P):
        hit_entry(k, P, shadow=True)
        k.op("s_cbranch_vccz %s" % lab("skip"), "only lanes that sleep: nobody can hit the node")
    k.op("v_sub_f32_e32 %%[av], %s, %%[ol]" % P.s_cl(c), "a ~ b = dot(centre - origin, dir)")
    k.op("v_cmp_le_f32_e64 %s, %%[a0], %%[av]" % P.M58, "b >= 0, by a margin")
    k.op("v_fma_f32 %[inn], %[av], %[av], %[p2]", "~ |centre - origin|^2")
    shadow_two_sided(k, P, P.s_r2i(c), lab("exact"), "")
    k.label(lab("decided"))
    k.op("s_bitcmp1_b32 %s, 31" % P.s_r2o(c), "an ITEM or the END node?  (sign bit of the outer bound)")
    k.op("s_cbranch_scc1 %s" % lab("flagged"))
    # BOUND: hit.distance is INF, so a bound culls iff the ray misses it
    k.op("s_andn2_b64 exec, %s, vcc" % P.ACT, "lanes that may not enter sleep until `skip`")
    k.op("v_mov_b32_e32 %%[resume], %s" % P.s_skip(c))
    k.op("s_mov_b64 exec, %s" % P.EX)
    if fused:
        k.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        k.op("v_cmp_ngt_f32_e64 vcc, %%[p2], |%s|" % P.s_r2o_own(c), "the group's own sphere: same centre, its own bounds")
        k.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        k.op("s_cbranch_vccz %s" % lab("next"))
        shadow_two_sided(k, P, P.s_r2i_own(c), lab("exactown"), "own")
        k.label(lab("owndecided"))
        k.op("s_branch .Lrt_fin_%=", "any hit ends those rays; hand them to the caller (it starts again behind this node)")
    k.label(lab("next"))
    P.load(k, n, P.NX, "somebody entered: fetch the group's first child")
    emit_next(k, P, name)
    # ITEM or END
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 31" % P.s_r2o_own(c))
    k.op("s_cbranch_scc1 .Lrt_exit_%=", "END: every lane is awake here and hits it")
    k.op("s_branch .Lrt_fin_%=")
    # ---- the reference's test, for the steps whose lanes the bounds cannot settle: terms from the exact stream
    k.label(lab("exact"))
    shadow_second_chance(k, P, P.s_r2o(c), lab("exact2"), lab("skip"), lab("decided"))
    k.label(lab("exact2"))
    tmp = "s" + P.M54[2:].split(":")[0]            # the masks are dead here: the reference's test forms the hit mask anew
    k.op("s_sub_u32 %s, %s, %d" % (tmp, P.NX, P.stride), "this node's offset")
    k.op("s_load_dwordx8 %s, %%[base2], %s" % (sp(P.bank(n), 8), tmp), "its Node record of the exact stream (the `next` bank is free until the group is entered)")
    k.op("s_waitcnt lgkmcnt(0)")
    shadow_exact(k, P, n, lab, "x", P.fld(n, 3))
    k.op("s_cbranch_vccz %s" % lab("skip"))
    k.op("v_cmp_le_f32_e64 %s, %%[a0], %%[av]" % P.M58, "(the root's scaled path may have used this mask: form `b >= 0 by a margin` again)")
    k.op("s_branch %s" % lab("decided"))
    P.tiny(k, lab("xtiny"), lab("xrooted"))
    if fused:
        k.label(lab("exactown"))
        shadow_second_chance(k, P, P.s_r2o_own(c), lab("exactown2"), lab("next"), lab("owndecided"))
        k.label(lab("exactown2"))
        k.op("s_sub_u32 %s, %s, %d" % (tmp, P.NX, P.stride), "this node's offset")
        k.op("s_waitcnt lgkmcnt(0)", "the skip successor's fetch may still be in flight INTO this bank, and scalar loads land out of order")
        k.op("s_load_dwordx8 %s, %%[base2], %s" % (sp(P.bank(s), 8), tmp), "its Node record (the skip bank is free: the group is entered)")
        k.op("s_waitcnt lgkmcnt(0)")
        shadow_exact(k, P, s, lab, "y", P.own(s))
        k.op("s_cbranch_vccz %s" % lab("next"))
        k.op("s_branch %s" % lab("owndecided"))
        P.tiny(k, lab("ytiny"), lab("yrooted"))
    return m, k


def shadow_exact64(k, P, c, lab, tag, own):
    """F64FS: the reference's own test for every lane -- the node's Node<double> record (fetched here), the sixteen f64 operations, the root where
    b < 0.  vcc = live lanes (P.ACT) whose ray hits the sphere (own: the group's own sphere, same centre, its rr)."""
    P.shadow_terms(k, c)
    if own:
        P.fused_disc(k, c)
    P.cand_cmp(k)
    k.op("s_and_b64 vcc, vcc, %s" % P.ACT)
    shadow_decide(k, P, lab, tag)


def shadow_copy_filt64(P, name, fused):
    """shadow_copy_filt for f64 scenes (F64FS): the same two-sided f32 bound in front, the f64 arithmetic behind it."""
    c, n, s = COPIES[name]
    lab = lambda x: ".Lrt_%s_%s_%%=" % (name, x)
    m, k = Asm(), Asm()
    m.label(lab("top"))
    step_top(m, P, c, n, s, P.shadow_terms)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, P, name, c, lab, P.s_skip(c))
    # ---------------- some live lane is inside the outer bound ----------------
    k.label(lab("hit"))
    if lazy_wake(P):
        hit_entry(k, P, shadow=True)
        k.op("s_cbranch_vccz %s" % lab("skip"), "only lanes that sleep: nobody can hit the node")
    k.op("v_sub_f32_e32 %%[av], %s, %%[ol]" % P.s_cl(c), "a ~ b = dot(centre - origin, dir)")
    k.op("v_cmp_le_f32_e64 %s, %%[a0], %%[av]" % P.M58, "b >= 0, by a margin")
    k.op("v_fma_f32 %[inn], %[av], %[av], %[p2]", "~ |centre - origin|^2")
    shadow_two_sided(k, P, P.s_r2i(c), lab("exact"), "")
    k.label(lab("decided"))
    k.op("s_bitcmp1_b32 %s, 31" % P.s_r2o(c), "an ITEM or the END node?  (sign bit of the outer bound)")
    k.op("s_cbranch_scc1 %s" % lab("flagged"))
    k.op("s_andn2_b64 exec, %s, vcc" % P.ACT, "lanes that may not enter sleep until `skip`")
    k.op("v_mov_b32_e32 %%[resume], %s" % P.s_skip(c))
    k.op("s_mov_b64 exec, %s" % P.EX)
    if fused:
        k.op("s_mov_b64 %s, vcc" % P.ACT, "the lanes that are live at the next node")
        k.op("v_cmp_ngt_f32_e64 vcc, %%[p2], |%s|" % P.s_r2o_own(c), "the group's own sphere: same centre, its own bounds")
        k.op("s_and_b64 vcc, vcc, %s" % P.ACT)
        k.op("s_cbranch_vccz %s" % lab("next"))
        shadow_two_sided(k, P, P.s_r2i_own(c), lab("exactown"), "own")
        k.label(lab("owndecided"))
        k.op("s_branch .Lrt_fin_%=", "any hit ends those rays; hand them to the caller (it starts again behind this node)")
    k.label(lab("next"))
    P.load(k, n, P.NX, "somebody entered: fetch the group's first child")
    emit_next(k, P, name)
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 31" % P.s_r2o_own(c))
    k.op("s_cbranch_scc1 .Lrt_exit_%=", "END: every lane is awake here and hits it")
    k.op("s_branch .Lrt_fin_%=")
    # ---- lanes between the bounds: first the behind-the-origin test, then the reference's f64 arithmetic
    k.label(lab("exact"))
    shadow_second_chance(k, P, P.s_r2o(c), lab("exact2"), lab("skip"), lab("decided"))
    k.label(lab("exact2"))
    shadow_exact64(k, P, c, lab, "x", False)
    k.op("s_cbranch_vccz %s" % lab("skip"))
    k.op("v_cmp_le_f32_e64 %s, %%[a0], %%[av]" % P.M58, "(the exact path used this mask: form `b >= 0 by a margin` again)")
    k.op("s_branch %s" % lab("decided"))
    if fused:
        k.label(lab("exactown"))
        shadow_second_chance(k, P, P.s_r2o_own(c), lab("exactown2"), lab("next"), lab("owndecided"))
        k.label(lab("exactown2"))
        shadow_exact64(k, P, c, lab, "y", True)
        k.op("s_cbranch_vccz %s" % lab("next"))
        k.op("s_branch %s" % lab("owndecided"))
    return m, k


HEADER = """// rt_skip_rot.hpp -- GENERATED by tools/gen_skip_asm.py; edit the generator, not this file.
//
// The traversal loops of k_render_skip in gfx950 assembly (f32 and f64, plain and fused).  Each is the arithmetic of the C++
// loop beside it in rt_skip.hpp (the reference implementation: every launch that counts tests), operation for operation:
//      b    = (vx*dx + vy*dy) + vz*dz              primitive.rs:57   (node terms as SGPR operands)
//      disc = (b*b - vv) + rr                      primitive.rs:58
//      root = correctly rounded sqrt(disc)         f32: y = v_rsq_f32, g = x*y, g + (x - g*g)*(y/2) in two FMAs (== sqrt_rn_lean, which is
//                                                  checked against the IEEE sqrt on all 2^32 inputs); f64: hipcc's own IEEE-correct
//                                                  expansion of __builtin_sqrt, instruction for instruction
//      t2 = b + root, t1 = b - root, d = t1 > 0 ? t1 : t2            primitive.rs:65-71
//      go = live && disc >= 0 && t2 >= 0 && d < hit.distance         (the negation of `d >= hit.distance`, group.rs:73 /
//                                                                      primitive.rs:79: a scene that passed rt_scene_create
//                                                                      cannot produce a NaN anywhere on the path, DESIGN.md 2)
// The root is only formed when some live lane has disc >= 0 (shadow rays: and b < 0, since t2 >= 0 is certain otherwise).
//
// Bookkeeping (why the loops are generated): scalar instructions compete with the vector ones for a SIMD's issue slots
// (profiles/r02_valu_issue_probe.json) and a 1080p frame is as long as its longest wave, so every instruction of a step counts.
// A position is a byte offset and every node is one stride long; the loop keeps NX, the offset behind the current node
// (`active = i >= resume` is NX > resume).  A node's likely successor (`skip`) is fetched at the top of its step and a group's
// first child when somebody enters, into two of three scalar register banks, and the step ends in the copy of the loop body (A, B, C; laid out A, C, B so
// that two of the three `skip` transitions fall through) whose current-node bank already holds the chosen successor -- no
// selects, no shifts.  The commonest step (nobody can hit the node) does not look at the node's kind: an ITEM's `skip` is the
// node behind it.  The stream ends in an END node (flag in the item word) that every lane is awake at and hits, so there is no
// end-of-stream test.  The *_fused flavour serves scenes in which every BOUND is followed by an ITEM with the same centre (the
// reference's pyramid), walking a compacted stream whose BOUND nodes carry that sphere: the BOUND step goes on to test it for
// the lanes that enter (v, b, b*b - vv are the same bits; only rr differs).
//
// Hazards follow what hipcc itself emits for gfx950: a VALU-written SGPR pair is not read as a v_cndmask mask within the
// next two instructions, a transcendental result (v_sqrt_f32, v_rsq_f64) is not consumed by the next instruction,
// s_waitcnt lgkmcnt(0) before loaded registers are read and at every exit (the speculative loads must have landed before
// their registers are free again).  64-bit selects narrow EXEC and use v_mov_b64.
//
// Node<T> (rt_skip.hpp): five geometry terms, item word (index | kNodeItem | kNodeEnd), skip as a byte offset, rr of the group's
// own sphere (fused streams, BOUND nodes).  Fixed SGPRs, f32: s[36:43] / s[44:51] / s[52:59] node banks, s60 NX, s[62:71] masks,
// s[72:73] EXEC at entry (80 SGPRs with the hardware's six: 8 waves per SIMD); f64: s[36:51] / s[52:67] / s[68:83], s85,
// s[86:95], s[96:97].
#pragma once
#include "rt_kernels.hpp"

namespace rt {

typedef float rt_v2f __attribute__((ext_vector_type(2)));      // an aligned register pair (packed_p2 in the generator)

"""

PRIMARY_FN = """// Primary-ray traversal: s.group.intersect(&mut h, r) for all 64 rays of the wave.  nodes: Node<%(ctype)s>[n + 3], END at [n];
// n_bytes = n * %(stride)d.  resume: 0 for lanes with a ray, n_bytes for lanes without (they sleep until END).  Returns
// hit.distance / item word per lane (mask the item with kNodeIndexMask).
__device__ __forceinline__ void %(name)s(const void *nodes, unsigned n_bytes, %(ctype)s dx, %(ctype)s dy, %(ctype)s dz, unsigned resume,
                                                 %(ctype)s &best_out, unsigned &item_out%(primary_extra_args)s)
{
    %(ctype)s best = %(inf)s;
    unsigned bitem = 0;
    (void)n_bytes;
    %(decl)s
    asm volatile(
%(body)s
        : [best] "+v"(best), [bitem] "+v"(bitem), [resume] "+v"(resume), %(out)s
        : [base] "s"(nodes), [n] "s"(n_bytes), [dx] "v"(dx), [dy] "v"(dy), [dz] "v"(dz)%(extra_in)s%(primary_extra_in)s
        : %(clobbers)s);
    best_out = best;
    item_out = bitem;
}

"""

SHADOW_FN = """// Shadow-ray traversal (any hit, render.rs:202-208) from byte offset `start` until the stream ends or some lane's ray hits
// an ITEM: the caller retires those lanes (resume = n_bytes), finds the next node any lane still wants and calls again.  Returns
// the byte offset it stopped at (n_bytes: stream finished); fin = 1 in the lanes that hit the ITEM there.  resume in bytes.
// hit.distance is INF throughout, so a node is "hit" iff disc >= 0 and t2 = b + root >= 0.
__device__ __forceinline__ unsigned %(name)s(const void *nodes, unsigned n_bytes, unsigned start, %(ctype)s ox, %(ctype)s oy, %(ctype)s oz,
                                                   %(ctype)s lx, %(ctype)s ly, %(ctype)s lz, unsigned &resume_io, unsigned &fin_out%(shadow_extra_args)s)
{
    unsigned resume = resume_io, fin = 0, stop;
    %(decl)s%(shadow_extra_decl)s
    asm volatile(
%(body)s
        : [resume] "+v"(resume), [fin] "+v"(fin), [stop] "=&s"(stop), %(out)s%(shadow_extra_out)s
        : [base] "s"(nodes), [n] "s"(n_bytes), [start] "s"(start), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [lx] "%(lreg)s"(lx), [ly] "%(lreg)s"(ly),
          [lz] "%(lreg)s"(lz)%(extra_in)s%(shadow_extra_in)s
        : %(clobbers)s);
    resume_io = resume;
    fin_out = fin;
    return stop;
}

"""


def clobbers(P, vregs=()):
    # Registers the compiler RESERVES in the one kernel a flavour's loops are built into are not named: it never allocates them, so there is
    # nothing to tell it -- and naming them is what `-Winline-asm` ("clobber list contains reserved registers") objects to.  That the loops
    # may use them rests on the kernel's .sgpr_count (the hardware's allocation covers them) and on no compiler-generated instruction
    # touching them: tests/test_kernel_resources.py checks both on the generated assembly (tools/check_reserved_registers.py).
    regs = ['"s%d"' % r for r in range(P.clobber_lo, P.clobber_hi + 1) if r not in getattr(P, "reserved", ())] + list(vregs)
    lines, cur = [], '"memory", "vcc", "scc"'
    for r in regs:
        if len(cur) + len(r) + 2 > 118:
            lines.append(cur + ",")
            cur = "          " + r
        else:
            cur += ", " + r
    lines.append(cur)
    return "\n".join(lines)


def assemble(a, P, copy_fn, fused):
    mains, colds = {}, {}
    for name in "ABC":
        mains[name], colds[name] = copy_fn(P, name, fused)
    for name in LAYOUT:
        a.extend(mains[name])
    for name in "ABC":
        a.extend(colds[name])


def primary(P, fused):
    a = Asm()
    a.op("s_mov_b32 %s, %d" % (P.NX, P.stride))
    a.op("s_mov_b64 %s, exec" % P.EX)
    P.load(a, 0, "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    assemble(a, P, primary_copy, fused)
    a.label(".Lrt_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def shadow(P, fused):
    a = Asm()
    a.op("s_add_u32 %s, %%[start], %d" % (P.NX, P.stride))
    a.op("s_mov_b64 %s, exec" % P.EX)
    P.load(a, 0, "%[start]")
    a.op("s_waitcnt lgkmcnt(0)")
    assemble(a, P, shadow_copy_filt64 if isinstance(P, F64FS) else shadow_copy_filt if P.filt else shadow_copy, fused)
    a.label(".Lrt_fin_%=")
    a.op("v_cndmask_b32_e64 %[fin], 0, 1, vcc", "the lanes whose ray hit the ITEM (or the group's own sphere) of the current node")
    a.op("s_sub_u32 %%[stop], %s, %d" % (P.NX, P.stride), "its position")
    a.op("s_branch .Lrt_out_%=")
    a.label(".Lrt_exit_%=")
    a.op("s_mov_b32 %[stop], %[n]", "stream finished")
    a.label(".Lrt_out_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    text = a.render()
    for name, reg in P.shadow_subst:
        text = text.replace(name, reg)
    return text


def main():
    text = HEADER
    for P in (F32(), F64(), F32F(), F64F(), F64F_LO()):
        for fused in (False, True):
            sfx = ("_filt" if P.filt else "") + getattr(P, "fn_suffix", "") + ("_fused" if fused else "")
            common = {"ctype": P.ctype, "stride": P.stride, "inf": P.inf, "extra_in": P.extra_in, "clobbers": clobbers(P), "lreg": getattr(P, "lreg", "s"),
                      "shadow_extra_in": P.shadow_extra_in, "shadow_extra_out": P.shadow_extra_out, "shadow_extra_decl": P.shadow_extra_decl,
                      "shadow_extra_args": ", float q1, float q2, float ol, float a0, float k1, float kc, const void *exact" if (P.filt and not P.primary_only) else "",
                      "primary_extra_args": P.primary_extra_args, "primary_extra_in": P.primary_extra_in}
            # (what tools/check_reserved_registers.py finds a flavour's statements by in the compiler's assembly output)
            mark = '        "\\t; rt-loops %s: undeclared s[%d:%d]\\n"\n' % (getattr(P, "mark_name", P.name), P.reserved[0], P.reserved[-1]) if getattr(P, "reserved", ()) else ""
            text += PRIMARY_FN % dict(common, name="skip_primary_rot" + sfx, body=mark + primary(P, fused), decl=P.primary_decl, out=P.primary_out)
            if not P.primary_only:
                text += SHADOW_FN % dict(common, name="skip_shadow_rot" + sfx, body=shadow(P, fused), decl=P.shadow_decl, out=P.shadow_out,
                                         clobbers=clobbers(P, P.shadow_vclobbers))
            elif isinstance(P, F64F):
                S = F64FS_LO() if isinstance(P, F64F_LO) else F64FS()
                sc = dict(common, shadow_extra_in=S.shadow_extra_in, shadow_extra_out="", shadow_extra_decl="", clobbers=clobbers(S, S.shadow_vclobbers),
                          shadow_extra_args=", float q1, float q2, float ol, float a0, float k1, float kc, const void *exact")
                text += SHADOW_FN % dict(sc, name="skip_shadow_rot" + sfx, body=mark + shadow(S, fused), decl=S.shadow_decl, out=S.shadow_out)
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
