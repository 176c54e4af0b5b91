#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_skip2_rot.hpp: the traversal loops of k_render_skip2 -- the f32 fused hierarchy walk with
TWO rays per lane on packed math -- in gfx950 assembly.

Why (tools/valu_issue_probe.hip, profiles/r02_valu_issue_probe.json, counting the LONGEST wave): on gfx950 a wave64 VOP2 issues
in 2.2 cycles, but a VOP2 whose SGPR operand differs from the previous instruction's takes 4.1 -- what every VOP3 / VOP3P
instruction takes.  Five of the eight operations of a sphere test read a node term from an SGPR, each a different one, so a test
costs 5 x 4.1 + 3 x 2.2 = 27 cycles per 64 rays.  A packed v_pk_mul/add_f32 takes the same 4.1 cycles with or without a
scalar operand (either half of an aligned SGPR pair, broadcast to both results with op_sel) and does the operation for TWO
rays: 8 x 4.1 = 33 cycles per 128 rays.  The scalar bookkeeping of a step (successor fetch, wait, position, branches) is
shared by twice the rays as well.  What does not pack is done per half on the 32-bit registers of the pair: the two `active`
compares, the two candidate compares and the exact path (root, t1, t2, d, `<` against hit.distance).

Everything else is rt_skip_rot.hpp's fused f32 loop (tools/gen_skip_asm.py, which documents the walk): byte offsets, one
stride per node, NX = offset behind the current node, `skip` fetched at the top of the step and the first child on entering,
three scalar register banks and three copies of the body laid out A, C, B, kind flags only looked at on a hit, END node.
A lane's two rays have their own `resume`, hit.distance and item; a wave walks the union of its 128 rays' nodes, so the kernel
pairs rays that share most of their walk (two samples' worth of neighbouring pixels).

The halves of a register pair have to be named, which inline-asm operands cannot do: the loops own FIXED registers (v[32:63],
s[24:73]; listed as clobbers).

Run:  python3 tools/gen_skip2_asm.py   (writes the header; the build does not need this script)."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_skip2_rot.hpp")

COPIES = {"A": (0, 1, 2), "B": (1, 0, 2), "C": (2, 0, 1)}          # current, next, skip
NEXT_COPY = {"A": "B", "B": "A", "C": "A"}
SKIP_COPY = {"A": "C", "B": "C", "C": "B"}
LAYOUT = "ACB"
FLAG_LIMIT = "0x3fffffff"
STRIDE = 32
# The loops' scalar registers start at SB.  24 (not the 36 of the one-ray loops): the filtered shadow walk needs 50, and a CU admits eight
# workgroups' worth of waves only up to .sgpr_count 80 (MI355X_MICROARCH.md, "Residency") -- s[24:73] + the hardware's six; the kernel keeps
# what it needs across the loops in s[0:23] and parks the rest in vector-register lanes (k_render_skip2 is held to 74: rt_skip2.hpp).
SB = 24
BANK = (SB, SB + 8, SB + 16)
NX = "s%d" % (SB + 24)
LIGHT = SB + 42                                                      # three registers: the shadow rays' direction (the fourth pads the pair)
SGPR_LAST, VGPR_FIRST, VGPR_LAST = SB + 45, 32, 63
RESERVED = (32, 72, 73)


def sp(first):
    return "s[%d:%d]" % (first, first + 1)


ACT = (sp(SB + 26), sp(SB + 28))          # per half: the rays awake at the current node
C = (sp(SB + 30), sp(SB + 32))            # per half: candidates -> the rays that go on (enter / hit)
M, M2, TINY, EX = sp(SB + 34), sp(SB + 40), sp(SB + 36), sp(SB + 38)


class Pair:
    def __init__(self, lo):
        self.p = "v[%d:%d]" % (lo, lo + 1)
        self.h = ("v%d" % lo, "v%d" % (lo + 1))


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


def fld(b, k):
    return BANK[b] + k


def item(b):
    return "s%d" % (BANK[b] + 5)


def thr(b):                     # primary FNode: the filter's threshold T sits where Node has its item word
    return "s%d" % (BANK[b] + 5)


def tag(b):                     # primary FNode: own rr (BOUND) / item | bit 31 (ITEM) / bits 31 + 30 (END)
    return "s%d" % (BANK[b] + 7)


def skip(b):
    return "s%d" % (BANK[b] + 6)


def own(b):
    return BANK[b] + 7


def load(a, b, off, comment=None):
    a.op("s_load_dwordx8 s[%d:%d], %%[base], %s" % (BANK[b], BANK[b] + 7, off), comment)


def pk(a, op, dst, x, y, sx=None, sy=None, neg_y=False, comment=None):
    """dst = x op y on both rays.  sx / sy: SGPR NUMBER whose value is broadcast to both rays in place of a VGPR pair."""
    sel, sel_hi = [0, 0], [1, 1]
    if sx is not None:
        x = "s[%d:%d]" % (sx & ~1, (sx & ~1) + 1)
        sel[0] = sel_hi[0] = sx & 1
    if sy is not None:
        y = "s[%d:%d]" % (sy & ~1, (sy & ~1) + 1)
        sel[1] = sel_hi[1] = sy & 1
    mods = ""
    if sel != [0, 0]:
        mods += " op_sel:[%d,%d]" % tuple(sel)
    if sel_hi != [1, 1]:
        mods += " op_sel_hi:[%d,%d]" % tuple(sel_hi)
    if neg_y:
        mods += " neg_lo:[0,1] neg_hi:[0,1]"
    a.op("v_pk_%s_f32 %s, %s, %s%s" % (op, dst, x, y, mods), comment)


def pk_fma(a, dst, x, y, z, sx=None, comment=None):
    """dst = x * y + z on both rays; sx: SGPR NUMBER broadcast to both rays in place of x."""
    sel, sel_hi = [0, 0, 0], [1, 1, 1]
    if sx is not None:
        x = "s[%d:%d]" % (sx & ~1, (sx & ~1) + 1)
        sel[0] = sel_hi[0] = sx & 1
    mods = ""
    if sel != [0, 0, 0]:
        mods += " op_sel:[%d,%d,%d]" % tuple(sel)
    if sel_hi != [1, 1, 1]:
        mods += " op_sel_hi:[%d,%d,%d]" % tuple(sel_hi)
    a.op("v_pk_fma_f32 %s, %s, %s, %s%s" % (dst, x, y, z, mods), comment)


class Regs:
    """v[32:63].  Primary: D = ray directions, BEST / BITEM the hit; shadow: D = ray origins, V the centre - origin vector."""

    def __init__(self, shadow):
        v = VGPR_FIRST
        self.DX, self.DY, self.DZ = Pair(v), Pair(v + 2), Pair(v + 4)
        self.RES = Pair(v + 6)
        self.T0, self.T1, self.T2 = Pair(v + 8), Pair(v + 10), Pair(v + 12)
        self.B, self.Q, self.DISC = Pair(v + 14), Pair(v + 16), Pair(v + 18)
        if shadow:
            self.VX, self.VY, self.VZ = Pair(v + 20), Pair(v + 22), Pair(v + 24)
            self.FIN = Pair(v + 26)
        else:
            self.BEST, self.BITEM = Pair(v + 20), Pair(v + 22)
        self.root = "v%d" % (v + 28)
        # 32-bit temporaries of the exact path: the T pairs are free once the step's arithmetic is done
        self.t0, self.t1, self.t3, self.t4, self.t5 = self.T0.h[0], self.T0.h[1], self.T1.h[0], self.T1.h[1], self.T2.h[0]


U = (TINY, sp(SB + 46))             # filtered shadow walk, per half: the candidates the bounds cannot settle (TINY is only the root's scratch)
K1 = LIGHT + 3                  # s81: k1 of the inner bound (the pad of the light's pair)
A0, KC = "s%d" % (SB + 48), "s%d" % (SB + 49)           # a >= a0 proves b >= 0;  (P2 + a^2) kc >= R2o proves the origin clearly outside
SGPR_LAST_FILT = SB + 49
FC = SB                         # FilterConsts (rt_skip.hpp) arrive in s[36:51] before the walk starts: m0, e1, e2, l, a0, k1, kc, ro2


class RegsSF:
    """v[32:63] of the FILTERED shadow walk.  Ray-persistent: the origin (D, for the reference's arithmetic), its coordinates in the
    plane perpendicular to the light (Q1, Q2) and along it (OL), RES.  Step-persistent: P2, AV.  E0..E6: temporaries -- of the bounds
    (differences, INN = P2 + AV^2, TT = P2 + k1 INN) and, when the bounds cannot settle a ray, of the reference's sixteen operations
    (V = E0..E2, products E3..E5, B = E6, then vv -> E0, Q -> E2, DISC -> E1) and of the root (halves of E3..E5)."""

    def __init__(self):
        v = VGPR_FIRST
        self.DX, self.DY, self.DZ = Pair(v), Pair(v + 2), Pair(v + 4)
        self.RES = Pair(v + 6)
        self.Q1, self.Q2, self.OL = Pair(v + 8), Pair(v + 10), Pair(v + 12)
        self.P2, self.AV = Pair(v + 14), Pair(v + 16)
        E = [Pair(v + 18 + 2 * k) for k in range(7)]
        self.E = E
        self.INN, self.TT = E[2], E[3]
        self.VX, self.VY, self.VZ = E[0], E[1], E[2]
        self.T0, self.T1, self.T2 = E[3], E[4], E[5]
        self.B, self.Q, self.DISC = E[6], E[2], E[1]
        self.t0, self.t1, self.t3, self.t4, self.t5 = E[3].h[0], E[3].h[1], E[4].h[0], E[4].h[1], E[5].h[0]
        self.root = E[5].h[1]


def refine(a, r, x):
    a.op("v_mul_f32_e32 %s, %s, %s" % (r.root, x, r.t0), "g = x*y")
    a.op("v_mul_f32_e32 %s, 0.5, %s" % (r.t0, r.t0), "h = y/2")
    a.op("v_fma_f32 %s, -%s, %s, %s" % (r.t1, r.root, r.root, x), "r = x - g*g")
    a.op("v_fma_f32 %s, %s, %s, %s" % (r.root, r.t1, r.t0, r.root), "g + r*h")


def root(a, r, disc, need_mask, done_label, tiny_label):
    """Correctly rounded sqrt(disc) into r.root (== sqrt_rn_lean).  need_mask: the lanes whose root is used."""
    a.op("v_rsq_f32_e32 %s, %s" % (r.t0, disc))
    a.op("v_cmp_lt_f32_e64 %s, |%s|, %%[tiny]" % (TINY, disc))
    a.op("s_and_b64 %s, %s, %s" % (M2, TINY, need_mask))
    a.op("s_cbranch_scc1 %s" % tiny_label, "some needed lane below 2^-96 (zero included): scaled path")
    refine(a, r, disc)
    a.label(done_label)


def tiny(a, r, disc, tiny_label, done_label):
    a.label(tiny_label)
    a.op("v_mul_f32_e32 %s, 0x5f800000, %s" % (r.t0, disc), "root with the 2^64 / 2^-32 scaling for tiny lanes (the scaled operand stays >= 2^-85: its residual is never subnormal)")
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.t5, disc, r.t0, TINY))
    a.op("v_rsq_f32_e32 %s, %s" % (r.t0, r.t5))
    a.op("v_cmp_eq_f32_e64 %s, 0, %s" % (M2, r.t5), "sqrt(+-0) = +-0 (rsq would make it 0 * inf)")
    refine(a, r, r.t5)
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.root, r.root, r.t5, M2))
    a.op("v_mul_f32_e32 %s, 0x2f800000, %s" % (r.t0, r.root))
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.root, r.root, r.t0, TINY))
    a.op("s_branch %s" % done_label)


def top_of(name):
    return ".Lr2_%s_top_%%=" % name


def emit_skip(a, name, c, lab):
    a.label(lab("skip"))
    a.op("s_add_u32 %s, %s, %d" % (NX, skip(c), STRIDE), "jump over the subtree")
    a.op("s_waitcnt lgkmcnt(0)")
    nxt = SKIP_COPY[name]
    if LAYOUT.index(nxt) != LAYOUT.index(name) + 1:
        a.op("s_branch %s" % top_of(nxt))


def emit_next(a, name):
    a.op("s_add_u32 %s, %s, %d" % (NX, NX, STRIDE))
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch %s" % top_of(NEXT_COPY[name]))


def step_top(a, r, c, s, terms):
    """The part of a step every node shares: fetch the likely successor, form both discriminants; vcc = the lanes in which some
    ray's line meets the sphere, awake or not.  Who is awake is only asked when there is such a lane (wake_check): the two
    compares are VOP3s of 4 issue cycles each, and three steps of four end at `skip` without needing them."""
    load(a, s, skip(c), "the likely successor, while this node is processed")
    terms(a, r, c)
    a.op("v_max_f32_e32 %s, %s, %s" % (r.t0, r.DISC.h[0], r.DISC.h[1]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.t0)


def wake_check(a, r, lab):
    """C[h] = the awake rays of half h whose line meets the sphere; nobody: on to `skip`."""
    for h in range(2):
        a.op("v_cmp_gt_u32_e64 %s, %s, %s" % (ACT[h], NX, r.RES.h[h]), "active = i >= resume  (NX = i + stride)" if h == 0 else None)
    a.op("v_cmp_le_f32_e64 %s, 0, %s" % (C[0], r.DISC.h[0]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.DISC.h[1])
    a.op("s_and_b64 %s, %s, %s" % (C[0], C[0], ACT[0]))
    a.op("s_and_b64 %s, vcc, %s" % (C[1], ACT[1]))
    a.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    a.op("s_cbranch_scc0 %s" % lab("skip"), "only sleeping rays: a BOUND is jumped over, an ITEM changes nothing")


def sleep_culled(a, r, c):
    """The awake rays that do not go on sleep until `skip`."""
    for h in range(2):
        a.op("s_andn2_b64 exec, %s, %s" % (ACT[h], C[h]))
        a.op("v_mov_b32_e32 %s, %s" % (r.RES.h[h], skip(c)))
    a.op("s_mov_b64 exec, %s" % EX)


def kind_test(a, c, lab):
    a.op("s_cmp_gt_u32 %s, %s" % (item(c), FLAG_LIMIT), "an ITEM or the END node?  (flag bits of the item word)")
    a.op("s_cbranch_scc1 %s" % lab("flagged"))


# ------------------------------------------------------------------------------------------------------------------ primary

def primary_terms(a, r, c):
    pk(a, "mul", r.T0.p, None, r.DX.p, sx=fld(c, 0), comment="b = (vx*dx + vy*dy) + vz*dz, both rays   primitive.rs:57")
    pk(a, "mul", r.T1.p, None, r.DY.p, sx=fld(c, 1))
    pk(a, "mul", r.T2.p, None, r.DZ.p, sx=fld(c, 2))
    pk(a, "add", r.T0.p, r.T0.p, r.T1.p)
    pk(a, "add", r.B.p, r.T0.p, r.T2.p)
    pk(a, "mul", r.T0.p, r.B.p, r.B.p, comment="disc = (b*b - vv) + rr   primitive.rs:58")
    pk(a, "add", r.Q.p, r.T0.p, None, sy=fld(c, 3), neg_y=True)
    pk(a, "add", r.DISC.p, None, r.Q.p, sx=fld(c, 4))


def primary_filter(a, r, c):
    """The conservative bound of rt_skip_rot.hpp's filtered loops for both rays: b' = fma(vz, dz, fma(vy, dy, vx*dx)) >= T, or the
    reference's test returns INF for that ray.  Three packed instructions instead of eight; T0 keeps vx*dx."""
    pk(a, "mul", r.T0.p, None, r.DX.p, sx=fld(c, 0), comment="filter: b' = fma(vz, dz, fma(vy, dy, vx*dx)) >= T for some ray?")
    pk_fma(a, r.DISC.p, None, r.DY.p, r.T0.p, sx=fld(c, 1))
    pk_fma(a, r.DISC.p, None, r.DZ.p, r.DISC.p, sx=fld(c, 2))
    a.op("v_max_f32_e32 %s, %s, %s" % (r.root, r.DISC.h[0], r.DISC.h[1]))
    a.op("v_cmp_le_f32_e32 vcc, %s, %s" % (thr(c), r.root))


def primary_terms_after_filter(a, r, c):
    pk(a, "mul", r.T1.p, None, r.DY.p, sx=fld(c, 1), comment="b = (vx*dx + vy*dy) + vz*dz, both rays   primitive.rs:57 (vx*dx is the filter's)")
    pk(a, "mul", r.T2.p, None, r.DZ.p, sx=fld(c, 2))
    pk(a, "add", r.T0.p, r.T0.p, r.T1.p)
    pk(a, "add", r.B.p, r.T0.p, r.T2.p)
    pk(a, "mul", r.T0.p, r.B.p, r.B.p, comment="disc = (b*b - vv) + rr   primitive.rs:58")
    pk(a, "add", r.Q.p, r.T0.p, None, sy=fld(c, 3), neg_y=True)
    pk(a, "add", r.DISC.p, None, r.Q.p, sx=fld(c, 4))
    a.op("v_max_f32_e32 %s, %s, %s" % (r.t0, r.DISC.h[0], r.DISC.h[1]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.t0)


def primary_go(a, r, h, tag, lab, tinies):
    """C[h] (awake rays of half h with disc >= 0) -> C[h] = go: t2 >= 0 and d < hit.distance; d left in t4.  Skipped when the
    half has no candidate."""
    done = lab("pg%s%d" % (tag, h))
    a.op("s_cmp_eq_u64 %s, 0" % C[h])
    a.op("s_cbranch_scc1 %s" % done)
    root(a, r, r.DISC.h[h], C[h], lab("rooted%s%d" % (tag, h)), lab("tiny%s%d" % (tag, h)))
    tinies.append((r.DISC.h[h], lab("tiny%s%d" % (tag, h)), lab("rooted%s%d" % (tag, h))))
    a.op("v_add_f32_e32 %s, %s, %s" % (r.t3, r.B.h[h], r.root), "t2")
    a.op("v_sub_f32_e32 %s, %s, %s" % (r.t4, r.B.h[h], r.root), "t1")
    a.op("v_cmp_lt_f32_e64 %s, 0, %s" % (M, r.t4), "t1 > 0")
    a.op("v_cmp_le_f32_e64 %s, 0, %s" % (M2, r.t3), "t2 >= 0")
    a.op("s_and_b64 %s, %s, %s" % (C[h], C[h], M2))
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.t4, r.t3, r.t4, M), "d = t1 > 0 ? t1 : t2")
    a.op("v_cmp_lt_f32_e64 %s, %s, %s" % (M, r.t4, r.BEST.h[h]), "d < hit.distance")
    a.op("s_and_b64 %s, %s, %s" % (C[h], C[h], M), "go")
    return done


def primary_update(a, r, h, c, done, own=False):
    # an ITEM records its tag (item | bit 31); a group's own sphere records WHERE it was hit (NX = the BOUND's offset + stride, bit 31
    # clear) and the kernel looks the item up in the stream's own_item table afterwards (the BOUND's tag word is the sphere's rr)
    a.op("s_mov_b64 exec, %s" % C[h], "primitive.rs:80-83")
    a.op("v_mov_b32_e32 %s, %s" % (r.BEST.h[h], r.t4))
    a.op("v_mov_b32_e32 %s, %s" % (r.BITEM.h[h], NX if own else tag(c)))
    a.op("s_mov_b64 exec, %s" % EX)
    a.label(done)


FUSED = True                    # the flavour being generated: False = plain streams (a BOUND has no sphere of its own: non-concentric scenes)
KK = SB + 46                         # s82: 1 + 2^-20 (bound_shortcut), the low half of the pair a packed instruction reads it through
G = (sp(SB + 42), sp(SB + 44))            # primary loop, per half: the rays the root-free decision lets enter (the shadow loops keep the light there)
SGPR_LAST_PRIMARY = SB + 47


def bound_shortcut(a, r, c, lab):
    """Primary BOUND step, C[h] = awake rays of half h with disc >= 0: decide `d < hit.distance` (group.rs:73) WITHOUT the root wherever the
    reference's own b and disc already settle it -- tools/gen_skip_asm.py F32F.bound_shortcut (NOTES.md A.1), here for both rays
    of a lane: w = b - hit.distance, w^2 and disc (1 + 2^-20) are one packed instruction each.  A ray nothing settles sends the whole step
    through the root (lab bexact: C[h] lost only rays with b <= 0, which that path rejects as well)."""
    a.op("s_cmp_eq_u32 %s, 0xff800000" % thr(c), "T = -inf: the eye is not clearly outside this sphere -- the reference's arithmetic decides")
    a.op("s_cbranch_scc1 %s" % lab("bexact"))
    pk(a, "add", r.T1.p, r.B.p, r.BEST.p, neg_y=True, comment="w = b - hit.distance  (-inf while nothing was hit)")
    pk(a, "mul", r.T0.p, r.DISC.p, None, sy=KK, comment="disc (1 + 2^-20)")
    pk(a, "mul", r.T2.p, r.T1.p, r.T1.p, comment="w^2")
    for h in range(2):
        a.op("v_cmp_lt_f32_e64 %s, 0, %s" % (M, r.B.h[h]), "b > 0" if h == 0 else None)
        a.op("s_and_b64 %s, %s, %s" % (C[h], C[h], M), "candidates in front of the eye" if h == 0 else None)
        a.op("v_cmp_gt_f32_e64 %s, 0, %s" % (G[h], r.T1.h[h]), "b < hit.distance: enters" if h == 0 else None)
        a.op("v_cmp_le_f32_e64 %s, %s, %s" % (M2, r.T0.h[h], r.T2.h[h]), "the root cannot reach down to b - hit.distance: culled" if h == 0 else None)
        a.op("s_or_b64 %s, %s, %s" % (M2, M2, G[h]), "settled rays" if h == 0 else None)
        a.op("s_andn2_b64 %s, %s, %s" % (M2, C[h], M2), "candidates nothing above settles" if h == 0 else None)
        a.op("s_cbranch_scc1 %s" % lab("bexact"))
    for h in range(2):
        a.op("s_and_b64 %s, %s, %s" % (C[h], C[h], G[h]), "go" if h == 0 else None)
    a.op("s_branch %s" % lab("bdecided"))


def primary_copy(r, name):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lr2_%s_%s_%%=" % (name, x)
    m, k, tinies = Asm(), Asm(), []
    m.label(lab("top"))
    load(m, s, skip(c), "the likely successor, while this node is processed")
    primary_filter(m, r, c)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, name, c, lab)
    # ---------------- the bound cannot rule the node out for some ray: the reference's discriminant, for every ray ----------------
    k.label(lab("hit"))
    primary_terms_after_filter(k, r, c)
    k.op("s_cbranch_vccz %s" % lab("skip"), "the bound let it through, the test does not: nobody can hit the node")
    wake_check(k, r, lab)
    k.op("s_bitcmp1_b32 %s, 31" % tag(c), "an ITEM or the END node?  (flag bits of the tag word)")
    k.op("s_cbranch_scc1 %s" % lab("flagged"))
    # BOUND (group.rs:73)
    bound_shortcut(k, r, c, lab)
    k.label(lab("bexact"))
    for h in range(2):
        k.label(primary_go(k, r, h, "b", lab, tinies))
    k.label(lab("bdecided"))
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("skip"), "nobody enters (the rays that culled it are awake again at `skip`)")
    load(k, n, NX, "somebody enters: fetch the group's first child")
    sleep_culled(k, r, c)
    if FUSED:
        # the group's own sphere, for the rays that entered: same centre, so v, b and b*b - vv are the values just formed
        pk(k, "add", r.DISC.p, None, r.Q.p, sx=own(c), comment="disc = (b*b - vv) + rr of the group's own sphere")
        k.op("v_cmp_le_f32_e64 %s, 0, %s" % (M, r.DISC.h[0]))
        k.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.DISC.h[1])
        k.op("s_and_b64 %s, %s, %s" % (C[0], C[0], M))
        k.op("s_and_b64 %s, %s, vcc" % (C[1], C[1]))
        k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
        k.op("s_cbranch_scc0 %s" % lab("next"))
        for h in range(2):
            primary_update(k, r, h, c, primary_go(k, r, h, "f", lab, tinies), own=True)
    k.label(lab("next"))
    emit_next(k, name)
    # ITEM (primitive.rs:77-84) or END
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 30" % tag(c))
    k.op("s_cbranch_scc1 .Lr2_exit_%=", "END: every ray is awake here and hits it")
    for h in range(2):
        primary_update(k, r, h, c, primary_go(k, r, h, "i", lab, tinies))
    k.op("s_branch %s" % lab("skip"), "an ITEM's `skip` is the node behind it")
    for disc, tl, dl in tinies:
        tiny(k, r, disc, tl, dl)
    return m, k


# ------------------------------------------------------------------------------------------------------------------- shadow

def shadow_terms(a, r, c):
    pk(a, "add", r.VX.p, None, r.DX.p, sx=fld(c, 0), neg_y=True, comment="v = centre - origin, both rays   primitive.rs:56")
    pk(a, "add", r.VY.p, None, r.DY.p, sx=fld(c, 1), neg_y=True)
    pk(a, "add", r.VZ.p, None, r.DZ.p, sx=fld(c, 2), neg_y=True)
    pk(a, "mul", r.T0.p, None, r.VX.p, sx=LIGHT + 0)
    pk(a, "mul", r.T1.p, None, r.VY.p, sx=LIGHT + 1)
    pk(a, "mul", r.T2.p, None, r.VZ.p, sx=LIGHT + 2)
    pk(a, "add", r.T0.p, r.T0.p, r.T1.p)
    pk(a, "add", r.B.p, r.T0.p, r.T2.p, comment="b = dot(v, dir)   primitive.rs:57")
    pk(a, "mul", r.VX.p, r.VX.p, r.VX.p)
    pk(a, "mul", r.VY.p, r.VY.p, r.VY.p)
    pk(a, "mul", r.VZ.p, r.VZ.p, r.VZ.p)
    pk(a, "add", r.VX.p, r.VX.p, r.VY.p)
    pk(a, "add", r.VX.p, r.VX.p, r.VZ.p, comment="dot(v, v)")
    pk(a, "mul", r.T0.p, r.B.p, r.B.p)
    pk(a, "add", r.Q.p, r.T0.p, r.VX.p, neg_y=True)
    pk(a, "add", r.DISC.p, None, r.Q.p, sx=fld(c, 3), comment="disc = (b*b - vv) + rr   primitive.rs:58")


def shadow_decide(a, r, h, tag, lab, tinies):
    """C[h] (awake rays of half h with disc >= 0) -> C[h] = the rays that hit the sphere (t2 >= 0; certain when b >= 0)."""
    done = lab("sd%s%d" % (tag, h))
    a.op("v_cmp_gt_f32_e64 %s, 0, %s" % (M, r.B.h[h]), "b < 0: t2 may still be negative")
    a.op("s_and_b64 %s, %s, %s" % (M, M, C[h]))
    a.op("s_cbranch_scc0 %s" % done, "nobody needs the root: hit = candidates")
    root(a, r, r.DISC.h[h], M, lab("rooted%s%d" % (tag, h)), lab("tiny%s%d" % (tag, h)))
    tinies.append((r.DISC.h[h], lab("tiny%s%d" % (tag, h)), lab("rooted%s%d" % (tag, h))))
    a.op("v_add_f32_e32 %s, %s, %s" % (r.t3, r.B.h[h], r.root), "t2")
    a.op("v_cmp_gt_f32_e64 %s, 0, %s" % (M2, r.t3), "t2 < 0")
    a.op("s_and_b64 %s, %s, %s" % (M2, M2, M), "root lanes that miss after all")
    a.op("s_andn2_b64 %s, %s, %s" % (C[h], C[h], M2))
    a.label(done)


def shadow_copy(r, name):
    c, n, s = COPIES[name]
    lab = lambda x: ".Lr2_%s_%s_%%=" % (name, x)
    m, k, tinies = Asm(), Asm(), []
    m.label(lab("top"))
    step_top(m, r, c, s, shadow_terms)
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    emit_skip(m, name, c, lab)
    k.label(lab("hit"))
    wake_check(k, r, lab)
    kind_test(k, c, lab)
    # BOUND: hit.distance is INF, so a bound culls iff the ray misses it
    for h in range(2):
        shadow_decide(k, r, h, "b", lab, tinies)
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("skip"))
    load(k, n, NX, "somebody enters: fetch the group's first child")
    sleep_culled(k, r, c)
    pk(k, "add", r.DISC.p, None, r.Q.p, sx=own(c), comment="disc = (b*b - vv) + rr of the group's own sphere")
    k.op("v_cmp_le_f32_e64 %s, 0, %s" % (M, r.DISC.h[0]))
    k.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.DISC.h[1])
    k.op("s_and_b64 %s, %s, %s" % (C[0], C[0], M))
    k.op("s_and_b64 %s, %s, vcc" % (C[1], C[1]))
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("next"))
    for h in range(2):
        shadow_decide(k, r, h, "f", lab, tinies)
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc1 .Lr2_fin_%=", "any hit ends those rays; hand them to the caller")
    k.label(lab("next"))
    emit_next(k, name)
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 30" % item(c))
    k.op("s_cbranch_scc1 .Lr2_exit_%=", "END: every ray is awake here and hits it")
    for h in range(2):
        shadow_decide(k, r, h, "i", lab, tinies)
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("skip"), "an ITEM's `skip` is the node behind it")
    k.op("s_branch .Lr2_fin_%=")
    for disc, tl, dl in tinies:
        tiny(k, r, disc, tl, dl)
    return m, k

# ---------------------------------------------------------------------------------------------------------- shadow, filtered

def fs(b, k):
    """FNodeS (rt_skip.hpp): {w1, w2, cl, R2o | ITEM flag, R2i, R2o_own | END flag, R2i_own, skip_off}."""
    return BANK[b] + k


def s_r2o(b): return "s%d" % fs(b, 3)
def s_r2i(b): return "s%d" % fs(b, 4)
def s_r2o_own(b): return "s%d" % fs(b, 5)
def s_r2i_own(b): return "s%d" % fs(b, 6)
def s_skip(b): return "s%d" % fs(b, 7)


def outer_cmp(a, r, r2o):
    """C[h] = the rays of half h that are NOT beyond the outer bound (a NaN -- a ray the bounds do not cover -- passes)."""
    for h in range(2):
        a.op("v_cmp_ngt_f32_e64 %s, %s, |%s|" % (C[h], r.P2.h[h], r2o), "P2 > R2o: the reference's test says miss" if h == 0 else None)


def inner_terms(a, r):
    pk_fma(a, r.INN.p, r.AV.p, r.AV.p, r.P2.p, comment="~ |centre - origin|^2")
    pk_fma(a, r.TT.p, None, r.INN.p, r.P2.p, sx=K1, comment="P2 + k1 |centre - origin|^2: the reference's rounding of disc grows with the distance")


def two_sided(a, r, r2i, exact_label):
    """C[h] = awake rays inside the outer bound.  A ray is a SURE hit if it is inside the inner bound (disc >= 0 by a margin) and
    t2 = b + root >= 0 for certain (b >= 0 by a margin, or the origin inside the sphere).  U[h] = the candidates that are not sure."""
    for h in range(2):
        a.op("v_cmp_le_f32_e64 %s, %s, %s" % (M, A0, r.AV.h[h]), "b >= 0, by a margin")
        a.op("v_cmp_le_f32_e64 %s, %s, %s" % (M2, r.INN.h[h], r2i), "origin inside the sphere, by a margin")
        a.op("s_or_b64 %s, %s, %s" % (M, M, M2))
        a.op("v_cmp_le_f32_e64 %s, %s, %s" % (M2, r.TT.h[h], r2i), "inside the inner bound")
        a.op("s_and_b64 %s, %s, %s" % (M, M, M2), "sure hits")
        a.op("s_andn2_b64 %s, %s, %s" % (U[h], C[h], M), "candidates between the bounds")
    a.op("s_or_b64 %s, %s, %s" % (M, U[0], U[1]))
    a.op("s_cbranch_scc1 %s" % exact_label)


def second_chance(k, r, r2o, exact2, none_label, some_label):
    """Most rays between the bounds have the sphere BEHIND them: b < 0 by a margin while the origin is clearly outside -- the
    reference's test says miss.  Settles U[h]; sure misses leave C[h]."""
    for h in range(2):
        k.op("v_mul_f32_e32 %s, %s, %s" % (r.T1.h[h], KC, r.INN.h[h]), "(P2 + a^2) / (1 + 4 tau)")
        k.op("v_cmp_le_f32_e64 %s, |%s|, %s" % (M, r2o, r.T1.h[h]), "the origin is clearly outside the sphere")
        k.op("v_cmp_le_f32_e64 %s, %s, -%s" % (M2, r.AV.h[h], A0), "b < 0, by a margin")
        k.op("s_and_b64 %s, %s, %s" % (M, M, M2), "sure misses")
        k.op("s_andn2_b64 %s, %s, %s" % (U[h], U[h], M))
        k.op("s_andn2_b64 %s, %s, %s" % (C[h], C[h], M))
    k.op("s_or_b64 %s, %s, %s" % (M, U[0], U[1]))
    k.op("s_cbranch_scc1 %s" % exact2, "still between the bounds: the reference's arithmetic decides")
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % none_label)
    k.op("s_branch %s" % some_label)


def exact_packed(k, r, src, rr, lab, tag, tinies):
    """The reference's own test (primitive.rs:55-72) for both rays of every lane, node terms from bank `src` (a Node<float> record of
    the exact stream): C[h] = awake rays (ACT[h]) that hit the sphere with squared radius SGPR number `rr`."""
    pk(k, "add", r.VX.p, None, r.DX.p, sx=fld(src, 0), neg_y=True, comment="v = centre - origin, both rays   primitive.rs:56")
    pk(k, "add", r.VY.p, None, r.DY.p, sx=fld(src, 1), neg_y=True)
    pk(k, "add", r.VZ.p, None, r.DZ.p, sx=fld(src, 2), neg_y=True)
    pk(k, "mul", r.T0.p, None, r.VX.p, sx=LIGHT + 0)
    pk(k, "mul", r.T1.p, None, r.VY.p, sx=LIGHT + 1)
    pk(k, "mul", r.T2.p, None, r.VZ.p, sx=LIGHT + 2)
    pk(k, "add", r.T0.p, r.T0.p, r.T1.p)
    pk(k, "add", r.B.p, r.T0.p, r.T2.p, comment="b = dot(v, dir)   primitive.rs:57")
    pk(k, "mul", r.VX.p, r.VX.p, r.VX.p)
    pk(k, "mul", r.VY.p, r.VY.p, r.VY.p)
    pk(k, "mul", r.VZ.p, r.VZ.p, r.VZ.p)
    pk(k, "add", r.VX.p, r.VX.p, r.VY.p)
    pk(k, "add", r.VX.p, r.VX.p, r.VZ.p, comment="dot(v, v)")
    pk(k, "mul", r.T0.p, r.B.p, r.B.p)
    pk(k, "add", r.Q.p, r.T0.p, r.VX.p, neg_y=True)
    pk(k, "add", r.DISC.p, None, r.Q.p, sx=rr, comment="disc = (b*b - vv) + rr   primitive.rs:58")
    k.op("v_cmp_le_f32_e64 %s, 0, %s" % (C[0], r.DISC.h[0]))
    k.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.DISC.h[1])
    k.op("s_and_b64 %s, %s, %s" % (C[0], C[0], ACT[0]))
    k.op("s_and_b64 %s, vcc, %s" % (C[1], ACT[1]))
    for h in range(2):
        shadow_decide(k, r, h, tag, lab, tinies)


def shadow_copy_filt(r, name):
    """Two-sided flavour of shadow_copy: the walk reads the FNodeS stream (%[base]); a step that leaves some ray between the bounds
    fetches the node's Node record from the exact stream (%[base2]) and runs the reference's arithmetic for every ray."""
    c, n, s = COPIES[name]
    lab = lambda x: ".Lr2_%s_%s_%%=" % (name, x)
    m, k, tinies = Asm(), Asm(), []
    m.label(lab("top"))
    load(m, s, s_skip(c), "the likely successor, while this node is processed")
    pk(m, "add", r.E[0].p, None, r.Q1.p, sx=fs(c, 0), neg_y=True, comment="P2 = |w - q|^2 in the plane perpendicular to the light, both rays")
    pk(m, "add", r.E[1].p, None, r.Q2.p, sx=fs(c, 1), neg_y=True)
    pk(m, "mul", r.E[0].p, r.E[0].p, r.E[0].p)
    pk_fma(m, r.P2.p, r.E[1].p, r.E[1].p, r.E[0].p)
    m.op("v_min_f32_e32 %s, %s, %s" % (r.E[0].h[0], r.P2.h[0], r.P2.h[1]))
    m.op("v_cmp_ngt_f32_e64 vcc, %s, |%s|" % (r.E[0].h[0], s_r2o(c)), "some ray not beyond the outer bound?  (NaN: a wave with an uncovered ray passes)")
    m.op("s_cbranch_vccnz %s" % lab("hit"))
    m.label(lab("skip"))
    m.op("s_add_u32 %s, %s, %d" % (NX, s_skip(c), STRIDE), "jump over the subtree")
    m.op("s_waitcnt lgkmcnt(0)")
    if LAYOUT.index(SKIP_COPY[name]) != LAYOUT.index(name) + 1:
        m.op("s_branch %s" % top_of(SKIP_COPY[name]))
    # ---------------- some ray (awake or not) is inside the outer bound ----------------
    k.label(lab("hit"))
    for h in range(2):
        k.op("v_cmp_gt_u32_e64 %s, %s, %s" % (ACT[h], NX, r.RES.h[h]), "active = i >= resume  (NX = i + stride)" if h == 0 else None)
    outer_cmp(k, r, s_r2o(c))
    for h in range(2):
        k.op("s_and_b64 %s, %s, %s" % (C[h], C[h], ACT[h]))
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("skip"), "only sleeping rays: a BOUND is jumped over, an ITEM changes nothing")
    pk(k, "add", r.AV.p, None, r.OL.p, sx=fs(c, 2), neg_y=True, comment="a ~ b = dot(centre - origin, dir)")
    inner_terms(k, r)
    two_sided(k, r, s_r2i(c), lab("exact"))
    k.label(lab("decided"))
    k.op("s_bitcmp1_b32 %s, 31" % s_r2o(c), "an ITEM or the END node?  (sign bit of the outer bound)")
    k.op("s_cbranch_scc1 %s" % lab("flagged"))
    # BOUND: hit.distance is INF, so a bound culls iff the ray misses it
    for h in range(2):
        k.op("s_andn2_b64 exec, %s, %s" % (ACT[h], C[h]), "rays that may not enter sleep until `skip`" if h == 0 else None)
        k.op("v_mov_b32_e32 %s, %s" % (r.RES.h[h], s_skip(c)))
    k.op("s_mov_b64 exec, %s" % EX)
    if FUSED:
        for h in range(2):
            k.op("s_mov_b64 %s, %s" % (ACT[h], C[h]), "the rays that are awake at the next node" if h == 0 else None)
        outer_cmp(k, r, s_r2o_own(c))
        for h in range(2):
            k.op("s_and_b64 %s, %s, %s" % (C[h], C[h], ACT[h]), "the group's own sphere: same centre, its own bounds" if h == 0 else None)
        k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
        k.op("s_cbranch_scc0 %s" % lab("next"))
        two_sided(k, r, s_r2i_own(c), lab("exactown"))
        k.label(lab("owndecided"))
        k.op("s_branch .Lr2_fin_%=", "any hit ends those rays; hand them to the caller (it starts again behind this node)")
    k.label(lab("next"))
    load(k, n, NX, "somebody entered: fetch the group's first child")
    emit_next(k, name)
    # ITEM or END
    k.label(lab("flagged"))
    k.op("s_bitcmp1_b32 %s, 31" % s_r2o_own(c))
    k.op("s_cbranch_scc1 .Lr2_exit_%=", "END: every ray is awake here and hits it")
    k.op("s_branch .Lr2_fin_%=")
    # ---- rays between the bounds
    tmp = "s" + M[2:].split(":")[0]                 # the scratch mask is dead here
    k.label(lab("exact"))
    second_chance(k, r, s_r2o(c), lab("exact2"), lab("skip"), lab("decided"))
    k.label(lab("exact2"))
    k.op("s_sub_u32 %s, %s, %d" % (tmp, NX, STRIDE), "this node's offset")
    k.op("s_load_dwordx8 s[%d:%d], %%[base2], %s" % (BANK[n], BANK[n] + 7, tmp), "its Node record of the exact stream (the `next` bank is free until the group is entered)")
    k.op("s_waitcnt lgkmcnt(0)")
    exact_packed(k, r, n, fld(n, 3), lab, "x", tinies)
    k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    k.op("s_cbranch_scc0 %s" % lab("skip"))
    inner_terms(k, r)                                # the own sphere's bounds read INN and TT again
    k.op("s_branch %s" % lab("decided"))
    if FUSED:
        k.label(lab("exactown"))
        second_chance(k, r, s_r2o_own(c), lab("exactown2"), lab("next"), lab("owndecided"))
        k.label(lab("exactown2"))
        k.op("s_sub_u32 %s, %s, %d" % (tmp, NX, STRIDE), "this node's offset")
        k.op("s_waitcnt lgkmcnt(0)", "the skip successor's fetch may still be in flight INTO this bank, and scalar loads land out of order")
        k.op("s_load_dwordx8 s[%d:%d], %%[base2], %s" % (BANK[s], BANK[s] + 7, tmp), "its Node record (the skip bank is free: the group is entered)")
        k.op("s_waitcnt lgkmcnt(0)")
        exact_packed(k, r, s, own(s), lab, "y", tinies)
        k.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
        k.op("s_cbranch_scc0 %s" % lab("next"))
        k.op("s_branch %s" % lab("owndecided"))
    for disc, tl, dl in tinies:
        tiny(k, r, disc, tl, dl)
    return m, k


def shadow_filt():
    """One invocation walks the whole stream: a ray that hits an ITEM (or a group's own sphere) retires INSIDE the loop -- resume =
    n + 1 (asleep at every node, told apart from the rays that never had a shadow ray: resume = n) -- and the walk goes on at the first
    node any ray still wants (the wave minimum of max(resume, NX), a DPP reduction).  The operands are dead once copied in."""
    a, r = Asm(), RegsSF()
    a.op("s_load_dwordx16 s[%d:%d], %%[fc], 0x0" % (FC, FC + 15), "FilterConsts: m0, e1, e2, l, a0, k1, kc, ro2")
    a.op("s_mov_b64 %s, exec" % EX)
    a.op("s_waitcnt lgkmcnt(0)")
    # (origins and resume arrive in the loop's own registers: see primary())
    # shadow_filter_origin (rt_skip.hpp) for both rays: q = ((o - m0) . e1, (o - m0) . e2), ol = (o - m0) . l, FMA chains in its order
    x, y, z, d2 = r.E[0], r.E[1], r.E[2], r.E[3]
    pk(a, "add", x.p, r.DX.p, None, sy=FC + 0, neg_y=True, comment="o - m0")
    pk(a, "add", y.p, r.DY.p, None, sy=FC + 1, neg_y=True)
    pk(a, "add", z.p, r.DZ.p, None, sy=FC + 2, neg_y=True)
    for dst, k0 in ((r.Q1, FC + 3), (r.Q2, FC + 6), (r.OL, FC + 9)):
        pk(a, "mul", dst.p, None, x.p, sx=k0)
        pk_fma(a, dst.p, None, y.p, dst.p, sx=k0 + 1)
        pk_fma(a, dst.p, None, z.p, dst.p, sx=k0 + 2)
    pk(a, "mul", d2.p, x.p, x.p)
    pk_fma(a, d2.p, y.p, y.p, d2.p)
    pk_fma(a, d2.p, z.p, z.p, d2.p)
    # An origin the constants do not cover (further than sqrt(ro2) from m0) must pass every outer bound and fail every sure test: q1 = NaN.
    # The step's outer test looks at min(P2 of ray 0, P2 of ray 1), which would drop a single NaN -- so one uncovered shadow ray makes
    # q1 NaN for the whole wave (none is on a scene the library built the constants for; such a wave runs the reference's arithmetic).
    for h in range(2):
        a.op("v_cmp_nle_f32_e64 %s, %s, s%d" % (C[h], d2.h[h], FC + 15))
        a.op("v_cmp_eq_u32_e64 %s, 0, %s" % (ACT[h], r.RES.h[h]), "only rays that have a shadow ray count")
        a.op("s_and_b64 %s, %s, %s" % (C[h], C[h], ACT[h]))
    a.op("s_or_b64 %s, %s, %s" % (M, C[0], C[1]))
    a.op("s_cbranch_scc0 .Lr2_covered_%=")
    for h in range(2):
        a.op("v_mov_b32_e32 %s, 0x7fc00000" % r.Q1.h[h])
    a.label(".Lr2_covered_%=")
    for k in range(3):
        a.op("s_mov_b32 s%d, s%d" % (LIGHT + k, FC + 9 + k), "the shadow rays' direction, -light (render.rs:206)" if k == 0 else None)
    a.op("s_mov_b32 %s, s%d" % (A0, FC + 12))
    a.op("s_mov_b32 s%d, s%d" % (K1, FC + 13))
    a.op("s_mov_b32 %s, s%d" % (KC, FC + 14))
    a.op("s_mov_b32 %s, %d" % (NX, STRIDE))
    load(a, 0, "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    assemble(a, r, shadow_copy_filt)
    # ---- some ray hit the ITEM (or the group's own sphere) of the current node: C[h]
    tmp = "s" + M[2:].split(":")[0]
    w, t = r.E[0].h[0], (r.E[0].h[1], r.E[1].h[0])
    a.label(".Lr2_fin_%=")
    a.op("s_waitcnt lgkmcnt(0)", "a successor fetch may still be in flight into the bank the walk restarts in")
    a.op("s_add_u32 %s, %%[n], 1" % tmp)
    for h in range(2):
        a.op("s_mov_b64 exec, %s" % C[h], "render.rs:208 only asks has_missed(): the ray is done" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, %s" % (r.RES.h[h], tmp))
    a.op("s_mov_b64 exec, -1")
    a.op("v_mov_b32_e32 %s, %%[n]" % w, "lanes the wave entered without take no part")
    a.op("s_mov_b64 exec, %s" % EX)
    for h in range(2):
        a.op("v_max_u32_e32 %s, %s, %s" % (t[h], NX, r.RES.h[h]), "the next node the ray wants: resume > i ? resume : i + stride" if h == 0 else None)
    a.op("v_min_u32_e32 %s, %s, %s" % (w, t[0], t[1]))
    a.op("v_min_u32_e32 %s, %%[n], %s" % (w, w), "retired rays (n, n + 1) want nothing")
    a.op("s_mov_b64 exec, -1")
    for ctrl in ("row_shr:1 row_mask:0xf", "row_shr:2 row_mask:0xf", "row_shr:4 row_mask:0xf", "row_shr:8 row_mask:0xf",
                 "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
        a.op("s_nop 1")
        a.op("v_min_u32_dpp %s, %s, %s %s bank_mask:0xf" % (w, w, w, ctrl))
    a.op("s_nop 1")
    a.op("v_readlane_b32 %s, %s, 63" % (tmp, w), "the wave's minimum")
    a.op("s_mov_b64 exec, %s" % EX)
    a.op("s_cmp_ge_u32 %s, %%[n]" % tmp)
    a.op("s_cbranch_scc1 .Lr2_exit_%=", "nobody is left")
    a.op("s_add_u32 %s, %s, %d" % (NX, tmp, STRIDE))
    load(a, 0, tmp)
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_branch %s" % top_of("A"))
    a.label(".Lr2_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def assemble(a, r, copy_fn):
    mains, colds = {}, {}
    for name in "ABC":
        mains[name], colds[name] = copy_fn(r, name)
    for name in LAYOUT:
        a.extend(mains[name])
    for name in "ABC":
        a.extend(colds[name])


def primary():
    a, r = Asm(), Regs(False)
    # (the rays' directions, resume and the results ARE the loop's registers: the statement binds its operands to v32.. directly -- eight
    # copies and eight registers fewer at the statement, which is what stood between k_render_skip2 and its eighth wave per SIMD)
    for h in range(2):
        a.op("v_mov_b32_e32 %s, 0x7f800000" % r.BEST.h[h], "hit.distance = INF (primitive.rs:96)" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, 0" % r.BITEM.h[h])
    a.op("s_mov_b32 %s, %d" % (NX, STRIDE))
    a.op("s_mov_b64 %s, exec" % EX)
    a.op("s_mov_b32 s%d, %%[kk]" % KK, "1 + 2^-20 where a packed instruction can read it")
    load(a, 0, "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    assemble(a, r, primary_copy)
    a.label(".Lr2_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    return a.render()


def shadow():
    a, r = Asm(), Regs(True)
    for h in range(2):
        a.op("v_mov_b32_e32 %s, %%[ox%d]" % (r.DX.h[h], h), "operands into the loop's own registers" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, %%[oy%d]" % (r.DY.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[oz%d]" % (r.DZ.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[res%d]" % (r.RES.h[h], h))
        a.op("v_mov_b32_e32 %s, 0" % r.FIN.h[h])
    a.op("s_mov_b32 s%d, %%[lx]" % (LIGHT + 0))
    a.op("s_mov_b32 s%d, %%[ly]" % (LIGHT + 1))
    a.op("s_mov_b32 s%d, %%[lz]" % (LIGHT + 2))
    a.op("s_add_u32 %s, %%[start], %d" % (NX, STRIDE))
    a.op("s_mov_b64 %s, exec" % EX)
    load(a, 0, "%[start]")
    a.op("s_waitcnt lgkmcnt(0)")
    assemble(a, r, shadow_copy)
    a.label(".Lr2_fin_%=")
    for h in range(2):
        a.op("v_cndmask_b32_e64 %s, 0, 1, %s" % (r.FIN.h[h], C[h]), "the rays that hit the ITEM (or the group's own sphere) of the current node" if h == 0 else None)
    a.op("s_sub_u32 %%[stop], %s, %d" % (NX, STRIDE), "its position")
    a.op("s_branch .Lr2_out_%=")
    a.label(".Lr2_exit_%=")
    a.op("s_mov_b32 %[stop], %[n]", "stream finished")
    a.label(".Lr2_out_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    for h in range(2):
        a.op("v_mov_b32_e32 %%[fin%d], %s" % (h, r.FIN.h[h]))
        a.op("v_mov_b32_e32 %%[rout%d], %s" % (h, r.RES.h[h]))
    return a.render()


HEADER = """// rt_skip2_rot.hpp -- GENERATED by tools/gen_skip2_asm.py; edit the generator, not this file.
//
// The traversal loops of k_render_skip2 in gfx950 assembly: rt_skip_rot.hpp's fused f32 walk with TWO rays per lane.  The
// eight operations of a sphere test are one v_pk_mul/add_f32 each for the lane's two rays, the node term the low or high half
// of an aligned SGPR pair broadcast with op_sel -- a packed instruction issues in the same 4.1 cycles as the VOP2 that reads a
// new SGPR does for one ray (tools/valu_issue_probe.hip) -- and the successor fetch, the wait, the position and the branches of
// a step serve 128 rays.  `active`, the candidate test and the exact path (root == sqrt_rn_lean, t2, t1, d, strict `<`) run per
// half on the 32-bit registers of the pairs; each ray has its own resume, hit.distance and item.  Same arithmetic, operation
// for operation, and the same walk as the one-ray loops (tools/gen_skip_asm.py documents it).
//
// The loops own v[32:63] and s[24:73] (clobbers): s[24:47] three node banks, s48 NX, s[50:53] awake masks, s[54:57] candidate /
// go masks, s[58:65] scratch masks and EXEC at entry, s[66:68] the shadow rays' direction (primary loop: s[66:69] the masks of the
// root-free BOUND decision, s70 its constant), s[70:73] the filtered shadow walk's second scratch mask and constants.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

"""

PRIMARY_FN = """// Primary-ray traversal: s.group.intersect(&mut h, r) for the wave's 128 rays.  nodes: the compacted FILTERED stream FNode[n + 3]
// (rt_skip.hpp: {vx, vy, vz, vv, rr, T, skip_off, tag}), END at [n]: a step first asks the conservative bound b' >= T for both rays
// (three packed instructions) and forms the reference's discriminant only when some ray passes.
// resume[h]: 0 for a ray, n * 32 for a lane half without one (it sleeps until END).  Returns hit.distance and, per ray, either
// item | bit 31 or -- bit 31 clear, a group's own sphere -- the byte offset behind its BOUND node (look the item up in own_item).
__device__ __forceinline__ void %(name)s(const void *nodes, const float (&dx)[2], const float (&dy)[2], const float (&dz)[2],
                                                        const unsigned (&resume)[2], float (&best_out)[2], unsigned (&item_out)[2])
{
    const float tiny = 0x1p-96f, kk = 0x1.00001p+0f;       // kk = 1 + 2^-20 (bound_shortcut)
    unsigned res0 = resume[0], res1 = resume[1];          // (the loop's RES pair; what it leaves there is of no interest)
    asm volatile(
        "\\t; rt-loops two-ray: undeclared s32, s[72:73]\\n"
%(body)s
        : [best0] "={v52}"(best_out[0]), [best1] "={v53}"(best_out[1]), [item0] "={v54}"(item_out[0]), [item1] "={v55}"(item_out[1]),
          [res0] "+{v38}"(res0), [res1] "+{v39}"(res1)
        : [base] "s"(nodes), [dx0] "{v32}"(dx[0]), [dx1] "{v33}"(dx[1]), [dy0] "{v34}"(dy[0]), [dy1] "{v35}"(dy[1]), [dz0] "{v36}"(dz[0]), [dz1] "{v37}"(dz[1]),
          [tiny] "s"(tiny), [kk] "s"(kk)
        : %(clobbers)s);
}

"""

SHADOW_FN = """// Shadow-ray traversal (any hit, render.rs:202-208) from byte offset `start` until the stream ends or some ray hits an ITEM: the
// caller retires those rays (resume = n_bytes), finds the next node any ray still wants and calls again.  Returns the byte offset
// it stopped at (n_bytes: stream finished); fin[h] = 1 for the rays that hit the ITEM there.  hit.distance is INF throughout, so
// a node is "hit" iff disc >= 0 and t2 = b + root >= 0.
__device__ __forceinline__ unsigned skip2_shadow_rot_fused(const void *nodes, unsigned n_bytes, unsigned start, const float (&ox)[2],
                                                           const float (&oy)[2], const float (&oz)[2], float lx, float ly, float lz,
                                                           unsigned (&resume)[2], unsigned (&fin)[2])
{
    const float tiny = 0x1p-96f;
    unsigned stop;
    start = (unsigned)__builtin_amdgcn_readfirstlane((int)start);       // wave-uniform by construction; the operand must be an SGPR
    asm volatile(
        "\\t; rt-loops two-ray: undeclared s32, s[72:73]\\n"
%(body)s
        : [fin0] "=v"(fin[0]), [fin1] "=v"(fin[1]), [rout0] "=v"(resume[0]), [rout1] "=v"(resume[1]), [stop] "=&s"(stop)
        : [base] "s"(nodes), [n] "s"(n_bytes), [start] "s"(start), [ox0] "v"(ox[0]), [ox1] "v"(ox[1]), [oy0] "v"(oy[0]), [oy1] "v"(oy[1]),
          [oz0] "v"(oz[0]), [oz1] "v"(oz[1]), [res0] "v"(resume[0]), [res1] "v"(resume[1]), [lx] "s"(lx), [ly] "s"(ly), [lz] "s"(lz),
          [tiny] "s"(tiny)
        : %(clobbers)s);
    return stop;
}

"""


SHADOW_FN_FILT = """// The shadow walk over the FILTERED stream (FNodeS[n + 3], rt_skip.hpp: the two-sided bounds of rt_skip_rot.hpp's filtered loops,
// DESIGN.md 4.1), start to END in one invocation: a step forms P2 for both rays with four packed instructions and asks the outer
// bound; a node some awake ray is inside of is settled by the inner bound (sure hit) or the behind-the-origin test (sure miss), and
// only when a ray stays between the bounds does the step fetch the node's Node record from `exact` (the compacted exact stream, same
// offsets) and run the reference's arithmetic for every ray.  fc: the scene's FilterConsts (its l is the rays' direction).
// resume[h] in: 0 for a shadow ray, n_bytes for a lane half without one.  resume[h] out: n_bytes + 1 iff the ray hit something.
__device__ __forceinline__ void %(name)s(const void *nodes, unsigned n_bytes, const float (&ox)[2], const float (&oy)[2],
                                                            const float (&oz)[2], unsigned (&resume)[2], const void *fc, const void *exact)
{
    const float tiny = 0x1p-96f;
    asm volatile(
        "\\t; rt-loops two-ray: undeclared s32, s[72:73]\\n"
%(body)s
        : [res0] "+{v38}"(resume[0]), [res1] "+{v39}"(resume[1])
        : [base] "s"(nodes), [n] "s"(n_bytes), [ox0] "{v32}"(ox[0]), [ox1] "{v33}"(ox[1]), [oy0] "{v34}"(oy[0]), [oy1] "{v35}"(oy[1]),
          [oz0] "{v36}"(oz[0]), [oz1] "{v37}"(oz[1]), [tiny] "s"(tiny), [fc] "s"(fc), [base2] "s"(exact)
        : %(clobbers)s);
}

"""


PRIMARY_BOUND = tuple(range(32, 40)) + tuple(range(52, 56))      # DX, DY, DZ, RES in; BEST, BITEM out  (Regs(False))
SHADOW_BOUND = tuple(range(32, 40))                                # DX, DY, DZ (the origins), RES in / out  (RegsSF)


def clobbers(last=SGPR_LAST, bound=()):
    """bound: vector registers of the loops that ARE operands of the statement (the caller's values arrive in / leave from them directly)."""
    # Not named: what the compiler RESERVES in k_render_skip2 (s32, the stack pointer of a kernel that has no stack; s[72:73] under
    # amdgpu_waves_per_eu(8)) -- it never allocates them, and naming them is what `-Winline-asm` objects to.  That the loops may use them
    # rests on the kernel's .sgpr_count and on no compiler-generated instruction touching them: tests/test_kernel_resources.py checks both
    # on the generated assembly (tools/check_reserved_registers.py).
    regs = ['"s%d"' % r for r in range(SB, last + 1) if r not in RESERVED] + ['"v%d"' % r for r in range(VGPR_FIRST, VGPR_LAST + 1) if r not in bound]
    lines, cur = [], '"memory", "vcc", "scc"'
    for r in regs:
        if len(cur) + len(r) + 2 > 118:
            lines.append(cur + ",")
            cur = "          " + r
        else:
            cur += ", " + r
    lines.append(cur)
    return "\n".join(lines)


def main():
    text = HEADER
    global FUSED
    text += PRIMARY_FN % {"name": "skip2_primary_rot_fused", "body": primary(), "clobbers": clobbers(SGPR_LAST_PRIMARY, PRIMARY_BOUND)}
    text += SHADOW_FN % {"body": shadow(), "clobbers": clobbers()}
    text += SHADOW_FN_FILT % {"name": "skip2_shadow_rot_filt_fused", "body": shadow_filt(), "clobbers": clobbers(SGPR_LAST_FILT, SHADOW_BOUND)}
    # the same two loops over PLAIN filtered streams: scenes whose bounds have no sphere of their own (the automatic hierarchy of an
    # arbitrary sphere list) -- a BOUND step ends where somebody enters
    FUSED = False
    text += "// The plain-stream flavours (a BOUND node carries no sphere of its own): FNode / FNodeS of the scene's plain streams.\n"
    text += PRIMARY_FN % {"name": "skip2_primary_rot", "body": primary(), "clobbers": clobbers(SGPR_LAST_PRIMARY, PRIMARY_BOUND)}
    text += SHADOW_FN_FILT % {"name": "skip2_shadow_rot_filt", "body": shadow_filt(), "clobbers": clobbers(SGPR_LAST_FILT, SHADOW_BOUND)}
    FUSED = True
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
