#!/usr/bin/env python3
"""Generates rust-tracer_amd/csrc/rt_flat_rot.hpp: the inner loops of the scalar-fed flat scan (rt_flat_sc.hpp) in gfx950
assembly, f32: TWO rays per lane on packed math, a conservative FMA filter in front of the reference's exact test.

A linear scan is wave-uniform: all rays of a wave test the same item at the same moment, so the items are scalars.  A group
of FOUR items is one 64-byte record = one s_load_dwordx16 into one of two SGPR banks; the next group's load is issued before the
current group's arithmetic.

What the instructions cost (tools/valu_issue_probe.hip, profiles/r02_valu_issue_probe.json; the time a SIMD needs per wave64
instruction at 8 waves per SIMD, from the probe's LONGEST wave): a VOP2 2.2 cycles, but a VOP2 whose SGPR operand differs
from the previous instruction's 4.1 -- what every VOP3 / VOP3P instruction takes, a packed v_pk_mul/add/fma_f32 included, whose
scalar operand (either half of an aligned SGPR pair, broadcast to both results with op_sel) is free on top.  Every operation
of a sphere test reads an item term no neighbour shares, so each lane carries two rays in VGPR pairs and one packed
instruction does one operation of the test for both; the scalar load, the wait and the branch of a group serve 128 rays.

The filter.  The reference's test is eight (primary) / sixteen (shadow) individually rounded operations per ray and item.
Almost every item is rejected by `disc < 0`, and a rejection does not need disc's bits, only its sign: the loops form a BOUND of
disc with fused multiply-adds -- 4 (primary) / 6 (shadow) packed instructions per item instead of 8 / 16 -- plus a margin that
covers every rounding of both computations, so that disc >= 0 implies bound >= 0.  A group none of whose bounds is >= 0 for any
ray is skipped (three v_max3_f32, one v_max_f32, one compare, one branch per four items and 128 rays); otherwise the items
whose bound is >= 0 for some ray get the reference's exact test, operation for operation, from their exact record (one more
scalar load), in item order, with the exact root path behind it.  Results are the reference's bits; the filter only decides
what is looked at.  Error analysis (eps = 2^-24, |dir| <= 1 + 2 eps, v the stored f32 centre - eye or the f32 centre - origin
both computations start from; B = the exact dot product of v and dir):
    exact     b  = fl(fl(fl(vx dx) + fl(vy dy)) + fl(vz dz))        |b  - B| <= 3.1 eps |v|
    filter    b' = fma(vz, dz, fma(vy, dy, fl(vx dx)))              |b' - B| <= 3.1 eps |v|     so |b b - b' b'| <= 12.4 eps |v|^2
    exact     disc = fl(fl(fl(b b) - vv) + rr) >= 0  implies  b b - vv + rr >= -(1.1 eps |v|^2 + 2 eps rr)
    primary   vv is the stored f32 dot(v, v): within 4 eps of |v|^2.  Hence disc >= 0 implies b' b' - vv + rr >= -(14 eps vv + 3 eps rr),
              and bound = fma(b', b', K) >= 0 for K = rr - vv + 2^-17 (vv + rr) + 2^-140 rounded UP (2^-17 = 128 eps: a factor of eight
              in hand; 2^-140 covers subnormal results, whose errors are absolute, <= 2^-149 per operation)
    shadow    the origin o is per ray, so nothing of vv can be pre-formed from v = c - o.  Relative to a point m0 inside the scene,
              c' = fl(c - m0), o' = fl(o - m0):  |c' - o'|^2 = |c'|^2 + |o'|^2 - 2 c'.o'  and  b = c'.l - o'.l, so per item only
                  u = fma(cz', 2oz', fma(cy', 2oy', fma(cx', 2ox', -Pm))) ; b' = CL - OL ; bound = fma(b', b', u) + G
              remain, with CL = c'.l (from double, rounded once) and G = rr - |c'|^2 + m (|c'|^2 + rr) + 2^-140 (rounded UP) per item,
              OL = o'.l and Pm = |o'|^2 (1 - m) (rounded towards zero) per ray, m = 2^-16.  With S = |c'|^2 + |o'|^2: the exact b and b'
              are within 5.3 eps and 4.5 eps (|c'| + |o'|) of c'.l - o'.l (re-centring costs eps (|c'| + |o'|)), so b b and b' b' differ
              by <= 39.6 eps S; the exact vv and |c' - o'|^2 by <= 14.4 eps S; the exact test's roundings add 2.2 eps S + 2 eps rr, the
              filter's own 21 eps S + eps rr.  disc >= 0 implies bound >= m (S + rr) - (77.2 eps S + 3 eps rr) >= 0: 256 eps against 77.
              S is taken about a point inside the scene, so the bound stays tight for scenes far from the coordinate origin
rt_debug_flat_filter_check evaluates both sides for every ray x item pair of a frame (tests/test_gpu_parity.py: 4.5e10 pairs of
the default scene, scenes scaled from 1e-20 to 5e13 and a scene 6,000 units from the origin: no pair with disc >= 0 and bound < 0;
the bounds let through 1.33x (primary) / 1.13x (shadow) the exact candidates).  A shadow ray that is settled (or a lane half without a ray) gets a NaN origin: every bound and
discriminant it forms from then on is NaN, which is never `>= 0`, so it needs no mask of its own.

The halves of a register pair have to be named, which inline-asm operands cannot do, so the loops own FIXED registers (v[32:63],
s[36:87]; listed as clobbers -- the kernels around them need 20 VGPRs) and move their operands in and out.

Exact arithmetic, operation for operation (primitive.rs:55-72; each + - * rounded once, no FMA outside the exact root; a packed
subtraction is an addition with the IEEE sign flip of the neg modifier):
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
LIGHT = 74                           # s74..s76: light direction of the shadow scan (s77 pads the pair)
EXACT = 80                           # s[80:87]: the exact record of one item (slow path)
STRIDE = 64
SGPR_LAST = 87
VGPR_FIRST, VGPR_LAST = 32, 63


class Pair:
    """A VGPR pair: .p the pair, .h[0] / .h[1] its halves (ray 0 / ray 1)."""

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

    def render(self, indent="        "):
        out = []
        for tab, text, comment in self.lines:
            s = '%s"%s%s\\n"' % (indent, "" if tab == "" else "\\t", text)
            if comment:
                s += "  /* %s */" % comment
            out.append(s)
        return "\n".join(out)


def sreg(bank, field, k):
    """SGPR number of item k's field in a bank: fields are stored [field][item]."""
    return BANK[bank] + 3 * field + k


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


def pk_fma(a, dst, x, y, z, sx=None, sz=None, comment=None):
    """dst = x * y + z on both rays, ONE rounding (the conservative filter only; never on the exact path)."""
    sel, sel_hi = [0, 0, 0], [1, 1, 1]
    if sx is not None:
        x = "s[%d:%d]" % (sx & ~1, (sx & ~1) + 1)
        sel[0] = sel_hi[0] = sx & 1
    if sz is not None:
        z = "s[%d:%d]" % (sz & ~1, (sz & ~1) + 1)
        sel[2] = sel_hi[2] = sz & 1
    mods = ""
    if sel != [0, 0, 0]:
        mods += " op_sel:[%d,%d,%d]" % tuple(sel)
    if sel_hi != [1, 1, 1]:
        mods += " op_sel_hi:[%d,%d,%d]" % tuple(sel_hi)
    a.op("v_pk_fma_f32 %s, %s, %s, %s%s" % (dst, x, y, z, mods), comment)


def load(a, bank, off, comment=None):
    a.op("s_load_dwordx16 s[%d:%d], %%[base], %s" % (BANK[bank], BANK[bank] + 15, off), comment)


def refine(a, x, t0, t1, root):
    a.op("v_mul_f32_e32 %s, %s, %s" % (root, x, t0), "g = x*y")
    a.op("v_mul_f32_e32 %s, 0.5, %s" % (t0, t0), "h = y/2")
    a.op("v_fma_f32 %s, -%s, %s, %s" % (t1, root, root, x), "r = x - g*g")
    a.op("v_fma_f32 %s, %s, %s, %s" % (root, t1, t0, root), "g + r*h")


def exact_root(a, disc, tag, r):
    """Correctly rounded sqrt(disc) into r.root for the lanes in EXEC (== sqrt_rn_lean)."""
    a.op("v_rsq_f32_e32 %s, %s" % (r.t0, disc))
    a.op("v_cmp_lt_f32_e64 %s, |%s|, %%[tiny]" % (TINY, disc))
    a.op("s_cmp_lg_u64 %s, 0" % TINY)
    a.op("s_cbranch_scc1 .Lfl_tiny_%s_%%=" % tag, "a lane below 2^-96 (zero included): scaled path")
    refine(a, disc, r.t0, r.t1, r.root)
    a.label(".Lfl_rooted_%s_%%=" % tag)


def exact_tiny(a, disc, tag, r):
    a.label(".Lfl_tiny_%s_%%=" % tag)
    a.op("v_mul_f32_e32 %s, 0x5f800000, %s" % (r.t0, disc), "root with the 2^64 / 2^-32 scaling for tiny lanes (the scaled operand stays >= 2^-85: its residual is never subnormal)")
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.t2, disc, r.t0, TINY))
    a.op("v_rsq_f32_e32 %s, %s" % (r.t0, r.t2))
    a.op("v_cmp_eq_f32_e32 vcc, 0, %s" % r.t2, "sqrt(+-0) = +-0 (rsq would make it 0 * inf)")
    refine(a, r.t2, r.t0, r.t1, r.root)
    a.op("v_cndmask_b32_e32 %s, %s, %s, vcc" % (r.root, r.root, r.t2))
    a.op("v_mul_f32_e32 %s, 0x2f800000, %s" % (r.t0, r.root))
    a.op("v_cndmask_b32_e64 %s, %s, %s, %s" % (r.root, r.root, r.t0, TINY))
    a.op("s_branch .Lfl_rooted_%s_%%=" % tag)


def any_candidate(a, r, bank):
    """One branch for the group: does any ray of the wave have disc >= 0 for any of the three items?"""
    m = r.T0.h[0]
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, r.D[0].h[0], r.D[1].h[0], r.D[2].h[0]))
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, m, r.D[0].h[1], r.D[1].h[1]))
    a.op("v_max_f32_e32 %s, %s, %s" % (m, m, r.D[2].h[1]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % m)
    a.op("s_cbranch_vccnz .Lfl_slow_%s_%%=" % bank, "some ray's line meets one of the three spheres")
    a.label(".Lfl_cont_%s_%%=" % bank)


# ------------------------------------------------------------------------------------------------------------------ primary

class PrimaryRegs:
    def __init__(self):
        v = VGPR_FIRST
        self.DX, self.DY, self.DZ = Pair(v), Pair(v + 2), Pair(v + 4)
        self.BEST, self.BITEM = Pair(v + 6), Pair(v + 8)
        self.T0, self.T1, self.T2 = Pair(v + 10), Pair(v + 12), Pair(v + 14)
        self.F = [Pair(v + 16), Pair(v + 18), Pair(v + 20), Pair(v + 22)]       # the filter's discriminant bounds of a group's 4 items
        self.B, self.DISC = Pair(v + 24), Pair(v + 26)
        self.root = "v%d" % (v + 28)
        self.t0, self.t1, self.t2 = self.T0.h[0], self.T0.h[1], self.T1.h[0]


def fsreg(bank, field, k):
    """SGPR number of item k's filter field (0..3: vx, vy, vz, K) in a bank of FOUR items: fields are stored [field][item]."""
    return BANK[bank] + 4 * field + k


def primary_group(a, bank, r):
    """The conservative filter for four items: bound = fma(b', b', K) with b' = fma(vz, dz, fma(vy, dy, vx*dx)) -- four packed
    FMAs per item for both rays.  K = rr - vv + a margin that covers every rounding of both computations (tools/gen_flat_asm.py
    docstring), so disc >= 0 implies bound >= 0: a group with no bound >= 0 holds no candidate."""
    for k in range(4):
        pk(a, "mul", r.T0.p, None, r.DX.p, sx=fsreg(bank, 0, k), comment="item %d: bound of disc, both rays" % k if k == 0 else None)
        pk_fma(a, r.T0.p, None, r.DY.p, r.T0.p, sx=fsreg(bank, 1, k))
        pk_fma(a, r.T0.p, None, r.DZ.p, r.T0.p, sx=fsreg(bank, 2, k))
        pk_fma(a, r.F[k].p, r.T0.p, r.T0.p, None, sz=fsreg(bank, 3, k))
    m = r.T0.h[0]
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, r.F[0].h[0], r.F[0].h[1], r.F[1].h[0]))
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, m, r.F[1].h[1], r.F[2].h[0]))
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, m, r.F[2].h[1], r.F[3].h[0]))
    a.op("v_max_f32_e32 %s, %s, %s" % (m, m, r.F[3].h[1]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % m)
    a.op("s_cbranch_vccnz .Lfl_slow_%s_%%=" % bank, "some ray's line may meet one of the four spheres")
    a.label(".Lfl_cont_%s_%%=" % bank)


def primary_slow(a, bank, r):
    """The exact test (primitive.rs:55-84, operation for operation) of the items whose bound is >= 0 for some ray, in item order."""
    a.label(".Lfl_slow_%s_%%=" % bank)
    a.op("s_lshr_b32 %s, %s, 4" % (IDX, OFF), "index of the group's first item: 4 * (offset / 64)%s" % (" + 4" if bank == "B" else ""))
    if bank == "B":
        a.op("s_add_u32 %s, %s, 4" % (IDX, IDX))
    for k in range(4):
        nxt = ".Lfl_item_%s%d_%%=" % (bank, k)
        a.op("v_max_f32_e32 %s, %s, %s" % (r.t0, r.F[k].h[0], r.F[k].h[1]), "item %d" % k)
        a.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.t0)
        a.op("s_cbranch_vccz %s" % nxt)
        a.op("s_lshl_b32 s76, %s, 5" % IDX)
        a.op("s_load_dwordx8 s[%d:%d], %%[exact], s76" % (EXACT, EXACT + 7), "{vx, vy, vz, vv, rr} of the item")
        a.op("s_waitcnt lgkmcnt(0)")
        pk(a, "mul", r.T0.p, None, r.DX.p, sx=EXACT + 0, comment="b = (vx*dx + vy*dy) + vz*dz, both rays")
        pk(a, "mul", r.T1.p, None, r.DY.p, sx=EXACT + 1)
        pk(a, "mul", r.T2.p, None, r.DZ.p, sx=EXACT + 2)
        pk(a, "add", r.T0.p, r.T0.p, r.T1.p)
        pk(a, "add", r.B.p, r.T0.p, r.T2.p)
        pk(a, "mul", r.T0.p, r.B.p, r.B.p, comment="disc = (b*b - vv) + rr")
        pk(a, "add", r.T0.p, r.T0.p, None, sy=EXACT + 3, neg_y=True)
        pk(a, "add", r.DISC.p, None, r.T0.p, sx=EXACT + 4)
        for h in range(2):
            tag = "%s%d%d" % (bank, k, h)
            b, d, best, bitem = r.B.h[h], r.DISC.h[h], r.BEST.h[h], r.BITEM.h[h]
            a.op("v_cmp_le_f32_e32 vcc, 0, %s" % d, "ray %d (primitive.rs:79: the first item keeps a tie)" % h)
            a.op("s_and_saveexec_b64 %s, vcc" % EXS)
            a.op("s_cbranch_execz .Lfl_next_%s_%%=" % tag)
            exact_root(a, d, tag, r)
            a.op("v_add_f32_e32 %s, %s, %s" % (r.t0, b, r.root), "t2")
            a.op("v_sub_f32_e32 %s, %s, %s" % (r.t1, b, r.root), "t1")
            a.op("v_cmp_lt_f32_e32 vcc, 0, %s" % r.t1)
            a.op("v_cndmask_b32_e32 %s, %s, %s, vcc" % (r.t1, r.t0, r.t1), "d = t1 > 0 ? t1 : t2")
            a.op("v_cmpx_le_f32_e32 0, %s" % r.t0, "t2 >= 0")
            a.op("v_cmpx_lt_f32_e32 %s, %s" % (r.t1, best), "d < hit.distance")
            a.op("v_mov_b32_e32 %s, %s" % (best, r.t1), "primitive.rs:80-83")
            a.op("v_mov_b32_e32 %s, %s" % (bitem, IDX))
            a.label(".Lfl_next_%s_%%=" % tag)
            a.op("s_mov_b64 exec, %s" % EXS)
        a.label(nxt)
        if k < 3:
            a.op("s_add_u32 %s, %s, 1" % (IDX, IDX))
    a.op("s_branch .Lfl_cont_%s_%%=" % bank)
    for k in range(4):
        for h in range(2):
            exact_tiny(a, r.DISC.h[h], "%s%d%d" % (bank, k, h), r)


# ------------------------------------------------------------------------------------------------------------------- shadow

class ShadowRegs:
    def __init__(self):
        v = VGPR_FIRST
        self.OX, self.OY, self.OZ = Pair(v), Pair(v + 2), Pair(v + 4)            # origins (exact test); NaN once a ray is settled
        self.O2X, self.O2Y, self.O2Z = Pair(v + 6), Pair(v + 8), Pair(v + 10)    # filter: 2 * (origin - scene centre)
        self.OL, self.NPM = Pair(v + 12), Pair(v + 14)                           # filter: dir . (origin - centre); -|origin - centre|^2 (1 - m)
        self.T0, self.T1 = Pair(v + 16), Pair(v + 18)
        self.F = [Pair(v + 20), Pair(v + 22), Pair(v + 24)]                      # the filter's discriminant bounds of a group's 3 items
        self.VX, self.VY, self.VZ = Pair(v + 26), Pair(v + 28), Pair(v + 30)
        self.B, self.DISC = Pair(v + 32), Pair(v + 34)
        self.OCC = Pair(v + 36)
        self.root = "v%d" % (v + 38)
        self.t0, self.t1, self.t2 = self.T0.h[0], self.T0.h[1], self.T1.h[0]


SHADOW_VGPR_LAST = VGPR_FIRST + 38


def shadow_group(a, bank, r):
    """The conservative filter for three items, six packed instructions per item for both rays.  With c' = centre - m0 and
    o' = origin - m0 (m0: a point inside the scene), |c - o|^2 = |c'|^2 + |o'|^2 - 2 c'.o' and b = c'.l - o'.l, so
        bound = (b'^2 + u) + G,   b' = CL - OL,   u = fma(cz', 2oz', fma(cy', 2oy', fma(cx', 2ox', -Pm)))
    with CL = c'.l and G = rr - |c'|^2 + m (|c'|^2 + rr) + 2^-140 (rounded up) per item, OL = o'.l and Pm = |o'|^2 (1 - m)
    (rounded down) per ray, m = 2^-16: disc >= 0 implies bound >= 0 (tools/gen_flat_asm.py docstring)."""
    for k in range(3):
        pk_fma(a, r.T0.p, None, r.O2X.p, r.NPM.p, sx=sreg(bank, 0, k), comment="item %d: bound of disc, both rays" % k if k == 0 else None)
        pk_fma(a, r.T0.p, None, r.O2Y.p, r.T0.p, sx=sreg(bank, 1, k))
        pk_fma(a, r.T0.p, None, r.O2Z.p, r.T0.p, sx=sreg(bank, 2, k))
        pk(a, "add", r.T1.p, None, r.OL.p, sx=sreg(bank, 3, k), neg_y=True)
        pk_fma(a, r.T0.p, r.T1.p, r.T1.p, r.T0.p)
        pk(a, "add", r.F[k].p, None, r.T0.p, sx=sreg(bank, 4, k))
    m = r.T0.h[0]
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, r.F[0].h[0], r.F[0].h[1], r.F[1].h[0]))
    a.op("v_max3_f32 %s, %s, %s, %s" % (m, m, r.F[1].h[1], r.F[2].h[0]))
    a.op("v_max_f32_e32 %s, %s, %s" % (m, m, r.F[2].h[1]))
    a.op("v_cmp_le_f32_e32 vcc, 0, %s" % m)
    a.op("s_cbranch_vccnz .Lfl_slow_%s_%%=" % bank, "some pending ray's line may meet one of the three spheres")
    a.label(".Lfl_cont_%s_%%=" % bank)


def shadow_slow(a, bank, r):
    """The exact test of the items whose bound is >= 0 for some ray.  A ray that hits is settled: flagged, and its origin becomes
    NaN.  A lane whose two rays are both settled leaves EXEC; the wave leaves the scan when EXEC is empty."""
    a.label(".Lfl_slow_%s_%%=" % bank)
    a.op("s_lshr_b32 %s, %s, 6" % (IDX, OFF), "index of the group's first item: 3 * (offset / 64)%s" % (" + 3" if bank == "B" else ""))
    a.op("s_mul_i32 %s, %s, 3" % (IDX, IDX))
    if bank == "B":
        a.op("s_add_u32 %s, %s, 3" % (IDX, IDX))
    for k in range(3):
        nxt = ".Lfl_item_%s%d_%%=" % (bank, k)
        a.op("v_max_f32_e32 %s, %s, %s" % (r.t0, r.F[k].h[0], r.F[k].h[1]), "item %d" % k)
        a.op("v_cmp_le_f32_e32 vcc, 0, %s" % r.t0)
        a.op("s_cbranch_vccz %s" % nxt)
        a.op("s_lshl_b32 s78, %s, 4" % IDX)
        a.op("s_load_dwordx4 s[%d:%d], %%[exact], s78" % (EXACT, EXACT + 3), "{cx, cy, cz, rr} of the item")
        a.op("s_waitcnt lgkmcnt(0)")
        pk(a, "add", r.VX.p, None, r.OX.p, sx=EXACT + 0, neg_y=True, comment="v = centre - origin, both rays")
        pk(a, "add", r.VY.p, None, r.OY.p, sx=EXACT + 1, neg_y=True)
        pk(a, "add", r.VZ.p, None, r.OZ.p, sx=EXACT + 2, neg_y=True)
        pk(a, "mul", r.T0.p, None, r.VX.p, sx=LIGHT + 0)
        pk(a, "mul", r.T1.p, None, r.VY.p, sx=LIGHT + 1)
        pk(a, "add", r.T0.p, r.T0.p, r.T1.p)
        pk(a, "mul", r.T1.p, None, r.VZ.p, sx=LIGHT + 2)
        pk(a, "add", r.B.p, r.T0.p, r.T1.p, comment="b = dot(v, dir)")
        pk(a, "mul", r.VX.p, r.VX.p, r.VX.p)
        pk(a, "mul", r.VY.p, r.VY.p, r.VY.p)
        pk(a, "add", r.VX.p, r.VX.p, r.VY.p)
        pk(a, "mul", r.VZ.p, r.VZ.p, r.VZ.p)
        pk(a, "add", r.VX.p, r.VX.p, r.VZ.p, comment="dot(v, v)")
        pk(a, "mul", r.T0.p, r.B.p, r.B.p)
        pk(a, "add", r.T0.p, r.T0.p, r.VX.p, neg_y=True)
        pk(a, "add", r.DISC.p, None, r.T0.p, sx=EXACT + 3, comment="disc = (b*b - vv) + rr")
        for h in range(2):
            tag = "%s%d%d" % (bank, k, h)
            b, d = r.B.h[h], r.DISC.h[h]
            a.op("v_cmp_le_f32_e32 vcc, 0, %s" % d, "ray %d: disc >= 0 (NaN for a settled ray)" % h)
            a.op("s_and_saveexec_b64 %s, vcc" % EXS)
            a.op("s_cbranch_execz .Lfl_next_%s_%%=" % tag)
            exact_root(a, d, tag, r)
            a.op("v_add_f32_e32 %s, %s, %s" % (r.t0, b, r.root), "t2")
            a.op("v_cmpx_le_f32_e32 0, %s" % r.t0, "t2 >= 0: the ray is occluded (render.rs:208 only asks has_missed())")
            a.op("v_mov_b32_e32 %s, 1" % r.OCC.h[h])
            a.op("v_mov_b32_e32 %s, 0x7fc00000" % r.OX.h[h], "settled: NaN discriminants and NaN bounds from now on")
            a.op("v_mov_b32_e32 %s, 0x7fc00000" % r.NPM.h[h])
            a.label(".Lfl_next_%s_%%=" % tag)
            a.op("s_mov_b64 exec, %s" % EXS)
        a.label(nxt)
        if k < 2:
            a.op("s_add_u32 %s, %s, 1" % (IDX, IDX))
    a.op("v_cmp_o_f32_e32 vcc, %s, %s" % (r.OX.h[0], r.OX.h[0]), "lanes that still carry an unsettled ray")
    a.op("v_cmp_o_f32_e64 %s, %s, %s" % (TINY, r.OX.h[1], r.OX.h[1]))
    a.op("s_or_b64 vcc, vcc, %s" % TINY)
    a.op("s_and_b64 exec, exec, vcc")
    a.op("s_cbranch_execz .Lfl_exit_%=", "every ray of the wave is settled")
    a.op("s_branch .Lfl_cont_%s_%%=" % bank)
    for k in range(3):
        for h in range(2):
            exact_tiny(a, r.DISC.h[h], "%s%d%d" % (bank, k, h), r)


def loop(a, group, slow, r):
    """Two groups per iteration, double-buffered: bank A holds the current group on entry."""
    a.label(".Lfl_loop_%=")
    a.op("s_add_u32 %s, %s, %d" % (IDX, OFF, STRIDE))
    load(a, "B", IDX, "the next group, while this one is tested")
    group(a, "A", r)
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_add_u32 %s, %s, %d" % (IDX, OFF, 2 * STRIDE))
    load(a, "A", IDX)
    group(a, "B", r)
    a.op("s_add_u32 %s, %s, %d" % (OFF, OFF, 2 * STRIDE))
    a.op("s_cmp_lt_u32 %s, %%[end]" % OFF)
    a.op("s_waitcnt lgkmcnt(0)")
    a.op("s_cbranch_scc1 .Lfl_loop_%=")
    a.op("s_branch .Lfl_exit_%=")
    slow(a, "A", r)
    slow(a, "B", r)


HEADER = """// rt_flat_rot.hpp -- GENERATED by tools/gen_flat_asm.py; edit the generator, not this file.
//
// The inner loops of the scalar-fed flat scan (rt_flat_sc.hpp) in gfx950 assembly, f32, two rays per lane on packed math.  A
// group of four (primary) or three (shadow) items is one 64-byte record = one s_load_dwordx16 into one of two SGPR banks (s[36:51], s[52:67]); the next
// group's load is issued before the current group's arithmetic.  Per item the loops form a conservative BOUND of the
// discriminant with packed FMAs (4 instructions primary, 6 shadow, for the lane's two rays; item terms are the low or high
// half of an aligned SGPR pair broadcast with op_sel), whose margin covers every rounding of both computations: disc >= 0 implies
// bound >= 0 (tools/gen_flat_asm.py has the analysis, rt_debug_flat_filter_check the exhaustive check).  Three v_max3_f32 + one
// v_max_f32 and one branch reject a group; otherwise the items whose bound is >= 0 for some ray get the reference's exact test,
// operation for operation, from their exact record, in item order (root == sqrt_rn_lean, t2, t1, d, strict `<`), per ray on the
// 32-bit halves.  The loops own v[32:63] (shadow: v[32:70]) and s[36:87] (clobbers): s68 byte offset of the current group pair, s69 item index,
// s[70:71] saved EXEC, s[72:73] mask scratch, s[74:76] the shadow rays' direction, s76 / s78 address scratch, s[80:87] the exact
// record.
//
// Filter group (rt_flat_sc.hpp, FGroup): primary {vx[4], vy[4], vz[4], K[4]}; shadow {cx'[3], cy'[3], cz'[3], CL[3], G[3], -}.  Exact record:
// primary FExact {vx, vy, vz, vv, rr, -, -, -}; shadow FExactShadow {cx, cy, cz, rr}.  The group arrays end in pad groups (K, rr' =
// -inf: never a candidate) so that the load issued one pair ahead stays inside them.
#pragma once
#include "rt_kernels.hpp"

namespace rt {

"""

PRIMARY = """// Nearest hit of all items for the wave's 128 primary rays: `groups` = the filter groups [0, n_bytes / 64) (n_bytes: a multiple
// of 128), `exact` = the items' exact records (32 bytes each); ray h of a lane has the direction (dx[h], dy[h], dz[h]).  Lane halves
// without a ray scan along (their result is ignored).  Returns hit.distance and the index of the winning item per ray.
__device__ __forceinline__ void flat_primary_scan(const void *groups, unsigned n_bytes, const void *exact, const float (&dx)[2],
                                                  const float (&dy)[2], const float (&dz)[2], float (&best_out)[2], unsigned (&item_out)[2])
{
    const float tiny = 0x1p-96f;
    asm volatile(
%(body)s
        : [best0] "=v"(best_out[0]), [best1] "=v"(best_out[1]), [item0] "=v"(item_out[0]), [item1] "=v"(item_out[1])
        : [base] "s"(groups), [end] "s"(n_bytes), [exact] "s"(exact), [dx0] "v"(dx[0]), [dx1] "v"(dx[1]), [dy0] "v"(dy[0]), [dy1] "v"(dy[1]),
          [dz0] "v"(dz[0]), [dz1] "v"(dz[1]), [tiny] "s"(tiny)
        : %(clobbers)s);
}

"""

SHADOW = """// Any hit over the filter groups [begin_bytes / 64, end_bytes / 64) (multiples of 128; `exact`: the items' exact records, 16 bytes
// each) for the wave's 128 shadow rays; ray h of a lane starts at (ox[h], oy[h], oz[h]) and counts only if pending[h] != 0.  For the
// filter the caller passes, per ray, 2 (origin - scene centre), ol = dir . (origin - centre) and npm = -|origin - centre|^2 (1 - 2^-16)
// rounded towards zero.  Returns 1 in occluded[h] for an occluded ray.  The wave leaves as soon as every pending ray is settled.
__device__ __forceinline__ void flat_shadow_scan(const void *groups, unsigned begin_bytes, unsigned end_bytes, const void *exact, const float (&ox)[2],
                                                 const float (&oy)[2], const float (&oz)[2], const float (&o2x)[2], const float (&o2y)[2],
                                                 const float (&o2z)[2], const float (&ol)[2], const float (&npm)[2], float lx, float ly, float lz,
                                                 const unsigned (&pending)[2], unsigned (&occluded)[2])
{
    const float tiny = 0x1p-96f;
    unsigned long long saved;
    asm volatile(
%(body)s
        : [occ0] "=v"(occluded[0]), [occ1] "=v"(occluded[1]), [saved] "=&s"(saved)
        : [base] "s"(groups), [begin] "s"(begin_bytes), [end] "s"(end_bytes), [exact] "s"(exact), [ox0] "v"(ox[0]), [ox1] "v"(ox[1]), [oy0] "v"(oy[0]),
          [oy1] "v"(oy[1]), [oz0] "v"(oz[0]), [oz1] "v"(oz[1]), [o2x0] "v"(o2x[0]), [o2x1] "v"(o2x[1]), [o2y0] "v"(o2y[0]), [o2y1] "v"(o2y[1]),
          [o2z0] "v"(o2z[0]), [o2z1] "v"(o2z[1]), [ol0] "v"(ol[0]), [ol1] "v"(ol[1]), [npm0] "v"(npm[0]), [npm1] "v"(npm[1]), [lx] "s"(lx),
          [ly] "s"(ly), [lz] "s"(lz), [pend0] "v"(pending[0]), [pend1] "v"(pending[1]), [tiny] "s"(tiny)
        : %(clobbers)s);
}

"""


def clobbers(vgpr_last=VGPR_LAST):
    regs = ['"s%d"' % r for r in range(36, SGPR_LAST + 1)] + ['"v%d"' % r for r in range(VGPR_FIRST, vgpr_last + 1)]
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
    a, r = Asm(), PrimaryRegs()
    for h in range(2):
        a.op("v_mov_b32_e32 %s, %%[dx%d]" % (r.DX.h[h], h), "operands into the loop's own registers" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, %%[dy%d]" % (r.DY.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[dz%d]" % (r.DZ.h[h], h))
        a.op("v_mov_b32_e32 %s, 0x7f800000" % r.BEST.h[h], "hit.distance = INF (primitive.rs:96)" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, 0" % r.BITEM.h[h])
    a.op("s_mov_b32 %s, 0" % OFF)
    load(a, "A", "0x0")
    a.op("s_waitcnt lgkmcnt(0)")
    loop(a, primary_group, primary_slow, r)
    a.label(".Lfl_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)", "the load issued one pair ahead must have landed before its registers are free again")
    for h in range(2):
        a.op("v_mov_b32_e32 %%[best%d], %s" % (h, r.BEST.h[h]))
        a.op("v_mov_b32_e32 %%[item%d], %s" % (h, r.BITEM.h[h]))
    return a.render()


def shadow_body():
    a, r = Asm(), ShadowRegs()
    a.op("s_mov_b64 %[saved], exec")
    for h in range(2):
        a.op("v_cmp_ne_u32_e32 vcc, 0, %%[pend%d]" % h, "a lane half without a pending ray: NaN origin, never a candidate" if h == 0 else None)
        a.op("v_mov_b32_e32 %s, 0x7fc00000" % r.t0)
        a.op("v_cndmask_b32_e32 %s, %s, %%[ox%d], vcc" % (r.OX.h[h], r.t0, h))
        a.op("v_cndmask_b32_e32 %s, %s, %%[npm%d], vcc" % (r.NPM.h[h], r.t0, h))
        a.op("v_mov_b32_e32 %s, %%[oy%d]" % (r.OY.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[oz%d]" % (r.OZ.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[o2x%d]" % (r.O2X.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[o2y%d]" % (r.O2Y.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[o2z%d]" % (r.O2Z.h[h], h))
        a.op("v_mov_b32_e32 %s, %%[ol%d]" % (r.OL.h[h], h))
        a.op("v_mov_b32_e32 %s, 0" % r.OCC.h[h])
    a.op("s_mov_b32 s%d, %%[lx]" % (LIGHT + 0))
    a.op("s_mov_b32 s%d, %%[ly]" % (LIGHT + 1))
    a.op("s_mov_b32 s%d, %%[lz]" % (LIGHT + 2))
    a.op("v_cmp_o_f32_e32 vcc, %s, %s" % (r.OX.h[0], r.OX.h[0]))
    a.op("v_cmp_o_f32_e64 %s, %s, %s" % (TINY, r.OX.h[1], r.OX.h[1]))
    a.op("s_or_b64 vcc, vcc, %s" % TINY)
    a.op("s_and_b64 exec, exec, vcc", "EXEC = the lanes that carry a pending ray")
    a.op("s_cbranch_execz .Lfl_done_%=")
    a.op("s_mov_b32 %s, %%[begin]" % OFF)
    load(a, "A", OFF)
    a.op("s_waitcnt lgkmcnt(0)")
    loop(a, shadow_group, shadow_slow, r)
    a.label(".Lfl_exit_%=")
    a.op("s_waitcnt lgkmcnt(0)")
    a.label(".Lfl_done_%=")
    a.op("s_mov_b64 exec, %[saved]")
    for h in range(2):
        a.op("v_mov_b32_e32 %%[occ%d], %s" % (h, r.OCC.h[h]))
    return a.render()


def main():
    text = HEADER
    text += PRIMARY % {"body": primary(), "clobbers": clobbers()}
    text += SHADOW % {"body": shadow_body(), "clobbers": clobbers(SHADOW_VGPR_LAST)}
    text += "}  // namespace rt\n"
    with open(OUT, "w") as f:
        f.write(text)
    print("wrote", OUT, "(%d lines)" % text.count("\n"))


if __name__ == "__main__":
    main()
