#!/usr/bin/env python3
"""Checks, on the compiler's own assembly output (csrc/rt_capi.gfx950.s, `make -C rust-tracer_amd/csrc asm`), what the generated loops that use
scalar registers WITHOUT declaring them rest on.

tools/gen_skip2_asm.py (the two-ray loops: s32, s[72:73]) and tools/gen_skip_asm.py (the filtered f64 loops: s[88:89]; their low-window copies: s32) leave out of their
clobber lists the registers the compiler reserves in the one kernel they are built into -- it never allocates a reserved register, and
naming one is what `-Winline-asm` ("clobber list contains reserved registers") objects to.  Every such statement starts with a comment
`; rt-loops <flavour>: undeclared ...`, which survives into the .s.  For every function of the .s that holds such a statement:
  * it is the kernel the flavour was written for (k_render_skip2 / k_render_skip_f64, its cooperative flavour, k_render_skip_fast64_coop) -- in any other kernel those registers could hold
    the compiler's values;
  * no instruction OUTSIDE the inline-assembly blocks (;;#ASMSTART .. ;;#ASMEND) reads or writes one of those registers;
  * the wave's allocation covers them: 8 * (SGPRBlocks + 1) >= highest undeclared register + 1 + the six the hardware keeps at the end
    of the allocation (VCC, FLAT_SCRATCH, XNACK_MASK).
Prints one line per kernel; exit status 1 on a violation.   usage: check_reserved_registers.py [file.s]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_S = os.path.join(ROOT, "rust-tracer_amd", "csrc", "rt_capi.gfx950.s")
FLAVOURS = {"two-ray": (("void rt::k_render_skip2<", "void rt::k_render_skip2_fast<"), (32, 72, 73)),
            "f64": (("void rt::k_render_skip_f64<", "void rt::k_render_skip_f64_coop<"), (88, 89)),
            "f64-lo": (("void rt::k_render_skip_fast64_coop<",), (32,))}


def sgprs_of(text):
    """scalar registers an instruction line names (s5, s[4:7]); comments stripped"""
    text = text.split(";", 1)[0]
    regs = set()
    for lo, hi in re.findall(r"\bs\[(\d+):(\d+)\]", text):
        regs.update(range(int(lo), int(hi) + 1))
    regs.update(int(r) for r in re.findall(r"\bs(\d+)\b", text))
    return regs


def check(path=DEFAULT_S):
    lines = open(path).read().split("\n")
    problems, report = [], []
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\w+):\s*; @", l)] if m]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True, check=True).stdout.split("\n")
    for (first, mangled), name in zip(starts, names):
        end = next(i for i in range(first, len(lines)) if lines[i].startswith(".Lfunc_end"))
        in_app, flavours, outside = False, set(), []
        for l in lines[first + 1:end]:
            t = l.strip()
            if t.startswith((";;#ASMSTART", ";APP")):
                in_app = True
            elif t.startswith((";;#ASMEND", ";NO_APP")):
                in_app = False
            elif in_app:
                m = re.search(r"; rt-loops ([\w-]+): undeclared", t)
                if m:
                    flavours.add(m.group(1))
            elif t and not t.startswith((";", ".")) and not t.endswith(":"):
                outside.append(t)
        if not flavours:
            continue
        blocks = None
        for l in lines[end:end + 80]:
            m = re.match(r"; SGPRBlocks: (\d+)", l)
            if m:
                blocks = int(m.group(1))
                break
        for fl in sorted(flavours):
            prefixes, regs = FLAVOURS[fl]
            if not name.startswith(prefixes):
                problems.append("%s holds %s loops, which leave %s undeclared" % (name, fl, regs))
            touched = [t for t in outside if sgprs_of(t) & set(regs)]
            if touched:
                problems.append("%s: compiler-generated code touches %s: %s" % (name, regs, touched[:3]))
            if blocks is None or 8 * (blocks + 1) < max(regs) + 1 + 6:
                problems.append("%s: SGPRBlocks %s does not cover s%d + the hardware's six" % (name, blocks, max(regs)))
            report.append("%-60s %s loops, s%s untouched in %d compiler-generated instructions, %d scalar registers allocated"
                          % (name.split("(")[0][:60], fl, "/s".join(str(r) for r in regs), len(outside), 8 * (blocks + 1) if blocks is not None else -1))
    return report, problems


if __name__ == "__main__":
    rep, bad = check(sys.argv[1] if len(sys.argv) > 1 else DEFAULT_S)
    print("\n".join(rep))
    for b in bad:
        print("VIOLATION:", b)
    sys.exit(1 if bad or not rep else 0)
