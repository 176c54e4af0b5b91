#!/usr/bin/env python3
"""profiles/<tag>_sq.json from tools/profile_sq.sh: per-launch averages of the SQ counters of the render kernels plus
derived figures (kernel duration from the counter rows' timestamps).  usage: summarize_sq.py <gpurun_out> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_src_sha, git_head  # noqa: E402  (the stamp bench.py compares with the sources it runs on)
KERNELS = [("k_render_skip_fast", "k_render_skip"), ("k_render_skip_f32<false", "k_render_skip"), ("k_render_skip_f32_coop<false", "k_render_skip"), ("k_render_skip<float, false", "k_render_skip"), ("k_flat_primary<float", "k_flat_primary"), ("k_flat_shadow<float", "k_flat_shadow"), ("k_flat_primary_sc", "k_flat_primary_sc"), ("k_flat_shadow_sc", "k_flat_shadow_sc")]


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = collections.defaultdict(dict)
    for part in ("sq1", "sq2", "sq3", "sq4"):
        for f in glob.glob(os.path.join(src, "%s_%s" % (part, tag), "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            dur = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                for pat, name in KERNELS:
                    if pat in r["Kernel_Name"]:
                        agg[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
                        dur[name].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            for (k, c), v in agg.items():
                out[k][c] = sum(v) / len(v)
            for k, v in dur.items():
                out[k]["duration_ns_while_counting_" + part] = sum(v) / len(v)
    for k, d in out.items():
        if "SQ_INSTS_VALU" in d and "duration_ns_while_counting_sq1" in d:
            # (SQ_WAVE_CYCLES is NOT a wave's lifetime: calibrated with tools/valu_issue_probe.hip it reads 1/4 of the cycles of a lone
            # wave and 1/8 at 8 waves per SIMD, so no occupancy is derived from it)
            d["derived_valu_insts_per_wave"] = d["SQ_INSTS_VALU"] / d["SQ_WAVES"]
    # what bench.py quotes beside its live timing (instruction counts per launch do not depend on the clock)
    stamp = {"tag": tag, "git_head": git_head(), "kernel_src_sha": kernel_src_sha()}
    out["_stamp"] = stamp
    json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_sq.json"), "w"), indent=1, sort_keys=True)
    quote = dict(stamp, _source="tools/summarize_sq.py from rocprofv3 --pmc SQ_* passes (tools/profile_sq.sh), per-launch averages")
    for k, d in out.items():
        if k != "_stamp" and "SQ_INSTS_VALU" in d:
            quote[k + "_n1"] = {"valu_insts": d["SQ_INSTS_VALU"], "salu_insts": d.get("SQ_INSTS_SALU"), "smem_insts": d.get("SQ_INSTS_SMEM"),
                                "branch_insts": d.get("SQ_INSTS_BRANCH"), "other_insts": (d.get("SQ_INSTS_SENDMSG") or 0) + (d.get("SQ_INSTS_VMEM") or 0) +
                                (d.get("SQ_INSTS_LDS") or 0), "waves": d.get("SQ_WAVES"),
                                "valu_active_quad_cycles": d.get("SQ_ACTIVE_INST_VALU"),
                                # where a wave's cycles go (MI355X_MICROARCH.md: WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, disjoint)
                                "wave_quad_cycles": d.get("SQ_WAVE_CYCLES"), "wait_any_quad_cycles": d.get("SQ_WAIT_ANY"),
                                "wait_inst_any_quad_cycles": d.get("SQ_WAIT_INST_ANY"), "active_inst_any_quad_cycles": d.get("SQ_ACTIVE_INST_ANY"),
                                # the scalar data cache (tools/profile_sq.sh, fourth pass)
                                "sqc_dcache_req": d.get("SQC_DCACHE_REQ"), "sqc_dcache_hits": d.get("SQC_DCACHE_HITS"), "sqc_dcache_misses": d.get("SQC_DCACHE_MISSES"),
                                "sqc_dcache_misses_duplicate": d.get("SQC_DCACHE_MISSES_DUPLICATE")}
    a, b = quote.get("k_flat_primary_sc_n1"), quote.get("k_flat_shadow_sc_n1")
    if a and b:                     # one flat frame = the primary pass + two shadow passes (the counters are per-launch averages)
        quote["k_flat_pipeline_n1"] = {k: (a.get(k) or 0) + 2 * (b.get(k) or 0) for k in a}
    json.dump(quote, open(os.path.join(ROOT, "profiles", "roofline_sq.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
