#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/profile_gpu.sh into the committed summaries under profiles/:

  profiles/<tag>_kernel_stats.csv      copy of rocprofv3 --kernel-trace --stats (per-kernel average duration)
  profiles/<tag>_pmc.json              FETCH_SIZE / WRITE_SIZE per kernel (separate --pmc passes), averages per launch
  profiles/roofline_traffic.json       what bench.py reports as roofline.traffic: HBM bytes per launch of the dominant
                                       kernels, = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (MI355X_MICROARCH.md, HBM: the counters
                                       are in KiB and gfx950's FETCH_SIZE tallies 128-B requests at 64 B -> doubled)

usage: summarize_profile.py <gpurun_out dir> <tag> [n_gpus]
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_src_sha, git_head  # noqa: E402  (the stamp bench.py compares with the sources it runs on)
SHORT = [("k_render_skip_fast", "k_render_skip"), ("k_render_skip_f32<false", "k_render_skip"), ("k_render_skip_f32_coop<false", "k_render_skip"), ("k_render_skip<float, false", "k_render_skip"), ("k_flat_primary<float", "k_flat_primary"), ("k_flat_shadow<float", "k_flat_shadow"), ("k_flat_primary_sc", "k_flat_primary_sc"), ("k_flat_shadow_sc", "k_flat_shadow_sc"),
         ("k_blit_tiles", "k_blit_tiles"), ("k_build_streams<float>", "k_build_streams"),
         ("k_resolve_samples<float>", "k_resolve_samples")]


def short(name):
    for k, v in SHORT:
        if k in name:
            return v
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    prof = os.path.join(ROOT, "profiles")
    ks = glob.glob(os.path.join(src, "prof_" + tag, "*", "*_kernel_stats.csv"))
    if ks:
        shutil.copy(ks[0], os.path.join(prof, tag + "_kernel_stats.csv"))
    pmc = collections.defaultdict(dict)
    for ctr in ("fetch", "write"):
        for f in glob.glob(os.path.join(src, "pmc_%s_%s" % (ctr, tag), "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if k:
                    agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in agg.items():
                pmc[k][c + "_KiB_avg"] = sum(v) / len(v)
                pmc[k]["launches_" + c] = len(v)
    for k, d in pmc.items():
        if "FETCH_SIZE_KiB_avg" in d and "WRITE_SIZE_KiB_avg" in d:
            d["hbm_bytes_per_launch_uncorrected"] = int((d["FETCH_SIZE_KiB_avg"] + d["WRITE_SIZE_KiB_avg"]) * 1024)
            d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KiB_avg"] + d["WRITE_SIZE_KiB_avg"]) * 1024)
    stamp = {"tag": tag, "git_head": git_head(), "kernel_src_sha": kernel_src_sha()}
    pmc["_stamp"] = stamp
    json.dump(pmc, open(os.path.join(prof, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    tpath = os.path.join(prof, "roofline_traffic.json")
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if traffic.get("kernel_src_sha") != stamp["kernel_src_sha"]:
        traffic = {}                                     # figures of other kernel sources do not mix with these
    traffic.update(stamp)
    for k in ("k_render_skip", "k_flat_primary", "k_flat_shadow", "k_flat_primary_sc", "k_flat_shadow_sc"):
        if k in pmc and "hbm_bytes_per_launch" in pmc[k]:
            traffic["%s_n%d" % (k, n)] = pmc[k]["hbm_bytes_per_launch"]
    traffic["_source"] = "tools/summarize_profile.py from rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, tag " + tag
    json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
    print(json.dumps(pmc, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
