#!/bin/bash
# Where a workload's wave cycles go and what its scalar cache does (two --pmc passes of one short bench.py leg per workload):
#   gpurun -- 'bash tools/profile_waits.sh r04d config2 make_image config5'   -> gpurun_out/waits_<tag>.log
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
: > gpurun_out/waits_$TAG.log
for WL in "$@"; do
  B="python3 bench.py --workload $WL --no-cpu-baseline --no-seam --no-make-image --no-flat --no-extras --no-configs --steps 4 --warmup 40 --repeats 1 --min-timed-region 0"
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d gpurun_out/w1_${TAG}_$WL -- $B > gpurun_out/w1_${TAG}_$WL.log 2>&1
  rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE --output-format csv -d gpurun_out/w2_${TAG}_$WL -- $B > gpurun_out/w2_${TAG}_$WL.log 2>&1
  python3 - "$TAG" "$WL" >> gpurun_out/waits_$TAG.log <<'PY'
import csv, glob, collections, sys
tag, wl = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for part in ("w1", "w2"):
    for f in glob.glob("gpurun_out/%s_%s_%s/*/*counter_collection.csv" % (part, tag, wl)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_render_skip" in k:
                acc[k.split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    rq = m.get("SQC_DCACHE_REQ", 0.0)
    line = "%-12s %-58s launches %4d" % (wl, k, n)
    if wc:
        line += "  waiting %.3f issuing %.3f stalled %.3f  VALU %.2f M SALU %.2f M SMEM %.2f M" % (
            m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc,
            m.get("SQ_INSTS_VALU", 0) / 1e6, m.get("SQ_INSTS_SALU", 0) / 1e6, m.get("SQ_INSTS_SMEM", 0) / 1e6)
    if rq:
        line += "  scalar cache: %.2f M requests, hit %.3f miss %.3f duplicate miss %.3f" % (
            rq / 1e6, m.get("SQC_DCACHE_HITS", 0) / rq, m.get("SQC_DCACHE_MISSES", 0) / rq, m.get("SQC_DCACHE_MISSES_DUPLICATE", 0) / rq)
    print(line)
PY
done
cat gpurun_out/waits_$TAG.log
