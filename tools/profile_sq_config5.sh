#!/bin/bash
# SQ counters of config 5 (4096x4096 spp 4, 87,381 spheres: k_render_skip2 + k_resolve_samples).  gpurun -- 'bash tools/profile_sq_config5.sh r03d'
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
B="python3 bench.py --workload config5 --no-cpu-baseline --no-seam --no-flat --no-extras --steps 2 --warmup 1 --repeats 1 --min-timed-region 0"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/c5sq1_$TAG -- $B > gpurun_out/c5sq1_$TAG.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_SALU SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/c5sq2_$TAG -- $B > gpurun_out/c5sq2_$TAG.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c5kt_$TAG -- $B > gpurun_out/c5kt_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("c5sq1_$TAG", "c5sq2_$TAG"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % tag):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:40]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
            if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
    for k in acc:
        print(tag, k, "launches", n[k], {c: round(v / max(1, n[k])) for c, v in acc[k].items()})
for f in glob.glob("gpurun_out/c5kt_$TAG/*/*kernel_stats.csv"):
    print(open(f).read()[:1500])
PY
