#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + the two HBM PMC passes for bench.py, into gpurun_out/.
#   gpurun -- 'bash tools/profile_gpu.sh r01'        then here:  python tools/summarize_profile.py gpurun_out r01
# Counters are collected in their own runs (no --sys-trace etc. together with --pmc), program directly after `--`.
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs --steps 10 --warmup 2 --repeats 1 > gpurun_out/pmc_write_$TAG.log 2>&1
tail -c 2500 gpurun_out/bench_$TAG.json
