#!/bin/bash
# Everything profiles/<tag>_* is made of, in one call on the GPU box:  gpurun --timeout 1200 -- 'bash tools/profile_all.sh r04j'
# then here: python3 tools/summarize_profile.py gpurun_out <tag>; python3 tools/summarize_sq.py gpurun_out <tag>; copy the bench / log files into
# profiles/; run bench.py once more ON the stamped tree and keep its line as profiles/<tag>_bench.json (its `from_profiles` then matches).
set -u
TAG=${1:-r04d}
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
bash tools/profile_gpu.sh $TAG > gpurun_out/pg_$TAG.log 2>&1; echo "profile_gpu done"
bash tools/profile_sq.sh $TAG > gpurun_out/ps_$TAG.log 2>&1; echo "profile_sq done"
for wl in config2 make_image config5 config5_100k; do
  python3 bench.py --workload $wl --no-cpu-baseline --no-seam --no-make-image --no-extras --no-configs > gpurun_out/bench_${TAG}_$wl.json 2> gpurun_out/bench_${TAG}_$wl.err; echo "bench $wl done"
done
python3 tools/f64_time.py > gpurun_out/f64_time_$TAG.log 2>&1; echo "f64 done"
python3 tools/wave_timeline.py 800 600 1 8 > gpurun_out/wave_timeline_800x600_$TAG.log 2>&1
WAVE_TIMELINE_JSON=gpurun_out/roofline_waves.json WAVE_TIMELINE_TAG=$TAG python3 tools/wave_timeline.py 1920 1080 1 8 > gpurun_out/wave_timeline_1080p_$TAG.log 2>&1; echo "timelines done"
python3 tools/shard_expect.py > gpurun_out/shard_render_times_$TAG.log 2>&1; echo "shard done"
tail -3 gpurun_out/shard_render_times_$TAG.log
