#!/bin/bash
# One ray per lane (k_render_skip) against two (k_render_skip2) over frame sizes, sample counts and both pyramid scenes: the numbers
# behind rt_capi.hip skip2_by_default.  usage: tools/skip2_sweep.sh > log   (GPU box; every leg is tools/ab.py, interleaved)
for cfg in "800 600 1 8" "1920 1080 1 8" "2560 1440 1 8" "3840 2160 1 8" "1024 768 2 8" "1024 768 4 8" "1920 1080 4 8" "2048 2048 4 8" \
           "800 600 1 9" "1920 1080 1 9" "2560 1440 1 9" "3840 2160 1 9" "1024 768 4 9" "1920 1080 2 9" "1920 1080 4 9" "4096 4096 4 9"; do
  set -- $cfg
  px=$(( $1 * $2 * $3 * $3 ))
  launches=$(( 400000000 / px + 2 )); [ $launches -gt 20 ] && launches=20
  echo -n "$cfg: "
  AB_KEY=skip_rays AB_VARIANTS=1,2 AB_LAUNCHES=$launches python3 tools/ab.py 8 $cfg 2>&1 | tail -2 | tr '\n' ' '
  echo
done
