#!/bin/bash
for cfg in "2048 1536 1 8" "2304 1296 1 8" "1920 1080 2 8" "1600 1200 2 8" "640 480 4 8" "800 600 4 8" "640 480 8 8" \
           "3200 1800 1 9" "800 600 4 9" "1024 768 2 9" "640 480 4 9" "1280 720 2 9"; do
  set -- $cfg
  px=$(( $1 * $2 * $3 * $3 ))
  launches=$(( 400000000 / px + 2 )); [ $launches -gt 20 ] && launches=20
  echo -n "$cfg: "
  AB_KEY=skip_rays AB_VARIANTS=1,2 AB_LAUNCHES=$launches python3 tools/ab.py 8 $cfg 2>&1 | tail -2 | tr '\n' ' '
  echo
done
