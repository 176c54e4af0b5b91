#!/bin/bash
# A/B of two builds of librtrace_hip.so on one box, interleaved: tools/ab_libs.sh old.so new.so [rounds] -- each leg is tools/ab.py
# with the default loops at 1080p, 800x600 and `make image`; the library file is swapped between legs.
set -e
OLD=$1; NEW=$2; ROUNDS=${3:-2}
# CFGS: newline-separated "w h spp level" lines (default: the one-ray kernel's three workloads)
DEFAULT_CFGS=$'1920 1080 1 8\n800 600 1 8\n1024 768 4 8'
LIB=${AB_LIB:-tests/c/librtrace_hip_test.so}      # what tools/ab.py loads (the -DRT_TEST_HOOKS build)
cp $LIB /tmp/_lib_keep.so
for r in $(seq $ROUNDS); do
  for leg in old new; do
    if [ $leg = old ]; then cp $OLD $LIB; else cp $NEW $LIB; fi
    while read -r cfg; do
      [ -z "$cfg" ] && continue
      echo -n "$leg $cfg: "; AB_VARIANTS=23 AB_LAUNCHES=${AB_LAUNCHES:-20} python3 tools/ab.py ${AB_ROUNDS:-30} $cfg 2>&1 | tail -1
    done <<< "${CFGS:-$DEFAULT_CFGS}"
  done
done
cp /tmp/_lib_keep.so $LIB
