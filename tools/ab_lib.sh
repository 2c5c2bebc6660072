#!/bin/bash
# A/B of the library builds under variants/ on one GPU box: every build renders the given scenes in turn, twice over
# (run through gpurun from the repository root): tools/ab_lib.sh "smoke 7680 4320 64" "final 1920 1080 256" ...
for round in 1 2; do
  for f in variants/*.so; do
    for sc in "$@"; do
      printf "%-22s %-26s " "$f" "$sc"
      AB_CHECK=0 RT_AMD_LIB=$PWD/$f timeout -k 10 300 python tools/pool_ab.py $sc pool=0 | tail -1 || exit 1
    done
  done
done
