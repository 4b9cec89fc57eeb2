#!/bin/bash
# A/B two builds of the library on ONE box (timings from different gpurun boxes differ by up to ~10 %).
# usage: tools/ab.sh build/libA.so build/libB.so [sweep args...]
A=$1; B=$2; shift 2
for round in 1 2 3; do
  for lib in $A $B; do
    echo "== $lib (round $round)"
    ABD_HIP_LIB=$PWD/$lib python tools/sweep.py "$@" 2>&1 | grep cpw
  done
done
