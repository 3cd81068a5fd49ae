#!/bin/bash
# on the GPU box: tools/ab.sh "<libA> <libB> ..." PATTERN [rounds] -- alternating whole-path runs, per-kernel averages for kernels matching PATTERN
libs=$1; pat=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do
  for l in $libs; do
    if [ $l = main ]; then unset GLF_LIBRARY; else export GLF_LIBRARY=$PWD/tools/dbg/libglf_$l.so; fi
    TOPN=12 tools/kstats.sh ab_$l 4096 0.005 64 0 2 | grep -E "$pat" | sed "s/^/$l r$r /" || exit 1
  done
done
