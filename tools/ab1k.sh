#!/bin/bash
# like ab.sh on a 1024^2 image (the small-kernel regime): tools/ab1k.sh "<libA> <libB>" PATTERN [rounds]
libs=$1; pat=$2; rounds=${3:-2}
for r in $(seq 1 $rounds); do
  for l in $libs; do
    if [ $l = main ]; then unset GLF_LIBRARY; else export GLF_LIBRARY=$PWD/tools/dbg/libglf_$l.so; fi
    TOPN=30 tools/kstats.sh ab_$l 1024 0.005 64 0 5 0.1 | grep -E "$pat" | sed "s/^/$l r$r /" || exit 1
  done
done
