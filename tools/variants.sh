#!/bin/bash
# on the GPU box: tools/variants.sh NAME... -> top kernels of one whole-path run per profiling build tools/dbg/libglf_NAME.so ("base" = the product library)
for n in "$@"; do
  echo "== $n"
  if [ $n = base ]; then unset GLF_LIBRARY; else export GLF_LIBRARY=$GRAFT_REPO_ROOT/tools/dbg/libglf_$n.so; fi
  TOPN=4 bash tools/kstats.sh v_$n 4096 0.005 64 0 1 || exit 1
done
