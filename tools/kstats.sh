#!/bin/bash
# on the GPU box: tools/kstats.sh TAG [run_once args...] -> top kernels of one whole-path run
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; shift
rm -rf gpurun_out/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python3 tools/run_once.py "$@" > gpurun_out/ks_$tag.log 2>&1 || exit 1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/ks_$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:${TOPN:-8}]:
    print("  %-52s %5s %10.3f ms  avg %9.1f us"%(r["Name"].split("(")[0][:52],r["Calls"],float(r["TotalDurationNs"])/1e6,float(r["AverageNs"])/1e3))
PY
