#!/bin/bash
# Runs on the GPU box (through gpurun): bench.py under rocprofv3 --stats, then the HBM-traffic PMC passes.
# usage: tools/profile_round.sh TAG      -> gpurun_out/prof_TAG/...
set -e
tag=${1:-r01}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag; rm -rf $out; mkdir -p $out
python3 bench.py --steps 3 --warmup 1 > $out/bench_plain.json 2> $out/bench_plain.err
# the profiled run holds the headline steps only (no skipping / throughput / direct-contraction legs, which launch the same kernels on
# other workloads): its per-kernel averages are the ones bench.py's roofline block reports
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-skip-leg --no-direct-leg --no-batch-leg --no-exact-leg --no-host-leg --no-nlm-leg > $out/bench_under_rocprof.json 2> $out/stats.err
for c in FETCH_SIZE WRITE_SIZE GRBM_GUI_ACTIVE; do
  echo "pmc pass $c"
  timeout -k 5 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python3 tools/run_once.py 4096 0.005 64 0 1 > $out/pmc_$c.log 2>&1
done
echo "pmc pass SQ"
timeout -k 5 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_SQ -- python3 tools/run_once.py 4096 0.005 64 0 1 > $out/pmc_SQ.log 2>&1
echo profile_round done
