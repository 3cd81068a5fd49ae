#!/bin/bash
# on the GPU box: tools/pmc_kernel.sh TAG SUBSTR [run_once args...] -> per-launch PMC sums of the kernels matching SUBSTR
# (separate --pmc passes: HBM traffic one counter per pass, SQ activity, LDS, waits)
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
tag=$1; sub=$2; shift; shift
out=gpurun_out/pmck_$tag; rm -rf $out; mkdir -p $out
i=0
for set in ${PMC_SETS:-"FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE"}; do
  i=$((i+1))
  echo "pass $i: $set"
  timeout -k 5 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/p$i -- python3 tools/run_once.py "$@" > $out/p$i.log 2>&1 || { echo "pass $i ($set) failed"; grep -m2 "error code\|Could not" $out/p$i.log; }
done
python3 tools/pmc_summary.py $out "$sub"
