#!/bin/bash
# SQ / TCC counter passes of one GEMM signature (where do the waves wait?).  usage: bash scripts/pmc_gemm_counters.sh tag M N K taps rank_cols
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $root/gpurun_out/${tag}_counter_list.txt 2>&1 || true
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); out=$root/gpurun_out/${tag}_p$i
  rocprofv3 --kernel-trace --pmc $set -d $out -o pmc -- python $root/scripts/${RUN_SCRIPT:-gemm_shape_run.py} "$@" > $out.log 2>&1
  db=$(find $out -name '*.db' | head -1)
  [ -n "$db" ] && python $root/scripts/rocpd_pmc.py $db 40 > $root/gpurun_out/${tag}_p$i.txt 2>&1
  rm -rf $out
  tail -2 $out.log
done
cat $root/gpurun_out/${tag}_p*.txt | grep -v "^$" | cut -c30-200 | head -40
