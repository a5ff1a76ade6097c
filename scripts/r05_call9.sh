#!/bin/bash
# Round 5: SQ counter passes over the layer shapes of the default train mode (plain / epilogue-form / pass-form launches + dt kernel,
# scripts/branch_probe.py): where the waves of the LR instantiations spend their cycles (MFMA busy, LDS, waits).
mkdir -p gpurun_out
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); out=$root/gpurun_out/r05_sq_p$i
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $out -o pmc -- python $root/scripts/branch_probe.py > $out.log 2>&1
  echo "pass $i rc=$?"
  db=$(find $out -name '*.db' | head -1)
  [ -n "$db" ] && python $root/scripts/rocpd_pmc.py $db 200 > $root/gpurun_out/r05_sq_p$i.txt 2>&1
  rm -rf $out
done
grep -h "gemm_w8_kernelILi128ELi384ELi4ELi2ELi1ELi2ELi5ELi64ELb0ELb1\|gemm_w8_kernelILi128ELi192ELi2ELi2ELi2ELi3ELi4ELi64ELb0ELb1\|lora_drop_dt" $root/gpurun_out/r05_sq_p1.txt | cut -c40-220 | head -40
