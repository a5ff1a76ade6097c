#!/bin/bash
# HBM traffic of the step's heaviest GEMM shape (conv 3x3, M=32768 N=320 K=2880) with the tile the autotuner picks for it
# (128x320, pinned so that counter collection cannot perturb the choice): separate FETCH_SIZE / WRITE_SIZE passes.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export T2V_GEMM_FORCE_CFG=5,2,1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/pmc2_$c -o pmc -- python $root/scripts/pmc_shape.py > $root/gpurun_out/pmc2_$c.log 2>&1
  db=$(find $root/gpurun_out/pmc2_$c -name '*.db' | head -1)
  python $root/scripts/rocpd_pmc.py $db 4 > $root/gpurun_out/pmc2_$c.txt 2>&1
  rm -rf $root/gpurun_out/pmc2_$c
done
grep -h "us/launch" $root/gpurun_out/pmc2_FETCH_SIZE.log; cat $root/gpurun_out/pmc2_FETCH_SIZE.txt $root/gpurun_out/pmc2_WRITE_SIZE.txt | cut -c1-200
