#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of single GEMM signatures in isolation (operands alternate between two buffer sets): which signatures
# re-read their operands.   usage (GPU box): bash scripts/pmc_shapes_fetch.sh <tag>  ->  gpurun_out/<tag>.txt
tag=${1:-pmc_shapes}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
: > $root/gpurun_out/$tag.txt
for shape in "32768 320 320 1 16" "32768 320 960 3 16" "8192 640 640 1 16" "8192 640 1920 3 16" "2048 1280 1280 1 16" "2048 1280 3840 3 16" "32768 2560 320 1 16" "32768 320 2880 9 16"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    out=$root/gpurun_out/${tag}_tmp
    rocprofv3 --kernel-trace --pmc $c -d $out -o pmc -- python $root/scripts/gemm_shape_run.py $shape 10 > $out.log 2>&1
    db=$(find $out -name '*.db' | head -1)
    echo "== $shape ($c; algorithmic MB: A once + W once = $(python -c "M,N,K,t,r=map(int,'$shape'.split()); print(round((M*K//t + (N+r)*K)*2/1e6,1), 'out', round(M*(N+r)*2/1e6,1))"))" >> $root/gpurun_out/$tag.txt
    python $root/scripts/rocpd_pmc.py $db 3 2>&1 | grep -E "gemm" | cut -c1-60,100-230 >> $root/gpurun_out/$tag.txt
    rm -rf $out
  done
done
cat $root/gpurun_out/$tag.txt
