#!/bin/bash
# other configurations, same box, old vs new library (the epilogue rework changed register allocation in every gemm_w8 kernel)
mkdir -p gpurun_out
for cfg in c4 c3 c1; do
  for which in old new; do
    if [ $which = old ]; then export T2V_LIB_FILE=$PWD/build_ab/libt2v_old.so; else unset T2V_LIB_FILE; fi
    timeout 900 python bench.py --config $cfg --no-cpu-baseline --no-default-mode --no-roofline --steps 12 > gpurun_out/epi3_${cfg}_$which.json 2> gpurun_out/epi3_${cfg}_$which.err
    grep '^{' gpurun_out/epi3_${cfg}_$which.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$cfg $which: ms/step', d['ms_per_step'])"
  done
done
