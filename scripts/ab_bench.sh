#!/bin/bash
# Same-box A/B of two builds of libt2v_hip.so (box-to-box spread is ~10 %, larger than most kernel changes):
#   bash scripts/ab_bench.sh build_ab/libt2v_old.so [rounds] [old tile table]   -> gpurun_out/ab_{old,new}_<k>.json, one summary line each
old=$1; rounds=${2:-2}; oldtab=${3:-}
for k in $(seq 1 $rounds); do
  for which in old new; do
    if [ $which = old ]; then export T2V_LIB_FILE=$PWD/$old; [ -n "$oldtab" ] && export T2V_GEMM_TABLE_FILE=$PWD/$oldtab; else unset T2V_LIB_FILE T2V_GEMM_TABLE_FILE; fi
    python bench.py --no-cpu-baseline --no-default-mode --steps 30 > gpurun_out/ab_${which}_$k.json 2> gpurun_out/ab_${which}_$k.err
    grep '^{' gpurun_out/ab_${which}_$k.json | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); print('$which $k: ms/step', d['ms_per_step'], 'gemm frac', d['roofline']['frac'], 'gemm kernel ms', d['roofline']['kernel_ms_per_step'], 'host ms', d['config'].get('host_ms_per_step'))"
  done
done
