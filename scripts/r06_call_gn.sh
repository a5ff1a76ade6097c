#!/bin/bash
# gn_apply / gn_stats with compile-time SiLU / dropout / addend flags: norm tests, then a same-box A/B against the previous library
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "norm or gn or resnet or temporal_conv or groupnorm or dropout" > gpurun_out/gn_pytest.log 2>&1
tail -2 gpurun_out/gn_pytest.log
bash scripts/ab_bench.sh build_ab/libt2v_old.so 2 2>&1 | tee gpurun_out/gn_ab.txt
for which in old new; do
  if [ $which = old ]; then export T2V_LIB_FILE=$PWD/build_ab/libt2v_old.so; else unset T2V_LIB_FILE; fi
  bash scripts/profile_bench.sh gn_$which > /dev/null 2>&1; grep "gn_" gpurun_out/gn_${which}_window.txt | cut -c1-140 | head -14
done
