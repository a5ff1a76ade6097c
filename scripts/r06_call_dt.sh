#!/bin/bash
# launch shape of t2v_lora_drop_dt: column split WC (waves per row group) forced to 1 / 2 against the rule's choice; kernel trace per run
mkdir -p gpurun_out
for wc in 0 1 2 4; do
  if [ $wc = 0 ]; then unset T2V_DT_WC; else export T2V_DT_WC=$wc; fi
  bash scripts/profile_bench.sh dtwc$wc > /dev/null 2>&1
  echo "== T2V_DT_WC=$wc"; head -1 gpurun_out/dtwc${wc}_window.txt | cut -c1-100; grep "lora_drop_dt" gpurun_out/dtwc${wc}_window.txt | cut -c40-140
done
