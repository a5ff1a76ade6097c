#!/bin/bash
# rocprofv3 kernel statistics of an arbitrary python script of this repo.  usage (GPU box): bash scripts/profile_cmd.sh <tag> <script> [args...]
# -> gpurun_out/<tag>_stats.txt (per-kernel count, total and average duration over the whole run, sorted by total time)
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $out -o run -- python $root/"$@" > $out.log 2>&1
db=$(find $out -name '*.db' | head -1)
python $root/scripts/rocpd_window.py $db 1000000 1 60 > $root/gpurun_out/${tag}_stats.txt 2>&1
rm -rf $out
head -45 $root/gpurun_out/${tag}_stats.txt | cut -c1-200
