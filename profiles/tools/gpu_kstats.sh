#!/bin/bash
# usage: gpu_kstats.sh <out.md> [bench args]  -- rocprofv3 kernel-trace summary (per-kernel count / average / min duration) of one bench run
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=$1; shift
D=gpurun_out/_kt_$$
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $D.log 2>&1
python3 profiles/summarize_rocpd.py $D/k_results.db $OUT > /dev/null 2>&1 || { tail -5 $D.log; ls -R $D | head; }
python3 profiles/timeline_rocpd.py $D/k_results.db > ${OUT%.md}_timeline.txt 2>/dev/null
rm -rf $D $D.log
head -24 $OUT
