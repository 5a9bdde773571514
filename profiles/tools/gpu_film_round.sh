#!/bin/bash
# usage: gpu_film_round.sh <outdir>  -- FiLM tests + a kernel-trace summary of the fused FiLM step at B = 64
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_film.py -m gpu -x -q > $O/film.log 2>&1; tail -12 $O/film.log
D=gpurun_out/_kt_$$
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 profiles/tools/gpu_prof_film.py 64 10 > $O/film_prof.log 2>&1
tail -2 $O/film_prof.log
python3 profiles/summarize_rocpd.py $D/k_results.db $O/kernel_stats_film.md > /dev/null 2>&1 || ls -R $D | head
python3 profiles/timeline_rocpd.py $D/k_results.db > $O/film_timeline.txt 2>/dev/null
rm -rf $D
head -30 $O/kernel_stats_film.md
tail -40 $O/film_timeline.txt
