#!/bin/bash
# the driver's command under rocprofv3 --kernel-trace --stats, with the bench line of the SAME process kept beside the summary
# -> gpurun_out/r06_u/{kernel_stats_c2.md, bench_c2_profiled.json, step_timeline_c2.txt}
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_u; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o k -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c2_profiled.txt 2> $O/kt.log
tail -n 1 $O/bench_c2_profiled.txt > $O/bench_c2_profiled.json
python3 profiles/summarize_rocpd.py $O/kt/k_results.db $O/kernel_stats_c2.md 20 5 | tail -8
python3 profiles/timeline_rocpd.py $O/kt/k_results.db > $O/step_timeline_c2.txt 2>/dev/null
rm -rf $O/kt
python3 -c "
import json; d=json.loads(open('$O/bench_c2_profiled.json').read()); r=d['roofline']
print('line: ms_per_step', d['ms_per_step'], 'kernel_avg_ms', r['kernel_avg_ms'], 'kernels', [(k['name'], k['avg_ms']) for k in r.get('kernels', [])], 'sustained', d.get('sustained', {}).get('ms_per_step'))"
