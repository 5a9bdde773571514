#!/bin/bash
# usage: gpu_ab_args.sh <rounds> "<bench args A>" "<bench args B>" ...  -- same-box A/B of bench.py options (config 2 unless the
# arguments say otherwise; 50 timed steps behind 20), round-robin.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
for round in $(seq $R); do
  for v in "$@"; do
    python bench.py --no-cpu-baseline --no-also --steps 50 --warmup 20 $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-36s' % '''$v''', 'Msamples/s', round(d['value']/1e6,1), ' step ms', round(d['ms_per_step'],4), ' kernel ms', round(r['kernel_avg_ms'],4), ' rest us', round(1e3*(d['ms_per_step']-r['kernel_avg_ms']),1), ' launches', d['launches_per_step'])"
  done
done
