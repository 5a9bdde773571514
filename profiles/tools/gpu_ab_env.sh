#!/bin/bash
# usage: gpu_ab_env.sh <rounds> "<ENV=1 ...|->" "<ENV=1 ...|->" ...   -- same-box A/B of environment selectors: bench.py (config 2, 50 timed
# steps behind 20) under each setting in turn, round-robin.  "-" = clean environment.
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=$1; shift
for round in $(seq $R); do
  for v in "$@"; do
    E=""; [ "$v" != "-" ] && E="$v"
    env $E python bench.py --no-cpu-baseline --no-also --steps 50 --warmup 20 ${BENCH_ARGS} 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-28s' % '''$v''', 'Msamples/s', round(d['value']/1e6,1), ' step ms', round(d['ms_per_step'],4), ' kernel ms', round(r['kernel_avg_ms'],4), ' tail us', round(1e3*(d['ms_per_step']-r['kernel_avg_ms']),1), d['config']['paths']['env_overrides'])"
  done
done
