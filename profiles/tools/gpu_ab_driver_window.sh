#!/bin/bash
# The L0X split against round 4's kernels (RENI_NO_L0X=1) in the DRIVER'S window: bench.py --steps 20 --warmup 5, first window of a fresh
# process, --no-also; alternating, three repetitions.  -> profiles/r05_ab_driver_window.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 1 2 3; do for e in "-" "RENI_NO_L0X=1"; do E=""; [ "$e" != "-" ] && E="$e"
env $E python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s' % '$e', 'first-window step', round(d['ms_per_step'],4), 'kernel', round(r['kernel_avg_ms'],4))"
done; done
