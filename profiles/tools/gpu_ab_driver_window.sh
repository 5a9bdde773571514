cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for k in 1 2 3; do for e in "-" "RENI_NO_L0X=1"; do E=""; [ "$e" != "-" ] && E="$e"
env $E python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-14s' % '$e', 'first-window step', round(d['ms_per_step'],4), 'kernel', round(r['kernel_avg_ms'],4))"
done; done
