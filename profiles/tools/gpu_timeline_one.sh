#!/bin/bash
# One rocprofv3 kernel-trace run of config 2 with the library RENI_HIP_LIB names; prints one step's kernel timeline.
# Used as   gpu_variants.sh --rounds 1 --cmd "bash profiles/tools/gpu_timeline_one.sh" <flags> ...
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
D=gpurun_out/_tl_$$
rocprofv3 --kernel-trace --stats -d $D -o k -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also > $D.log 2>&1
python3 profiles/timeline_rocpd.py $D/k_results.db 2>/dev/null | head -15 | cut -c1-110 | grep -v "^$"
rm -rf $D $D.log
