#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06i; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/tests.txt
python -m pytest tests/test_gpu_trajectory.py -q -s -k "g15" 2>&1 | grep "G15" | cut -c1-300 | tee $O/g15.txt
for k in 1 2 3; do for lib in reni_amd/lib/libreni_hip.so reni_amd/csrc/_build/libreni_r06_before_consistent_training.so; do
RENI_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --no-also --steps 50 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-70s' % '$lib', 'step', round(d['ms_per_step'],4), 'kernel', round(r['kernel_avg_ms'],4), 'ring', round(r['kernels'][1]['avg_ms'],4))"
done; done | tee $O/ab_consistent_training.txt
