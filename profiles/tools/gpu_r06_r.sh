#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wide_train.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q 2>&1 | tail -3 | tee $O/tests.txt
for k in 1 2; do for lib in reni_amd/lib/libreni_hip.so reni_amd/csrc/_build/libreni_r06_before_dwfrag2.so; do
RENI_HIP_LIB=$PWD/$lib python bench.py --config c2_h256 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-64s' % '$lib', 'step', round(d['ms_per_step'],4), 'chain', round(r['kernel_avg_ms'],4), 'dw_frag+head per step', round(r['kernels'][1]['ms_per_step'],4))"
done; done | tee $O/ab_dwfrag.txt
