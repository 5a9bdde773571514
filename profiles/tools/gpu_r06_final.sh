#!/bin/bash
# round 6, final check on the committed sources: the whole -m gpu suite, smoke(), the 300-step soak, the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06final; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/gpu_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python profiles/tools/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
python profiles/tools/gpu_film_train_h256.py 2>&1 | grep "FiLM" | tee $O/film_train.txt
RENI_NO_PERSIST=1 python profiles/tools/gpu_film_train_h256.py 2>&1 | grep "FiLM" | tee -a $O/film_train.txt
python profiles/tools/gpu_film_fwd_h256.py 2>&1 | grep "samples/s" | tee $O/film_fwd.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err; echo "bench rc=$? lines=$(wc -l < $O/bench.out) last_line_bytes=$(tail -n 1 $O/bench.out | wc -c)" | tee $O/bench_rc.txt
