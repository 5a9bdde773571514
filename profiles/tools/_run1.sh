mkdir -p gpurun_out/r03b
TAILN=300 profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE" > gpurun_out/r03b/cycle_trace_final.txt 2>&1
python profiles/tools/gpu_parity_numbers.py > gpurun_out/r03b/parity_numbers.txt 2>&1
python profiles/tools/gpu_batch_sweep.py > gpurun_out/r03b/batch_sweep.txt 2>&1; tail -6 gpurun_out/r03b/batch_sweep.txt
