TAILN=300 profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE" > gpurun_out/r03_trace12.txt 2>&1
profiles/tools/gpu_variants.sh --rounds 2 "@base" > gpurun_out/r03_b12.txt 2>&1; cat gpurun_out/r03_b12.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
