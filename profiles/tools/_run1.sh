profiles/tools/gpu_variants.sh --rounds 2 "@base" > gpurun_out/r03_b16.txt 2>&1; cat gpurun_out/r03_b16.txt
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
