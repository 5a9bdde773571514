TAILN=300 profiles/tools/gpu_variants.sh --rounds 1 --cmd "python profiles/tools/gpu_trace.py" "-DRENI_TRACE" "-DRENI_TRACE -DRENI_EXP=524288" > gpurun_out/r03_trace13.txt 2>&1
