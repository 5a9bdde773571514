#!/bin/bash
# samples sclk / power while bench.py runs (is the training kernel power-limited?)
(for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | tr '\n' ' '; echo; sleep 0.4; done) > gpurun_out/clocks.log &
SP=$!
python bench.py --steps 6000 --warmup 3 2>/dev/null | tail -1 | cut -c1-200
wait $SP
sort gpurun_out/clocks.log | uniq -c | sort -rn | head -12
