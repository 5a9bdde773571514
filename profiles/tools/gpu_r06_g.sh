#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
for d in f32 bf16; do python profiles/tools/gpu_g16_film.py $d 2>&1 | grep G16; RENI_NO_PERSIST=1 python profiles/tools/gpu_g16_film.py $d 2>&1 | grep G16; done | tee $O/g16.txt
