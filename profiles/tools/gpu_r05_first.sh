#!/bin/bash
# round 5, first GPU call of the L0X split: the training-path GPU tests, then a same-box A/B of the split against round 4's kernels
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_first; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_workflows.py tests/test_gpu_train_step.py tests/test_gpu_parity.py -m gpu -q -x --timeout 600 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
timeout 600 bash profiles/tools/gpu_ab_env.sh 2 "-" "RENI_NO_L0X=1" > $O/ab.txt 2>&1
cat $O/ab.txt
