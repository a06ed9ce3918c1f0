#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run7; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -6 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log || exit 1
timeout -k 10 600 python3 tools/diag/ab_env.py "PTK_THREADED_COPIES=6;PTK_THREADED_COPIES=1" "bunny_cornell.txt,0,3;bunny_cornell.txt,2,3;bunny_cornell.txt,2,1;bunny_cornell.txt,1,3;bunny_cornell.txt,2,3,0,2048,2048,16" 3 > $out/ab_copies.txt 2>&1; tail -12 $out/ab_copies.txt
