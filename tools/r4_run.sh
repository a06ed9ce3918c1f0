#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run10; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -4 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log || exit 1
timeout -k 10 600 python3 tools/diag/ab_scenes.py "base exact" "cornell.txt,0,3;cornell.txt,0,1;cornell.txt,2,3;mis_test.txt,2,3,0,1024,1024,16;metallic_analytic.txt,2,3,0,1024,1024,16;bunny_cornell.txt,0,3" 3 > $out/ab.txt 2>&1; tail -13 $out/ab.txt
