#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run9; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -6 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log || exit 1
timeout -k 10 600 python3 tools/diag/ab_scenes.py "noinl exact" "bunny_cornell.txt,0,1;bunny_cornell.txt,0,3;bunny_cornell.txt,2,3;bunny_cornell.txt,0,3,0,2048,2048,16" 3 > $out/ab_inline.txt 2>&1; tail -9 $out/ab_inline.txt
PTK_WIDE_LPR=4 timeout -k 10 300 python3 tools/diag/ab_scenes.py "noinl exact" "bunny_cornell.txt,0,3;bunny_cornell.txt,0,3,0,2048,2048,16" 2 > $out/ab_inline_lpr4.txt 2>&1; tail -5 $out/ab_inline_lpr4.txt
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o trace -- python3 bench.py --scene bunny_cornell.txt --bvh-scene "" --cpu-seconds 0 --lean 1 --in-flight 1 --graph 0 --steps 8 --warmup 2 > $out/bench1.json 2> $out/stats.err
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_inflight1.csv; head -4 $out/kernel_stats_inflight1.csv | cut -c1-60,220-300
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
for lpr in 4 2; do PTK_WIDE_STATS=2 PTK_WIDE_LPR=$lpr timeout -k 10 120 python3 tools/diag/lane_busy.py bunny_cornell.txt 4 2>/dev/null >> $out/lane_busy.txt; done; cat $out/lane_busy.txt
