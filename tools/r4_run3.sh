#!/bin/bash
# round 4, GPU call 3: mesh parity tests with the front/back work list, A/B of its variants, single-stream kernel stats, lane-busy counters
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run3; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_golden.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -4 $out/pytest.log
grep -q "pytest rc 0" $out/pytest.log || exit 1
timeout -k 10 700 python3 tools/diag/ab_env.py "PTK_LIST_SORT=0;PTK_LIST_FRONT_KEY=20;PTK_LIST_FRONT_KEY=0;PTK_LIST_FRONT_KEY=14;PTK_LIST_FRONT_KEY=28" "bunny_cornell.txt,0,1;bunny_cornell.txt,0,3;bunny_cornell.txt,2,3;bunny_cornell.txt,0,3,0,2048,2048,16" 2 > $out/ab_sort.txt 2>&1; tail -22 $out/ab_sort.txt
for v in 0 1; do
  PTK_LIST_SORT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_sort$v -o trace -- python3 bench.py --scene bunny_cornell.txt --bvh-scene "" --cpu-seconds 0 --lean 1 --in-flight 1 --graph 0 --steps 8 --warmup 2 > $out/bench_sort$v.json 2> $out/stats_sort$v.err
  f=$(find $out/stats_sort$v -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_sort$v.csv; head -6 $out/kernel_stats_sort$v.csv | cut -c1-60,230-300
done
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
for lpr in 4 2; do for srt in 0 1; do
  PTK_WIDE_STATS=2 PTK_LIST_SORT=$srt PTK_WIDE_LPR=$lpr timeout -k 10 120 python3 tools/diag/lane_busy.py bunny_cornell.txt 4 2>/dev/null >> $out/lane_busy.txt
done; done
cat $out/lane_busy.txt
