#!/bin/bash
# round 4, first GPU call: the GPU test tier, A/B of the cost-ordered work list, single-stream kernel stats of the bunny scene, the bench line
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run1; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" | tee -a $out/pytest.log; tail -5 $out/pytest.log
timeout -k 10 600 python3 tools/diag/ab_env.py "PTK_LIST_SORT=0;PTK_LIST_SORT=1" "bunny_cornell.txt,0,3;bunny_cornell.txt,0,1;bunny_cornell.txt,2,3;bunny_cornell.txt,0,3,0,2048,2048,16" 3 > $out/ab_sort.txt 2>&1; tail -9 $out/ab_sort.txt
for v in 0 1; do
  PTK_LIST_SORT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_sort$v -o trace -- python3 bench.py --scene bunny_cornell.txt --bvh-scene "" --cpu-seconds 0 --lean 1 --in-flight 1 --graph 0 --steps 8 --warmup 2 > $out/bench_sort$v.json 2> $out/stats_sort$v.err
  f=$(find $out/stats_sort$v -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_sort$v.csv; head -8 $out/kernel_stats_sort$v.csv | cut -c1-150
done
find $out -name "*.csv" -size +1M -delete; find $out -name "*.db" -delete
timeout -k 10 900 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc $?"; tail -c 1500 $out/bench.json
