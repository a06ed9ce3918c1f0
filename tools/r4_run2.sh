#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r4_run2; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
for lpr in 4 2; do for srt in 0 1; do
  PTK_WIDE_STATS=2 PTK_LIST_SORT=$srt PTK_WIDE_LPR=$lpr timeout -k 10 120 python3 tools/diag/lane_busy.py bunny_cornell.txt 4 >> $out/lane_busy.txt 2>&1 || { tail -5 $out/lane_busy.txt; exit 1; }
done; done
cat $out/lane_busy.txt
timeout -k 10 600 python3 tools/diag/ab_scenes.py "exact persist" "cornell.txt,0,1;cornell.txt,0,3;cornell.txt,2,1;cornell.txt,2,3;mis_test.txt,2,3,0,1024,1024,16" 3 > $out/ab_persist.txt 2>&1; tail -12 $out/ab_persist.txt
