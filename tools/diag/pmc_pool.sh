#!/bin/bash
# GPU box: vector instructions and wave cycles of the wide walk, one ray per group against the pooled kernel (bunny scene, single stream):  bash tools/diag/pmc_pool.sh [W H depth]
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r5_pool_pmc; rm -rf $out; mkdir -p $out; cd $R
W=${1:-0}; H=${2:-0}; D=${3:--1}
for pool in 0 256 512; do
  for lpr in 2 4; do
    tag=pool${pool}_lpr$lpr
    PTK_WIDE_POOL=$pool PTK_WIDE_LPR=$lpr rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $out/$tag -o pmc -- python3 tools/diag/render_loop.py bunny_cornell.txt 0 1 8 $W $H $D > $out/$tag.log 2> $out/$tag.err || { tail -3 $out/$tag.err; continue; }
    python3 tools/pmc_sum.py $(find $out/$tag -name "*counter_collection.csv" | head -1) > $out/$tag.txt; rm -rf $out/$tag
    echo "== $tag"; grep -A8 "k_traverse" $out/$tag.txt | grep -v "^--"
  done
done
