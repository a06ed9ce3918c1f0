#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <tag> "<counters>" <bench args...>  -> gpurun_out/pmc_<tag>.txt (per-kernel means; --pmc with --kernel-trace only)
set -o pipefail
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmcany_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 bench.py --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 "$@" > /dev/null 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
python3 tools/pmc_sum.py $(find $out -name "*counter_collection.csv" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
rm -rf $out
grep -A12 "k_intersect\|k_shade" $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt | head -60
