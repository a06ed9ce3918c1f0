#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <bench args...>  -> gpurun_out/pmc_<tag>/ per-dispatch counter CSV (rocprofv3 --pmc)
set -o pipefail
tag=$1; ctrs=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out -o pmc -- python3 bench.py --lean 1 "$@" > $out/bench.json 2> $out/bench.err
echo "rc=$?"
ls $out
