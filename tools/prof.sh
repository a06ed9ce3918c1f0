#!/bin/bash
# usage: tools/prof.sh <tag> <bench args...>   -> gpurun_out/prof_<tag>/ kernel trace + stats (rocprofv3)
set -o pipefail
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o trace -- python3 bench.py --lean 1 "$@" > $out/bench.json 2> $out/bench.err
echo "rc=$?"
ls -R $out | head -20
