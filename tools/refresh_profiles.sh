#!/bin/bash
# Runs on the GPU box (gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh'): refreshes every measurement the docs cite into gpurun_out/final/.
#   kernel stats : rocprofv3 --kernel-trace --stats of `python3 bench.py` (cornell, then the BVH scene)
#   traffic      : rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, single stream, eager launches
#   bench        : the plain default `python3 bench.py` line
set -o pipefail
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
for sc in cornell.txt bunny_cornell.txt; do
  tag=${sc%%.txt}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -o trace -- python3 bench.py --scene $sc --bvh-scene "" --cpu-seconds 1 > $out/bench_under_rocprof_$tag.json 2> $out/stats_$tag.err || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${ctr}_$tag -o pmc -- python3 bench.py --scene $sc --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0.1 --profile-steps 1 --in-flight 1 --graph 0 > /dev/null 2> $out/pmc_${ctr}_$tag.err || exit 1
  done
  echo "done $sc"
done
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
python3 tools/pmc_traffic.py $out > $out/pmc_traffic.json || exit 1
find $out -name "*.csv" -size +2M -delete
tail -c 600 $out/bench.json
