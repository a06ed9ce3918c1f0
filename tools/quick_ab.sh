#!/bin/bash
# usage (GPU box): bash tools/quick_ab.sh <tag> [scene]   -> gpurun_out/ab_<tag>/: the default bench line, the single-stream kernel stats
# (rocprofv3 --kernel-trace --stats) and one SQ PMC pass (--pmc with --kernel-trace only), summarised per kernel.  For A/B measurements.
set -o pipefail
tag=$1; sc=${2:-cornell.txt}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/ab_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
python3 bench.py --scene $sc --steps 20 --warmup 5 --cpu-seconds 0 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1 -o trace -- python3 bench.py --lean 1 --scene $sc --bvh-scene "" --cpu-seconds 0 --steps 8 --warmup 2 --in-flight 1 --graph 0 > $out/bench_inflight1.json 2> $out/stats1.err || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $out/pmc_SQ1 -o pmc -- python3 bench.py --lean 1 --scene $sc --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --profile-steps 1 --in-flight 1 --graph 0 > /dev/null 2> $out/pmc_SQ1.err || exit 1
python3 tools/pmc_sum.py $(find $out/pmc_SQ1 -name "*counter_collection.csv" | head -1) > $out/pmc_SQ1_summary.txt
f=$(find $out/stats1 -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats_inflight1.csv
find $out -name "*.csv" -size +1M -delete
python3 - <<PY
import json,csv
d=json.load(open("$out/bench.json"))
print("value", d["value"], "ms/iter", d["ms_per_iteration"], "ref_order", d["reference_order"]["value"], "bvh", d.get("roofline_bvh_scene",{}).get("mrays_per_s"))
for r in csv.DictReader(open("$out/kernel_stats_inflight1.csv")):
    print("  %-60s calls %6s avg %9.2f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
PY
grep -A7 "k_intersect\|k_shade_binned" $out/pmc_SQ1_summary.txt | grep "k_\|SQ_INSTS_VALU\|SQ_WAVES " 
