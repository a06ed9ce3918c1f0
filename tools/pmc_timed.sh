#!/bin/bash
# usage (GPU box): bash tools/pmc_timed.sh <round>   -> gpurun_out/pmc_timed_<round>/ and profiles/<round>_pmc_timed.json
# counters of the TIMED configuration (the default: four iterations in flight, graph replay), one rocprofv3 --pmc pass with --kernel-trace only; the
# program itself after `--`, no hop.  Summed over every dispatch of the run by tools/pmc_timed_sum.py.
set -o pipefail
rd=${1:-r4}
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/pmc_timed_$rd; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd $R
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/pmc -o pmc -- python3 bench.py --lean 1 --cpu-seconds 0 --bvh-scene "" --steps 8 --warmup 2 --profile-steps 1 > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
python3 tools/pmc_timed_sum.py $(find $out/pmc -name "*counter_collection.csv" | head -1) $out/bench.json > profiles/${rd}_pmc_timed.json || exit 1
find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete
cp profiles/${rd}_pmc_timed.json $out/${rd}_pmc_timed.json      # (files written outside gpurun_out/ do not come back from the GPU box)
cat profiles/${rd}_pmc_timed.json
