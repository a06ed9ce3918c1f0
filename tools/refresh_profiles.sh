#!/bin/bash
# Runs on the GPU box (gpurun --timeout 1100 -- 'bash tools/refresh_profiles.sh [round]'): refreshes every measurement the docs cite into
# gpurun_out/final_<round>/ and copies the summaries into profiles/<round>_* (tools/collect_profiles.py).
#   kernel stats : rocprofv3 --kernel-trace --stats of `python3 bench.py` (cornell = the default line, then the BVH scene), in the timed
#                  configuration (3 iterations in flight, graph replay) and single-stream for both pipelines (0 = fused default, 8 = staged)
#   PMC          : rocprofv3 --pmc, separate passes with --kernel-trace only (never with a sys/hip/hsa trace), single stream, eager launches:
#                  FETCH_SIZE, WRITE_SIZE (one counter per pass) and two SQ passes (VALU / wave occupancy / LDS), per pipeline;
#                  one SQ pass of the contracted flavour (libptk_fma.so)
#   bench        : the plain default `python3 bench.py` line, and the driver-style `--steps 20 --warmup 5`
set -o pipefail
RD=${1:-r5}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/final_$RD
PART=${2:-all}          # 1: kernel stats + PMC passes; 2: PMC of the timed configuration, bench lines, other configs; all: both (needs ~20 minutes)
if [ "$PART" != 2 ]; then rm -rf $out; fi
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
SQ1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
SQ2="SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
if [ "$PART" != 2 ]; then
for sc in cornell.txt bunny_cornell.txt; do
  tag=${sc%%.txt}
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -o trace -- python3 bench.py --scene $sc --bvh-scene "" --cpu-seconds 0 --lean 1 > $out/bench_under_rocprof_$tag.json 2> $out/stats_$tag.err || exit 1
  for p in 0 8; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats1_${tag}_p$p -o trace -- python3 bench.py --scene $sc --bvh-scene "" --cpu-seconds 0 --lean 1 --in-flight 1 --graph 0 --pipeline $p > $out/bench_under_rocprof_inflight1_${tag}_p$p.json 2> $out/stats1_${tag}_p$p.err || exit 1
    for ctr in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${ctr}_${tag}_p$p -o pmc -- python3 bench.py --scene $sc --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 --pipeline $p > /dev/null 2> $out/pmc_${ctr}_${tag}_p$p.err || exit 1
    done
    rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $out/pmc_SQ1_${tag}_p$p -o pmc -- python3 bench.py --scene $sc --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 --pipeline $p > /dev/null 2> $out/pmc_SQ1_${tag}_p$p.err || exit 1
    rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $out/pmc_SQ2_${tag}_p$p -o pmc -- python3 bench.py --scene $sc --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 --pipeline $p > /dev/null 2> $out/pmc_SQ2_${tag}_p$p.err || exit 1
  done
  echo "done $sc"
done
# the contracted flavour: how many VALU instructions -ffp-contract=fast removes
for p in 0 8; do
  PTK_FLAVOUR=fma rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $out/pmc_SQ1_cornell_fma_p$p -o pmc -- python3 bench.py --bvh-scene "" --steps 4 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 --pipeline $p > /dev/null 2> $out/pmc_SQ1_cornell_fma_p$p.err || exit 1
done
python3 tools/pmc_traffic.py $out > $out/${RD}_pmc.json || exit 1
cp $out/${RD}_pmc.json profiles/${RD}_pmc.json          # bench.py reads the committed summary: use the fresh one for the lines below
python3 tools/collect_profiles.py $RD > /dev/null
fi
if [ "$PART" = 1 ]; then find $out -name "*.csv" -size +2M -delete; find $out -name "*.db" -delete; exit 0; fi
bash tools/pmc_wait.sh $RD > $out/pmc_wait.log 2>&1 || { tail -5 $out/pmc_wait.log; exit 1; }      # where the wait goes (round 5) -> profiles/<round>_pmc_wait.json
# timelines of the steady state, four iterations in flight (tools/timeline.py): how many kernels run at once, per-queue gaps
for c in "cornell.txt 0 4 800" "bunny_cornell.txt 0 4 400"; do set -- $c; tag=${1%%.txt}
  rocprofv3 --kernel-trace --output-format csv -d $out/tl_$tag -o t -- python3 tools/diag/render_loop.py $c > $out/tl_$tag.txt 2> $out/tl_$tag.err || true
  f=$(find $out/tl_$tag -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && { echo "== $c: $(cat $out/tl_$tag.txt) (under rocprofv3 --kernel-trace)"; python3 tools/timeline.py $f; } > profiles/${RD}_timeline_$tag.txt 2>&1
  rm -rf $out/tl_$tag; cp profiles/${RD}_timeline_$tag.txt $out/
done
cp profiles/${RD}_pmc_wait.json $out/      # (files written outside gpurun_out/ do not come back from the GPU box)
bash tools/pmc_timed.sh $RD > $out/pmc_timed.log 2>&1 || { tail -5 $out/pmc_timed.log; exit 1; }
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench_driver_style.err || exit 1
bash tools/other_configs.sh > $out/other_configs.txt 2>&1 || true
cp $out/other_configs.txt profiles/${RD}_other_configs.txt
python3 tools/collect_profiles.py $RD > /dev/null
find $out -name "*.csv" -size +2M -delete
tail -c 600 $out/bench.json
