#!/bin/bash
# usage (GPU box): bash tools/pmc_wait.sh <round>   -> gpurun_out/pmc_wait_<round>/*.txt + profiles/<round>_pmc_wait.json
# Where a wave's WAIT goes (VERDICT r4 #2a): separate rocprofv3 --pmc passes (with --kernel-trace only), single stream, eager launches, over
#   cornell.txt  (fused default: k_shade_trace)   and   bunny_cornell.txt (k_traverse_wide and the staged kernels):
#   SQ_A : SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
#   SQ_B : SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
#   TA   : TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum          L1  : TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum        L2 : TCC_HIT_sum TCC_MISS_sum
# There is no barrier counter: barrier + scalar wait = SQ_WAIT_ANY minus what the LDS / vector-memory counters attribute (tools/pmc_wait_sum.py).
set -o pipefail
RD=${1:-r5}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_wait_$RD
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $R
declare -A SETS
SETS[SQ_A]="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
SETS[SQ_B]="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM"
SETS[TA]="TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"
SETS[L1]="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"
SETS[L2]="TCC_HIT_sum TCC_MISS_sum"
for sc in cornell.txt bunny_cornell.txt; do
  tag=${sc%%.txt}
  for s in SQ_A SQ_B TA L1 L2; do
    rocprofv3 --pmc ${SETS[$s]} --kernel-trace --output-format csv -d $out/${tag}_$s -o pmc -- python3 bench.py --scene $sc --bvh-scene "" --steps 3 --warmup 1 --cpu-seconds 0 --lean 1 --profile-steps 1 --in-flight 1 --graph 0 > /dev/null 2> $out/${tag}_$s.err || { echo "pass $s on $sc failed"; tail -3 $out/${tag}_$s.err; continue; }
    f=$(find $out/${tag}_$s -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 tools/pmc_sum.py $f > $out/${tag}_$s.txt
    rm -rf $out/${tag}_$s
  done
  echo "done $sc"
done
python3 tools/pmc_wait_sum.py $out > profiles/${RD}_pmc_wait.json && cat profiles/${RD}_pmc_wait.json | head -80
