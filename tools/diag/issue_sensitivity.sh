#!/bin/bash
# usage (GPU box): bash tools/diag/issue_sensitivity.sh [reps]   — how the headline's time answers to the VECTOR INSTRUCTION COUNT of k_shade_trace.
# Diagnostic builds libptk_pad<N>.so (tools/diag/build_variant.sh pad<N> -DPTK_DIAG_PAD_VALU=<N>: N more fma per wave and launch on live registers, no
# memory, same results) and the contracted flavour (fma: ~5 % FEWER instructions) against the product, alternating child processes; the timed configuration
# (three iterations in flight, graph replay) and one iteration at a time.  A launch that is bound by vector-ALU issue pays N / (its instructions per wave)
# more time; one bound by latency at its occupancy pays less.  Instructions per wave and launch of the product: profiles/r5_pmc.json (1 094 / 1 226).
reps=${1:-3}
fl="exact fma"
for n in 160 320 640; do [ -f project3-cuda-path-tracer-1_amd/csrc/libptk_pad$n.so ] && fl="$fl pad$n"; done
echo "# timed configuration (3 in flight, graph replay): flavour, Mrays/s, ms per iteration"
bash tools/diag/ab_flavours.sh "$fl" "0" $reps
echo "# one iteration at a time"
bash tools/diag/ab_flavours.sh "$fl" "0" $reps --in-flight 1 --graph 0
# the mesh walk: libptk_walk<N>.so = N more fma per NODE VISIT of k_traverse_wide (-DPTK_DIAG_PAD_WALK=<N>; ~150 vector instructions per visit of a pair in the product)
wl="exact"
for n in 32 64; do [ -f project3-cuda-path-tracer-1_amd/csrc/libptk_walk$n.so ] && wl="$wl walk$n"; done
echo "# bunny_cornell.txt, timed configuration"
bash tools/diag/ab_flavours.sh "$wl" "0" $reps --scene bunny_cornell.txt
echo "# bunny_cornell.txt, one iteration at a time"
bash tools/diag/ab_flavours.sh "$wl" "0" $reps --scene bunny_cornell.txt --in-flight 1 --graph 0
