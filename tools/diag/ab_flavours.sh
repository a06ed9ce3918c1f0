#!/bin/bash
# usage (GPU box): bash tools/diag/ab_flavours.sh "<flavour list>" "<pipeline list>" [reps] [extra bench args]: alternating runs, headline + single stream
fl=$1; pl=$2; reps=${3:-3}; shift; shift; shift
for r in $(seq $reps); do for f in $fl; do for p in $pl; do
  PTK_FLAVOUR=$f python3 bench.py --steps 20 --warmup 5 --lean 1 --profile-steps 1 --pipeline $p "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('flavour', sys.argv[1], 'pipeline', sys.argv[2], d['value'], d['ms_per_iteration'], d.get('flavour'))" $f $p
done; done; done
