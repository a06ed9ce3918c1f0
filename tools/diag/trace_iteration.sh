#!/bin/bash
# GPU box: rocprofv3 kernel trace of a few single-stream iterations of a scene; prints the launches of the last complete iteration.
# usage: bash tools/diag/trace_iteration.sh <tag> <bench args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/trace_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $out -o trace -- python3 bench.py --lean 1 --cpu-seconds 0 --bvh-scene "" --steps 1 --warmup 1 --iters-per-step 8 --in-flight 1 --graph 0 "$@" > $out/bench.json 2> $out/bench.err || { tail -5 $out/bench.err; exit 1; }
ls -R $out | head; tail -2 $out/bench.err
python3 - "$out" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# iterations start with a GEN launch (k_shade_trace<*, true, ...> or k_generate)
import re
gen = [i for i, r in enumerate(rows) if "k_generate" in r["Kernel_Name"] or re.search(r"k_shade_trace<\d, true", r["Kernel_Name"])]
print(len(rows), "launches,", len(gen), "iterations")
a, b = gen[10], gen[11]          # the 11th of the 16 iterations of the timed configuration (the profile passes follow)
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    print("%8.1f +%7.1f us  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, re.sub(r"^void \(anonymous namespace\)::", "", r["Kernel_Name"])[:48]))
print("iteration: %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
PY
rm -rf $out
