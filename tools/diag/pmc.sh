#!/bin/bash
# usage: pmc.sh <tag> <counters...> -- <run_scene args>;  separate rocprofv3 --pmc pass (kernel-trace only), summary -> gpurun_out/pmc_<tag>.txt
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --pmc "${ctrs[@]}" -d $out -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/diag/run_scene.py "$@" > $out/run.log 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
f = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
if not f: print("no counter csv", glob.glob(out + "/**/*", recursive=True)); sys.exit(0)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    m = re.search(r"(k_\w+(<[^>]*>)?)", row["Kernel_Name"]); k = m.group(1) if m else row["Kernel_Name"][:40]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as fo:
    for k in agg:
        for c, v in agg[k].items():
            line = f"{k:42s} {c:28s} launches {cnt[(k, c)]:5d}  per-launch {v / cnt[(k, c)]:16.1f}"
            print(line); fo.write(line + "\n")
PY
