"""Aggregates the FETCH_SIZE / WRITE_SIZE passes of tools/refresh_profiles.sh into the per-launch HBM traffic of the intersect
kernel (bench.py's roofline.traffic).  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide
coalesced read (MI355X_MICROARCH.md, HBM / rocprofv3 section), hence the doubling."""
import collections
import csv
import glob
import json
import os
import sys


def per_dispatch(path, kernel_prefix):
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None, 0
    tot = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if kernel_prefix in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    if not tot:
        return None, 0
    k = next(iter(tot))
    return tot[k] / n[k], n[k]


def main(out):
    res = {"_how": "rocprofv3 --pmc <counter> --kernel-trace -- python3 bench.py --scene S --steps 4 --warmup 1 --in-flight 1 --graph 0; one counter per pass "
                   "(tools/refresh_profiles.sh); FETCH_SIZE/WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md); averages per dispatch of the "
                   "kernel named, all bounces", "scenes": {}}
    for scene, tag in (("cornell.txt", "cornell"), ("bunny_cornell.txt", "bunny_cornell")):
        entry = {}
        for kern, key in (("k_intersect<false, false>", "k_intersect"), ("k_intersect<false, true>", "k_intersect_split"), ("k_traverse<false>", "k_traverse"),
                          ("k_shade_binned", "k_shade_binned"), ("k_segments", "k_segments")):
            fe, n = per_dispatch(os.path.join(out, "pmc_FETCH_SIZE_" + tag), kern)
            wr, _ = per_dispatch(os.path.join(out, "pmc_WRITE_SIZE_" + tag), kern)
            if fe is None or wr is None:
                continue
            entry[key] = {"dispatches": n, "FETCH_SIZE_KiB_raw": round(fe, 1), "WRITE_SIZE_KiB": round(wr, 1),
                          "traffic_bytes_per_launch": int(round(2 * fe * 1024 + wr * 1024))}
        if entry:
            main_k = "k_intersect" if "k_intersect" in entry else "k_intersect_split"
            entry["kernel"] = main_k
            entry["traffic_bytes_per_launch"] = entry[main_k]["traffic_bytes_per_launch"] + (entry.get("k_traverse", {}).get("traffic_bytes_per_launch", 0) if main_k == "k_intersect_split" else 0)
            res["scenes"][scene] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
