"""Aggregates the PMC passes of tools/refresh_profiles.sh into profiles/r2_pmc.json: per-launch HBM traffic of the kernels of
the intersect stage (bench.py's roofline.traffic / achieved_hbm) and the SQ counter table (VALU instructions and busy
fraction, wave occupancy, LDS) bench.py cites as roofline.counters.
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read (MI355X_MICROARCH.md, HBM /
rocprofv3 section), hence the doubling.  SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* count quad-cycles (same guide)."""
import collections
import csv
import glob
import json
import os
import re
import sys

KERNELS = (("k_intersect<false, false>", "k_intersect"), ("k_intersect<false, true>", "k_intersect_split"), ("k_list", "k_list"),
           ("k_traverse_wide<false, 4>", "k_traverse_wide"), ("k_traverse<false>", "k_traverse"), ("k_finalize", "k_finalize"),
           ("k_shade_binned", "k_shade_binned"), ("k_generate", "k_generate"))


def per_dispatch(path, kernel_prefix):
    """-> ({counter: mean per dispatch}, dispatches, mean duration in us)"""
    f = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}, 0, 0.0
    tot = collections.defaultdict(float); n = collections.Counter(); dur = 0.0
    for r in csv.DictReader(open(f[0])):
        if kernel_prefix in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
            dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not tot:
        return {}, 0, 0.0
    k = next(iter(tot))
    return {c: v / n[c] for c, v in tot.items()}, n[k], dur / sum(n.values())


def main(out):
    res = {"_how": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --scene S --steps 4 --warmup 1 --in-flight 1 --graph 0 (tools/refresh_profiles.sh): "
                   "FETCH_SIZE and WRITE_SIZE one counter per pass, two SQ passes; no PMC pass is combined with a sys/hip/hsa trace.  FETCH_SIZE/WRITE_SIZE in KiB, "
                   "FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md); SQ_*_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_* in quad-cycles.  Means per dispatch of the kernel "
                   "named, over all bounces (launches past the live count included).", "scenes": {}}
    for scene, tag in (("cornell.txt", "cornell"), ("bunny_cornell.txt", "bunny_cornell")):
        entry = {}
        for kern, key in KERNELS:
            fe, n, _ = per_dispatch(os.path.join(out, "pmc_FETCH_SIZE_" + tag), kern)
            wr, _, _ = per_dispatch(os.path.join(out, "pmc_WRITE_SIZE_" + tag), kern)
            if not fe or not wr:
                continue
            e = {"dispatches": n, "FETCH_SIZE_KiB_raw": round(fe["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(wr["WRITE_SIZE"], 1),
                 "traffic_bytes_per_launch": int(round(2 * fe["FETCH_SIZE"] * 1024 + wr["WRITE_SIZE"] * 1024))}
            sq = {}
            for p in ("SQ1", "SQ2"):
                c, _, us = per_dispatch(os.path.join(out, "pmc_%s_%s" % (p, tag)), kern)
                sq.update({k: round(v) for k, v in c.items()})
                if us:
                    sq["launch_us_under_pmc"] = round(us, 2)
            if sq:
                w = sq.get("SQ_WAVE_CYCLES", 0)
                if w:
                    sq["wave_time_split"] = {"issuing": round(sq.get("SQ_ACTIVE_INST_ANY", 0) / w, 3), "waiting_on_data_or_barrier": round(sq.get("SQ_WAIT_ANY", 0) / w, 3),
                                             "waiting_to_issue": round(sq.get("SQ_WAIT_INST_ANY", 0) / w, 3)}
                if sq.get("SQ_WAVES") and sq.get("SQ_INSTS_VALU"):
                    sq["valu_per_launched_wave"] = round(sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"], 1)
                if sq.get("launch_us_under_pmc") and sq.get("SQ_ACTIVE_INST_VALU"):
                    # 1024 SIMDs; ACTIVE_INST_VALU in quad-cycles; clock unknown -> report busy cycles per SIMD and the fraction at 2.4 GHz
                    per_simd = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / 1024.0
                    sq["valu_busy_cycles_per_simd"] = round(per_simd)
                    sq["valu_busy_frac_at_2p4GHz"] = round(per_simd / (sq["launch_us_under_pmc"] * 2400.0), 3)
                e["counters"] = sq
            entry[key] = e
        if entry:
            stage = [k for k in ("k_intersect", "k_intersect_split", "k_list", "k_traverse_wide", "k_traverse", "k_finalize") if k in entry]
            entry["kernel"] = "+".join(stage)
            entry["traffic_bytes_per_launch"] = sum(entry[k]["traffic_bytes_per_launch"] for k in stage)
            main_k = "k_intersect" if "k_intersect" in entry else "k_intersect_split"
            entry["counters"] = dict(entry[main_k].get("counters", {}), kernel=main_k)
            res["scenes"][scene] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
