"""Aggregates the PMC passes of tools/refresh_profiles.sh into profiles/<round>_pmc.json: per-launch HBM traffic of the kernels of
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
           ("k_traverse_wide<false, 4>", "k_traverse_wide"), ("k_traverse<false>", "k_traverse"), ("k_finalize(", "k_finalize"),
           ("k_shade_binned", "k_shade_binned"), ("k_generate", "k_generate"),
           # the fused default pipeline (round 3)
           ("k_shade_trace<0, false, false", "k_shade_trace"), ("k_shade_trace<0, true, false", "k_shade_trace_gen"),
           ("k_shade_trace<0, false, true", "k_shade_trace_split"), ("k_shade_trace<0, true, true", "k_shade_trace_gen_split"),
           ("k_finalize2", "k_finalize2"))


def per_dispatch(path, kernel_prefix):
    """-> ({counter: mean per dispatch}, dispatches, mean duration in us)"""
    f = []
    for suffix in ("", "_p0", "_p8"):                      # one pass per pipeline: their kernels have different names
        f += glob.glob(os.path.join(path + suffix, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return {}, 0, 0.0
    tot = collections.defaultdict(float); n = collections.Counter(); dur = 0.0
    for ff in f:
        for r in csv.DictReader(open(ff)):
            if kernel_prefix in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
                dur += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if not tot:
        return {}, 0, 0.0
    k = next(iter(tot))
    return {c: v / n[c] for c, v in tot.items()}, n[k], dur / sum(n.values())


def main(out):
    res = {"_how": "rocprofv3 --pmc <counters> --kernel-trace -- python3 bench.py --scene S --steps 4 --warmup 1 --in-flight 1 --graph 0 --pipeline P (tools/refresh_profiles.sh; P = 0 the fused default, 8 the staged kernels): "
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
            # the counters cited for the stage are those of its DOMINANT kernel (longest mean launch)
            main_k = max(stage, key=lambda k: entry[k].get("counters", {}).get("launch_us_under_pmc", 0.0))
            entry["counters"] = dict(entry[main_k].get("counters", {}), kernel=main_k)
            if scene == "cornell.txt":                 # the contracted flavour's instruction counts beside the exact build's
                fma = {}
                for kern, key in KERNELS:
                    c, n, _ = per_dispatch(os.path.join(out, "pmc_SQ1_cornell_fma"), kern)
                    if c.get("SQ_INSTS_VALU") and c.get("SQ_WAVES") and key in entry and entry[key].get("counters", {}).get("SQ_INSTS_VALU"):
                        fma[key] = {"SQ_INSTS_VALU": round(c["SQ_INSTS_VALU"]), "SQ_INSTS_VALU_exact": entry[key]["counters"]["SQ_INSTS_VALU"],
                                    "valu_per_launched_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1),
                                    "valu_per_launched_wave_exact": entry[key]["counters"].get("valu_per_launched_wave"),
                                    "removed_by_contraction": round(1.0 - c["SQ_INSTS_VALU"] / entry[key]["counters"]["SQ_INSTS_VALU"], 4)}
                entry["contracted_fma_by_kernel"] = fma
                head = fma.get("k_shade_trace") or fma.get("k_intersect") or {}
                entry["contracted_fma"] = head
            res["scenes"][scene] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
