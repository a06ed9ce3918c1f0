"""Summarises the passes of tools/pmc_wait.sh (per-kernel per-dispatch means printed by tools/pmc_sum.py) into one JSON: for every kernel of interest,
the wave-time split with the wait attributed — LDS instructions, vector memory in flight, the rest (barriers, scalar loads, export)."""
import json, os, re, sys
d = sys.argv[1]
# (tools/pmc_sum.py cuts kernel names at 48 characters: "void (anonymous namespace)::k_shade_trace<0, fal")
KERNELS = {"k_shade_trace<0, f": "k_shade_trace", "k_shade_trace<0, t": "k_shade_trace_gen", "k_trace_shade<0, f": "k_trace_shade", "k_trace_shade<0, t": "k_trace_shade_gen",
           "k_traverse_wide": "k_traverse_wide", "k_traverse_pool": "k_traverse_pool", "k_intersect<false, t": "k_intersect_split", "k_intersect<false, f": "k_intersect",
           "k_shade_binned": "k_shade_binned", "k_merge_planes": "k_merge_planes"}
out = {"_how": "tools/pmc_wait.sh: rocprofv3 --pmc passes (with --kernel-trace only) of `bench.py --lean 1 --in-flight 1 --graph 0 --steps 3 --warmup 1`, per-dispatch means per "
               "kernel.  SQ_* cycle counters are in quad-cycles summed over waves; SQ_INST_LEVEL_x / SQ_INSTS_x = mean latency (cycles x 4) of an x instruction; "
               "SQ_WAIT_ANY = wave-cycles waiting for anything; SQ_WAIT_INST_ANY = wave-cycles waiting to ISSUE an instruction (arbitration, dependencies; SQ_WAIT_INST_LDS: its LDS part); "
               "their difference = waiting on DATA (s_waitcnt) or at a BARRIER — no counter separates those two, the summary prices the data part from the dependent round trips "
               "(mean_l1_to_l2_read_latency_cycles x the kernel's dependent-load depth) and calls the rest barrier", "scenes": {}}
for f in sorted(os.listdir(d)):
    m = re.match(r"(.+)_(SQ_A|SQ_B|TA|L1|L2)\.txt$", f)
    if not m: continue
    scene, cur = m.group(1) + ".txt", None
    for line in open(os.path.join(d, f)):
        if not line.startswith(" "):
            name = line.strip().replace("void ", "").replace("(anonymous namespace)::", "")
            cur = next((v for k, v in KERNELS.items() if name.startswith(k)), None)
            continue
        if cur is None: continue
        t = line.split()
        out["scenes"].setdefault(scene, {}).setdefault(cur, {})[t[0]] = float(t[4])
for scene, ks in out["scenes"].items():
    for k, c in ks.items():
        wc = c.get("SQ_WAVE_CYCLES")
        if wc and c.get("SQ_WAIT_ANY"):
            w = c["SQ_WAIT_ANY"]
            c["wait_share_of_wave_cycles"] = round(w / wc, 3)
            if "SQ_WAIT_INST_LDS" in c: c["waiting_to_issue_lds_share_of_wave_cycles"] = round(c["SQ_WAIT_INST_LDS"] / wc, 4)
            if "SQ_WAIT_INST_ANY" in c:
                c["waiting_to_issue_share_of_wave_cycles"] = round(c["SQ_WAIT_INST_ANY"] / wc, 3)
                c["waiting_on_data_or_barrier_share_of_wave_cycles"] = round((w - c["SQ_WAIT_INST_ANY"]) / wc, 3)
            if c.get("SQ_WAVES"): c["wave_lifetime_cycles"] = round(4 * wc / c["SQ_WAVES"])
        if wc and c.get("SQ_INST_LEVEL_VMEM"): c["vmem_in_flight_share_of_wave_cycles"] = round(c["SQ_INST_LEVEL_VMEM"] / wc, 3)
        if wc and c.get("SQ_INST_LEVEL_LDS"): c["lds_in_flight_share_of_wave_cycles"] = round(c["SQ_INST_LEVEL_LDS"] / wc, 3)
        n_vm = (c.get("SQ_INSTS_VMEM_RD") or 0) + (c.get("SQ_INSTS_VMEM_WR") or 0)
        if n_vm and c.get("SQ_INST_LEVEL_VMEM"): c["mean_vmem_latency_cycles"] = round(4 * c["SQ_INST_LEVEL_VMEM"] / n_vm, 1)
        if c.get("SQ_INSTS_LDS") and c.get("SQ_INST_LEVEL_LDS"): c["mean_lds_latency_cycles"] = round(4 * c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"], 1)
        if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TCC_READ_REQ_LATENCY_sum"): c["mean_l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0: c["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
print(json.dumps(out, indent=1))
