"""Summarises the passes of tools/pmc_wait.sh (per-kernel per-dispatch means printed by tools/pmc_sum.py) into one JSON: for every kernel of interest,
the wave-time split with the wait attributed — LDS instructions, vector memory in flight, the rest (barriers, scalar loads, export)."""
import json, os, re, sys
d = sys.argv[1]
# (tools/pmc_sum.py cuts kernel names at 48 characters: "void (anonymous namespace)::k_shade_trace<0, fal")
KERNELS = {"k_shade_trace<0, f": "k_shade_trace", "k_shade_trace<0, t": "k_shade_trace_gen", "k_trace_shade<0, f": "k_trace_shade", "k_trace_shade<0, t": "k_trace_shade_gen",
           "k_traverse_wide": "k_traverse_wide", "k_traverse_pool": "k_traverse_pool", "k_intersect<false, t": "k_intersect_split", "k_intersect<false, f": "k_intersect",
           "k_shade_binned": "k_shade_binned", "k_merge_planes": "k_merge_planes"}
out = {"_how": "tools/pmc_wait.sh: rocprofv3 --pmc passes (with --kernel-trace only) of `bench.py --lean 1 --in-flight 1 --graph 0 --steps 3 --warmup 1`, per-dispatch means per "
               "kernel.  SQ_* cycle counters are in quad-cycles summed over waves.  A wave's cycles split three ways (the three counters add up to SQ_WAVE_CYCLES within 0.1 %): "
               "SQ_ACTIVE_INST_ANY issuing (profiles/<round>_pmc.json), SQ_WAIT_INST_ANY waiting to ISSUE (arbitration, dependencies; SQ_WAIT_INST_LDS: its LDS part), "
               "SQ_WAIT_ANY waiting on DATA (s_waitcnt) or at a BARRIER.  No counter separates data from barrier; the camera launch of the fused kernel — the same trace phase and "
               "barriers, no loads at its head — is the control: what the shading launches wait beyond it is their load chain.  TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ = "
               "mean L1 -> L2 read latency in cycles; TA_BUSY_avr = texture-addresser busy cycles per dispatch (compare launch_us x clock)", "scenes": {}}
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
        if wc and c.get("SQ_WAIT_ANY") is not None:
            c["waiting_on_data_or_barrier_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / wc, 3)
            if "SQ_WAIT_INST_ANY" in c: c["waiting_to_issue_share_of_wave_cycles"] = round(c["SQ_WAIT_INST_ANY"] / wc, 3)
            if "SQ_WAIT_INST_LDS" in c: c["waiting_to_issue_lds_share_of_wave_cycles"] = round(c["SQ_WAIT_INST_LDS"] / wc, 4)
        if c.get("TCP_TCC_READ_REQ_sum") and c.get("TCP_TCC_READ_REQ_LATENCY_sum"): c["mean_l1_to_l2_read_latency_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0: c["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
print(json.dumps(out, indent=1))
