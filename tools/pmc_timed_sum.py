"""Sums a rocprofv3 --pmc pass over EVERY dispatch of `bench.py --lean 1` (the timed configuration: iterations in flight + graph replay) into the
utilisation figure bench.py prints as `valu_busy_timed` (tools/pmc_timed.sh).  No assumed clock: the cycles the GPU was active come from
GRBM_GUI_ACTIVE of the same dispatches (rocprofv3 reports its sum over the 8 XCDs, MI355X_MICROARCH.md); SQ_ACTIVE_INST_VALU counts
quad-cycles (same guide).  frac = 4 x sum(SQ_ACTIVE_INST_VALU) / (1024 SIMDs x sum(GRBM_GUI_ACTIVE) / 8) <= 1 by construction.
Note: rocprofv3 serialises dispatches while it collects counters, so the denominator is the kernels' own active time back to back, not the
overlapped wall time — the figure is the vector-ALU share of the time the kernels occupy the GPU."""
import collections, csv, json, sys
tot = collections.Counter(); disp = set(); dur_ns = 0; seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    tot[r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r.get("Dispatch_Id"), r.get("Correlation_Id"))
    disp.add(key)
    if key not in seen:
        seen.add(key); dur_ns += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
bench = {}
try:
    bench = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
except Exception:
    pass
simds = 1024
gui = tot.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
valu = 4.0 * tot.get("SQ_ACTIVE_INST_VALU", 0.0)
out = {"frac": round(valu / (simds * gui), 4) if gui else None, "valu_active_cycles": round(valu), "gpu_active_cycles_per_xcd": round(gui), "simds": simds,
       "effective_clock_GHz": round(gui / dur_ns, 3) if dur_ns else None, "dispatches": len(disp), "kernel_time_ms": round(dur_ns / 1e6, 3),
       "SQ_INSTS_VALU": round(tot.get("SQ_INSTS_VALU", 0)), "SQ_WAVES": round(tot.get("SQ_WAVES", 0)), "SQ_BUSY_CYCLES": round(tot.get("SQ_BUSY_CYCLES", 0)),
       "bench_value_under_pmc": bench.get("value"), "bench_config": bench.get("config", {}).get("workload"),
       "note": "all dispatches of `bench.py --lean 1` (the default: four iterations in flight, graph replay) under rocprofv3 --pmc; dispatches are serialised while counters are collected, "
               "so this is the vector-ALU share of the kernels' own GPU-active cycles (GRBM_GUI_ACTIVE / 8 XCDs), not of the overlapped wall time",
       "source": "tools/pmc_timed.sh"}
print(json.dumps(out, indent=1))
