"""tools/diag/issue_sensitivity.sh's output -> profiles/<round>_issue_sensitivity.json: what N more vector instructions per wave and launch cost.
usage: issue_sensitivity_sum.py <out.txt> <pmc.json> <out.json>"""
import json
import re
import sys
from collections import defaultdict

txt, pmc, dst = sys.argv[1:4]
sect = None
ms = {"timed": defaultdict(list), "one_at_a_time": defaultdict(list), "bunny_timed": defaultdict(list), "bunny_one_at_a_time": defaultdict(list)}
for l in open(txt):
    if l.startswith("# timed"): sect = "timed"
    elif l.startswith("# one iteration"): sect = "one_at_a_time"
    elif l.startswith("# bunny_cornell.txt, timed"): sect = "bunny_timed"
    elif l.startswith("# bunny_cornell.txt, one"): sect = "bunny_one_at_a_time"
    m = re.match(r"flavour (\S+) pipeline 0 (\S+) (\S+)", l)
    if m and sect: ms[sect][m.group(1)].append(float(m.group(3)))
rec = json.load(open(pmc))["scenes"]["cornell.txt"]
vi = rec["k_shade_trace"]["counters"]["SQ_INSTS_VALU"] * 8 + rec["k_shade_trace_gen"]["counters"]["SQ_INSTS_VALU"]
waves = 3873589 / 64.0            # traced rays of one iteration (bench line: rays_per_step / 16) / 64 = waves that run a trace phase, where the padding sits
out = {"what": "N more v_fma_f32 per wave and launch in k_shade_trace's trace phase (four independent chains on live registers, no memory): diagnostic builds "
               "libptk_pad<N>.so, alternating child processes of bench.py --lean 1, mean of the repetitions; fma = the contracted flavour (~5 % FEWER vector instructions)",
       "vector_instructions_per_iteration": int(vi), "padded_waves_per_iteration": int(waves),
       "cost_model": "an added fma occupies its SIMD for 1.3 ns (profiles/r3_issue_cost.txt, 2-8 waves per SIMD); issue-bound => added time == added issue time"}
for s, d in ms.items():
    if not d: continue
    base = sum(d["exact"]) / len(d["exact"])
    rows = {"exact": {"ms_per_iteration": round(base, 4), "runs": d["exact"]}}
    for f, v in d.items():
        if f == "exact": continue
        m_ = sum(v) / len(v)
        r = {"ms_per_iteration": round(m_, 4), "runs": v, "time": "%+.1f %%" % (100 * (m_ / base - 1))}
        if f.startswith("pad"):
            n = int(f[3:])
            added = waves * n
            issue_us = added / 1024 * 1.3e-3
            r.update({"instructions": "%+.1f %%" % (100 * added / vi), "added_issue_time_us": round(issue_us, 1), "added_time_us": round(1e3 * (m_ - base), 1),
                      "paid_share_of_issue_time": round(1e3 * (m_ - base) / issue_us, 2)})
        if f.startswith("walk"):
            # the mesh walk: N more fma per NODE VISIT of k_traverse_wide; per listed ray 6.76 node visits x ~150 + 2.37 leaf visits x ~90 vector instructions
            # (profiles/r5_wide_tree_model.txt, llvm listing), the walk ~0.7 of the iteration's vector instructions (profiles/r5_pmc.json)
            n = int(f[4:])
            r.update({"walk_instructions": "%+.1f %%" % (100 * 6.76 * n / (6.76 * 150 + 2.37 * 90)), "added_chain_per_trip": "%d fma x ~4.5 cycles = %+.0f %% of a ~1 900-cycle trip" % (n, 100 * n * 4.5 / 1900)})
        rows[f] = r
    if d: out[s] = rows
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
