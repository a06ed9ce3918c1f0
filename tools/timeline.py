"""What the GPU does over time while several iterations are in flight: from a `rocprofv3 --kernel-trace --output-format csv` trace
(…_kernel_trace.csv: one row per dispatch with Start_Timestamp / End_Timestamp / Queue_Id / Kernel_Name) print
  * wall time of the traced window, the union of all kernel intervals (GPU busy), the sum of kernel durations and their ratio (mean concurrency);
  * per queue: dispatches, busy time, and the gaps between consecutive dispatches (mean / median / total);
  * per kernel: calls, mean duration, share of the summed durations.
usage: timeline.py <kernel_trace.csv> [from=0.55 [to=0.95]]   (fractions of the traced window: warm-up, graph capture and the drain at the end are dropped)"""
import csv
import sys
from collections import defaultdict

import numpy as np

path = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
stop = float(sys.argv[3]) if len(sys.argv) > 3 else 0.95
rows = []
for r in csv.DictReader(open(path)):
    name = r.get("Kernel_Name") or r.get("Name")
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), name))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut, end = t0 + skip * (t1 - t0), t0 + stop * (t1 - t0)
rows = [r for r in rows if r[0] >= cut and r[1] <= end]
t0, t1 = rows[0][0], max(r[1] for r in rows)
wall = (t1 - t0) / 1e3
# union of intervals
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = sum(e - s for s, e, _, _ in rows)
print(f"window {wall:.0f} us, {len(rows)} dispatches; GPU busy (union of kernel intervals) {busy / 1e3:.0f} us = {busy / 1e3 / wall:.3f}; "
      f"sum of kernel durations {total / 1e3:.0f} us = {total / max(busy, 1):.2f} kernels running on average while busy")
# time with exactly k kernels running
ev = []
for s, e, _, _ in rows: ev.append((s, 1)); ev.append((e, -1))
ev.sort()
lvl = 0; last = ev[0][0]; hist = defaultdict(int)
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
print("time with k kernels running: " + ", ".join(f"{k}: {v / 1e3 / wall:.3f}" for k, v in sorted(hist.items())))
byq = defaultdict(list)
for s, e, q, n in rows: byq[q].append((s, e, n))
for q, v in sorted(byq.items()):
    if len(v) < 3: continue
    gaps = np.array([v[i + 1][0] - v[i][1] for i in range(len(v) - 1)], np.float64) / 1e3
    b = sum(e - s for s, e, _ in v) / 1e3
    print(f"queue {q}: {len(v)} dispatches, busy {b:.0f} us ({b / wall:.3f}), gaps mean {gaps.mean():.2f} median {np.median(gaps):.2f} p90 {np.percentile(gaps, 90):.2f} us, "
          f"total {gaps[gaps > 0].sum():.0f} us ({gaps[gaps > 0].sum() / wall:.3f})")
# which dispatch the queue was idle BEFORE (round 5, second session): gap to the previous dispatch on the same queue, by kernel
def short(n): return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
gk = defaultdict(list)
for q, v in byq.items():
    for i in range(1, len(v)): gk[short(v[i][2])].append(max(0.0, (v[i][0] - v[i - 1][1]) / 1e3))
print("idle time of a queue before a dispatch, by kernel (us): calls, mean, median, p90, share of all idle time")
tot_gap = sum(sum(v) for v in gk.values())
for n, v in sorted(gk.items(), key=lambda kv: -sum(kv[1])):
    a = np.array(v)
    print(f"  {n:60s} {len(v):6d} {a.mean():8.2f} {np.median(a):8.2f} {np.percentile(a, 90):8.2f}   {a.sum() / max(tot_gap, 1e-9):.3f}")
byk = defaultdict(list)
for s, e, _, n in rows: byk[n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]].append((e - s) / 1e3)
for n, v in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {n:60s} calls {len(v):6d} mean {np.mean(v):8.2f} us  share {sum(v) * 1e3 / total:.3f}")
if len(sys.argv) > 4:                   # timeline.py trace from to dump_us: the dispatches of the first dump_us microseconds of the window, per queue
    lim = t0 + float(sys.argv[4]) * 1e3
    print("---- dispatches (queue, start us, duration us, gap to the previous dispatch on the queue, kernel)")
    last_end = {}
    for s, e, q, n in rows:
        if s > lim: break
        g = (s - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = e
        print(f"q{q} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {g:8.1f}  {n.replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0][:40]}")
