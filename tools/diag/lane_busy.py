"""GPU diagnostic: busy lane share of the wide BVH walk (run with PTK_WIDE_STATS=2; PTK_WIDE_LPR=2|4; the sorted work list it was written
to judge — PTK_LIST_SORT — was measured and removed in round 4, profiles/r4_lane_busy_sorted_list.txt).
k_traverse_wide<STATS> adds, per wave, the visits of its slowest ray (what the wave runs for) and the visits of all its rays:
busy = all visits / (sum of slowest x rays per wave).   usage: lane_busy.py [scene] [iterations]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
scn = os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "bunny_cornell.txt")
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lpr = int(os.environ.get("PTK_WIDE_LPR", "4"))
s = ptk.Scene(scn)
r = ptk.Renderer(s, 0, pipeline=8)           # staged kernels (the traversal counters live there)
for it in range(1, 3): r.iterate(it)
r.sync()
r.traversal_stats(True)
r0 = r.ray_total()
for it in range(3, 3 + iters): r.iterate(it)
r.sync()
slow, allv = r.traversal_stats()
n = r.ray_total() - r0
G = 64 // lpr
print(f"lpr={lpr}: {n} rays, visits of listed rays {allv}, sum over waves of the slowest ray {slow} "
      f"-> busy lane share {allv / max(slow * G, 1):.3f} (rays per wave {G}); wave-visits per traced ray {slow / n:.4f}")
r.close(); s.close()
