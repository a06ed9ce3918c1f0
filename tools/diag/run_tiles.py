"""GPU box: renders `iters` iterations of a scene with K tiles in flight (graph replay) — a plain workload for rocprofv3.   run_tiles.py scene mode K iters"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
scn, mode, k, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
s = ptk.Scene(os.path.join(ROOT, "scenes", scn))
r = ptk.Renderer(s, mode, tiles=k, graph=True) if k > 0 else ptk.Renderer(s, mode, in_flight=-k, graph=True)
for it in range(1, 17): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(17, 17 + iters): r.iterate(it)
ms = r.timer_stop()
print("%.3f Grays/s %.4f ms/iteration" % ((r.ray_total() - r0) / ms / 1e6, ms / iters))
r.close(); s.close()
