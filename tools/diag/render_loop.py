"""Renders N iterations of a scene back to back (nothing else): the process to put under rocprofv3 for timelines / counters of the steady state.
usage: render_loop.py scene mode lanes iters [W H [depth [pipeline]]]   (W = 0: the scene's own size)      prints Grays/s and ms per iteration of the second half"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
scn, mode, lanes, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
res = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 and int(sys.argv[5]) > 0 else None
depth = int(sys.argv[7]) if len(sys.argv) > 7 else -1
s = ptk.Scene(os.path.join(ROOT, "scenes", scn), res=res, depth=depth)
r = ptk.Renderer(s, mode, in_flight=lanes, graph=lanes > 1, pipeline=int(sys.argv[8]) if len(sys.argv) > 8 else 0)
h = iters // 2
for it in range(1, h + 1): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(h + 1, iters + 1): r.iterate(it)
ms = r.timer_stop()
print("%.3f Grays/s %.4f ms/iteration" % ((r.ray_total() - r0) / ms / 1e6, ms / (iters - h)))
r.close(); s.close()
