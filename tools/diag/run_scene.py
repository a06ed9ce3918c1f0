"""GPU helper for profiling: renders N iterations of a scene (default bunny_cornell.txt) with one pipeline.
usage: run_scene.py [scene] [pipeline] [iters] [mode] [in_flight]"""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
a = sys.argv[1:]
scn = os.path.join(ROOT, "scenes", a[0] if len(a) > 0 else "bunny_cornell.txt")
pipe = int(a[1]) if len(a) > 1 else 0
iters = int(a[2]) if len(a) > 2 else 10
mode = int(a[3]) if len(a) > 3 else 0
lanes = int(a[4]) if len(a) > 4 else 1
s = ptk.Scene(scn)
r = ptk.Renderer(s, mode, pipeline=pipe, in_flight=lanes, graph=lanes > 1)
for it in range(1, 4): r.iterate(it)
r.sync()
r0 = r.ray_total()
r.timer_start()
for it in range(4, 4 + iters): r.iterate(it)
ms = r.timer_stop()
print(f"{os.path.basename(scn)} pipeline {pipe} lanes {lanes}: {(r.ray_total() - r0) / ms / 1e6:.2f} Grays/s, {ms / iters:.3f} ms/iter")
r.close(); s.close()
