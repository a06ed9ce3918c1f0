"""GPU diagnostic: bunny scene, per-kernel device time of the intersect stage for the wide (pipeline 0) and the threaded
(pipeline 5) walk, and the wide walk's visit counts (PTK_WIDE_STATS=1)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
scn = os.path.join(ROOT, "scenes", sys.argv[1] if len(sys.argv) > 1 else "bunny_cornell.txt")
for pipe in (0, 5):
    s = ptk.Scene(scn)
    r = ptk.Renderer(s, 0, pipeline=pipe)
    for it in range(1, 4): r.iterate(it)
    r.sync()
    r.profile_begin()
    for it in range(4, 8): r.iterate(it)
    prof = r.profile_end()
    rays0 = r.ray_total()
    r.timer_start()
    for it in range(8, 28): r.iterate(it)
    ms = r.timer_stop()
    rays = r.ray_total() - rays0
    print(f"pipeline {pipe}: {rays / ms / 1e6:.2f} Grays/s, {ms / 20:.3f} ms/iter; per-iter kernel ms: " + ", ".join(f"{k} {v[0] / 4:.3f}" for k, v in prof.items()))
    if os.environ.get("PTK_WIDE_STATS") or pipe == 5:
        r.traversal_stats(True)
        r0 = r.ray_total()
        r.iterate(28); r.sync()
        a, b = r.traversal_stats()
        n = r.ray_total() - r0
        print(f"   stats per ray ({n} rays): {a / n:.3f} node visits, {b / n:.3f} leaf visits / tri tests")
    r.close(); s.close()
