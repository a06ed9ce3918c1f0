"""GPU box: pure host cost of ptk_pathtrace (graph launch + merge + events) with EMPTY queues: after a sync, a burst of 2 calls per lane is timed on the
host; against it the steady-state time per iteration.  If the two are close the render loop is bound by the host's enqueue, not by the GPU.
usage: host_enqueue.py scene,mode,lanes[,W,H,depth] ..."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
for case in sys.argv[1:] or ["cornell.txt,0,3"]:
    a = case.split(",")
    res = (int(a[3]), int(a[4])) if len(a) > 4 else None; depth = int(a[5]) if len(a) > 5 else -1
    L = int(a[2])
    s = ptk.Scene(os.path.join(ROOT, "scenes", a[0]), res=res, depth=depth)
    r = ptk.Renderer(s, int(a[1]), in_flight=L, graph=L > 1)
    it = 1
    for _ in range(96): r.iterate(it); it += 1
    r.sync()
    burst = []
    for rep in range(20):
        t0 = time.perf_counter()
        for _ in range(2 * L): r.iterate(it); it += 1
        t1 = time.perf_counter()
        r.sync()
        burst.append(1e6 * (t1 - t0) / (2 * L))
    r.sync(); r.timer_start(); n = 192
    t0 = time.perf_counter()
    for _ in range(n): r.iterate(it); it += 1
    t1 = time.perf_counter()
    ms = r.timer_stop()
    burst.sort()
    print(f"{case:44s} host enqueue with empty queues: median {burst[len(burst) // 2]:6.1f} us per call (min {burst[0]:.1f}); steady state {1e3 * ms / n:7.1f} us per iteration "
          f"(host loop {1e6 * (t1 - t0) / n:.1f} us per call)", flush=True)
    r.close(); s.close()
