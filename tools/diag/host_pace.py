"""GPU box: is the render loop paced by the host?  Steady state: 1 024 calls after 64 warm-up calls; prints Grays/s, the time per iteration and the host loop's time per
call; with PTK_HOST_TIMING=1 the library adds where the host time goes (hipGraphLaunch / the merge's runtime calls / the rest) when the context is freed.
usage: [PTK_HOST_TIMING=1] host_pace.py scene,mode,lanes[,graph[,W,H[,depth]]] ..."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
for case in sys.argv[1:]:
    a = case.split(","); L = int(a[2]); graph = int(a[3]) if len(a) > 3 else 1
    res = (int(a[4]), int(a[5])) if len(a) > 5 else None; depth = int(a[6]) if len(a) > 6 else -1
    s = ptk.Scene(os.path.join(ROOT, "scenes", a[0]), res=res, depth=depth)
    r = ptk.Renderer(s, int(a[1]), in_flight=L, graph=bool(graph))
    it = 1
    for _ in range(64): r.iterate(it); it += 1
    r.sync(); n = 1024
    r0 = r.ray_total(); r.timer_start(); t0 = time.perf_counter()
    for _ in range(n): r.iterate(it); it += 1
    t1 = time.perf_counter(); ms = r.timer_stop()
    print(f"{case:40s} {(r.ray_total() - r0) / ms / 1e6:8.3f} Grays/s  {1e3 * ms / n:7.1f} us per iteration; host loop {1e6 * (t1 - t0) / n:7.1f} us per call", flush=True)
    r.close(); s.close()
