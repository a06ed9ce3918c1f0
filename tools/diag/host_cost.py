"""GPU box: host-side cost of a ptk_pathtrace call (enqueue only) against the GPU time per iteration, for the two accumulation policies.
usage: host_cost.py [scene,mode,lanes[,W,H,depth]] ..."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
for case in sys.argv[1:] or ["cornell.txt,0,3"]:
    a = case.split(",")
    res = (int(a[3]), int(a[4])) if len(a) > 4 else None; depth = int(a[5]) if len(a) > 5 else -1
    s = ptk.Scene(os.path.join(ROOT, "scenes", a[0]), res=res, depth=depth)
    for ordered in (0, 1, 0, 1):
        r = ptk.Renderer(s, int(a[1]), in_flight=int(a[2]), graph=int(a[2]) > 1, ordered=ordered)
        for it in range(1, 65): r.iterate(it)
        r.sync()
        n = 256
        t0 = time.perf_counter()
        for it in range(65, 65 + n): r.iterate(it)
        t1 = time.perf_counter()
        r.sync()
        t2 = time.perf_counter()
        print(f"{case:44s} ordered {ordered}: enqueue {1e6 * (t1 - t0) / n:7.1f} us per call, total {1e6 * (t2 - t0) / n:7.1f} us per iteration", flush=True)
        r.close()
    s.close()
