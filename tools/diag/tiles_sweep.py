"""GPU box: Grays/s of tiles in flight (K bands, graph replay) against one iteration at a time and against iterations in flight.
usage: tiles_sweep.py "<scene,mode[,W,H[,depth]]>;..." "<K,K,...>" """
import importlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %r)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
a = sys.argv[1].split(","); kind, k = sys.argv[2], int(sys.argv[3])
scn, mode = a[0], int(a[1]); res = (int(a[2]), int(a[3])) if len(a) > 3 else None; depth = int(a[4]) if len(a) > 4 else -1
s = ptk.Scene(os.path.join(%r, "scenes", scn), res=res, depth=depth)
kw = dict(tiles=k, graph=True) if kind == "tiles" else (dict(in_flight=k, graph=k > 1) if kind == "iters" else dict())
r = ptk.Renderer(s, mode, **kw)
n = 128 if s.width * s.height <= 1000000 else 32
for it in range(1, 41): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(41, 41 + n): r.iterate(it)
ms = r.timer_stop()
print("%%.3f %%.4f" %% ((r.ray_total() - r0) / ms / 1e6, ms / n))
r.close(); s.close()
''' % (ROOT, ROOT)
cases = sys.argv[1].split(";"); ks = [int(x) for x in sys.argv[2].split(",")]
for c in cases:
    row = []
    for kind, k in [("iters", 1), ("iters", 3)] + [("tiles", k) for k in ks]:
        best = 0.0
        for rep in range(2):
            o = subprocess.run([sys.executable, "-c", CHILD, c, kind, str(k)], capture_output=True, text=True)
            if o.returncode: print("FAILED", c, kind, k, o.stderr[-300:], flush=True); break
            best = max(best, float(o.stdout.split()[0]))
        row.append(f"{kind}{k} {best:.2f}")
    print(f"{c:40s} " + "  ".join(row), flush=True)
