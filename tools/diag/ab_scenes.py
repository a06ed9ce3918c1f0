"""A/B on the GPU box: Grays/s of (scene, mode, res, depth, lanes) cases for several library flavours (PTK_FLAVOUR; libptk_<name>.so
from tools/diag/build_variant.sh), alternating runs, one child process per measurement.
usage: ab_scenes.py "<flavours>" "<case>;<case>..." [reps]   case = scene,mode[,lanes[,pipeline[,W,H[,depth]]]]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %r)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
a = sys.argv[1].split(",")
scn, mode = a[0], int(a[1]); lanes = int(a[2]) if len(a) > 2 else 3; pipe = int(a[3]) if len(a) > 3 else 0
res = (int(a[4]), int(a[5])) if len(a) > 5 else None; depth = int(a[6]) if len(a) > 6 else -1
s = ptk.Scene(os.path.join(%r, "scenes", scn), res=res, depth=depth)
r = ptk.Renderer(s, mode, pipeline=pipe, in_flight=lanes, graph=True)
n = 96
for it in range(1, 49): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(49, 49 + n): r.iterate(it)
ms = r.timer_stop()
print("%%.3f %%.4f" %% ((r.ray_total() - r0) / ms / 1e6, ms / n))
r.close(); s.close()
''' % (ROOT, ROOT)
fl = sys.argv[1].split(); cases = sys.argv[2].split(";"); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {}
for rep in range(reps):
    for c in cases:
        for f in fl:
            env = dict(os.environ); env["PTK_FLAVOUR"] = "" if f == "exact" else f
            o = subprocess.run([sys.executable, "-c", CHILD, c], env=env, capture_output=True, text=True)
            if o.returncode: print("FAILED", f, c, o.stderr[-400:], flush=True); continue
            g = float(o.stdout.split()[0]); res.setdefault((c, f), []).append(g)
            print(f"{c:50s} {f:10s} {g:8.3f} Grays/s", flush=True)
print("---- best / median")
for (c, f), v in res.items():
    v = sorted(v); print(f"{c:50s} {f:10s} best {v[-1]:8.3f}  median {v[len(v) // 2]:8.3f}")
