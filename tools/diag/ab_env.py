"""A/B on the GPU box over ENVIRONMENT settings (one child process per measurement, alternating runs): Grays/s of (scene, mode, lanes, ...)
cases for several `VAR=value,VAR=value` settings.
usage: ab_env.py "<setting>;<setting>..." "<case>;<case>..." [reps]   setting = "-" (nothing) or VAR=v[,VAR=v...];
       case = scene,mode[,lanes[,pipeline[,W,H[,depth[,tiles]]]]]  (W = 0: the scene's own size)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %r)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
a = sys.argv[1].split(",")
scn, mode = a[0], int(a[1]); lanes = int(a[2]) if len(a) > 2 else 3; pipe = int(a[3]) if len(a) > 3 else 0
res = (int(a[4]), int(a[5])) if len(a) > 5 and int(a[4]) > 0 else None; depth = int(a[6]) if len(a) > 6 else -1; tiles = int(a[7]) if len(a) > 7 else 1
s = ptk.Scene(os.path.join(%r, "scenes", scn), res=res, depth=depth)
r = ptk.Renderer(s, mode, pipeline=pipe, in_flight=lanes, graph=lanes > 1 or tiles > 1, tiles=tiles, ordered=int(os.environ.get('AB_ORDERED', '-1')))      # AB_ORDERED: ptk_set_ordered_accumulation
n = 96 if s.width * s.height <= 1000000 else 32
for it in range(1, 33): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(33, 33 + n): r.iterate(it)
ms = r.timer_stop()
print("%%.3f %%.4f" %% ((r.ray_total() - r0) / ms / 1e6, ms / n))
r.close(); s.close()
''' % (ROOT, ROOT)
settings = sys.argv[1].split(";"); cases = sys.argv[2].split(";"); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
res = {}
for rep in range(reps):
    for c in cases:
        for st in settings:
            env = dict(os.environ)
            if st != "-":
                for kv in st.split(","):
                    k, v = kv.split("="); env[k] = v
            o = subprocess.run([sys.executable, "-c", CHILD, c], env=env, capture_output=True, text=True)
            if o.returncode: print("FAILED", st, c, o.stderr[-400:], flush=True); continue
            g = float(o.stdout.split()[0]); res.setdefault((c, st), []).append(g)
            print(f"{c:46s} {st:28s} {g:8.3f} Grays/s", flush=True)
print("---- best / median")
for (c, st), v in res.items():
    v = sorted(v); print(f"{c:46s} {st:28s} best {v[-1]:8.3f}  median {v[len(v) // 2]:8.3f}")
