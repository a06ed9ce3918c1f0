"""GPU box: where does the in-kernel BVH walk (meshes of <= PTK_INLINE_BVH_MAX_TRIS triangles) stop paying?  obj_test.txt with K copies of
cube.obj (12 K triangles), default pipeline, threshold 32 (work list from K = 3) against threshold 100000 (always inline)."""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import importlib, os, sys
sys.path.insert(0, %r)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
s = ptk.Scene(sys.argv[1]); r = ptk.Renderer(s, int(sys.argv[2]), in_flight=3, graph=True)
for it in range(1, 33): r.iterate(it)
r.sync(); r0 = r.ray_total(); r.timer_start()
for it in range(33, 129): r.iterate(it)
ms = r.timer_stop(); print("%%.3f" %% ((r.ray_total() - r0) / ms / 1e6)); r.close()
''' % ROOT
base = open(os.path.join(ROOT, "scenes", "obj_test.txt")).read().replace("../scenes/model/cube.obj", os.path.join(ROOT, "scenes", "model", "cube.obj"))
import numpy as np
rng = np.random.default_rng(5)
with tempfile.TemporaryDirectory() as tmp:
    MESH = sys.argv[1] if len(sys.argv) > 1 else "cube.obj"; TRI = {"cube.obj": 12, "sphere.obj": 320, "bunny10k.obj": 10000}[MESH]
    for K in ((1, 2, 3, 4, 6, 8, 12, 16, 24) if MESH == "cube.obj" else (1, 2, 4, 8, 16)):
        txt = base.replace("cube.obj", MESH)
        for k in range(1, K):
            p = rng.uniform([-3.5, 1, -3.5], [3.5, 8, 2.5])
            txt += "\nOBJECT extra%d\n%s\nmaterial 4\nTRANS %.3f %.3f %.3f\nROTAT %.1f %.1f 0\nSCALE 1 1 1\n" % (k, os.path.join(ROOT, "scenes", "model", MESH), p[0], p[1], p[2], rng.uniform(0, 90), rng.uniform(0, 90))
        path = os.path.join(tmp, "k%d.txt" % K); open(path, "w").write(txt)
        out = []
        for thr in ("32", "100000"):
            for mode in ("0", "2"):
                o = subprocess.run([sys.executable, "-c", CHILD, path, mode], env=dict(os.environ, PTK_INLINE_BVH_MAX_TRIS=thr), capture_output=True, text=True)
                out.append(o.stdout.strip() or o.stderr[-200:])
        print("K %2d (%5d triangles): threshold 32 BSDF %s MIS %s | always inline BSDF %s MIS %s" % (K, TRI * K, out[0], out[1], out[2], out[3]), flush=True)
