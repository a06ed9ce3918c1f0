"""GPU box: does the contracted flavour (libptk_fma.so) reproduce the reference's 20-spp gallery images pixel for pixel WITHOUT the
fitted slot table?  (It would if clang's contraction made the same hit/miss decisions on the sphere silhouettes as nvcc's.)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import test_gallery as tg
g = tg._fixture()
for flavour in ("fma", "exact"):
    for name in ("Direct", "MIS", "BSDF"):
        out = f"/tmp/gal_{flavour}_{name}.npz"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "flavour_child.py"), out, "mis_test2.txt", "-1", "-1", "-1", str(tg.MODES[name]), "20"],
                           env=dict(os.environ, PTK_FLAVOUR=flavour), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-1000:]
        img = np.load(out)["img20"]
        u8 = tg.save_transform(img, 20)
        exact, within1, n = tg._plate_match(u8, g, name + "20")
        print(f"{flavour:5s} {name}20 (no table): exact {exact:.4f}, within 1 level {within1:.4f} of {n} lit plate pixels")
