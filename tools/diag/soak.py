"""GPU soak: long renders with iterations in flight + graph replay (and the sparse tail) twice; the two images must be
bit-identical (determinism), finite and non-negative."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
for scn, mode, iters, kw in (("cornell.txt", 0, 3000, dict(in_flight=4, graph=True)), ("mis_test.txt", 2, 3000, dict(in_flight=4, graph=True)),
                             ("bunny_cornell.txt", 2, 600, dict(in_flight=4, graph=True)), ("bunny_cornell.txt", 0, 300, dict(in_flight=1)),
                             ("bunny_cornell.txt", 2, 600, dict(in_flight=4, graph=True, pipeline=10)), ("textureEnv_test.txt", 2, 400, dict(in_flight=4, graph=True, pipeline=10)),
                             ("textureEnv_test.txt", 2, 400, dict(in_flight=4, graph=True)), ("obj_test.txt", 2, 600, dict(in_flight=4, graph=True)),
                             ("mis_test2.txt", 1, 1000, dict(in_flight=4, graph=True))):
    imgs = []
    for rep in range(2):
        s = ptk.Scene(os.path.join(ROOT, "scenes", scn))
        r = ptk.Renderer(s, mode, **kw)
        t0 = time.time()
        for it in range(1, iters + 1): r.iterate(it)
        img = r.image(); dt = time.time() - t0
        rays = r.ray_total()
        r.close(); s.close()
        imgs.append(img)
    ok = np.array_equal(imgs[0], imgs[1]) and np.isfinite(imgs[0]).all() and (imgs[0] >= 0).all()
    same = ""
    if kw.get("in_flight", 1) > 1:
        # round 4, ordered accumulation: the image with iterations in flight IS the image of one iteration at a time, however long the render
        s = ptk.Scene(os.path.join(ROOT, "scenes", scn))
        r = ptk.Renderer(s, mode, pipeline=kw.get("pipeline", 0))
        for it in range(1, iters + 1): r.iterate(it)
        one = r.image(); r.close(); s.close()
        eq = np.array_equal(one, imgs[0]); ok = ok and eq
        same = ", == one iteration at a time" if eq else f", DIFFERS from one iteration at a time in {int((one != imgs[0]).any(axis=1).sum())} pixels"
    print(f"{scn} mode {mode} {iters} iterations {kw}: {'deterministic' if ok else 'MISMATCH'}{same}, {rays / dt / 1e9:.2f} Grays/s incl. readback, mean radiance {imgs[0].mean() / iters:.5f}")
    assert ok

# graph re-capture with replays in flight: the sample mode flips every 40 iterations with four lanes replaying hipGraphs and no
# host synchronisation in between; every flip re-captures each lane's graph (after a stream synchronise, pathtrace.hip build_graph) and resets
# the sparse-tail hint.  Two runs must agree bit for bit.
imgs = []
for rep in range(2):
    s = ptk.Scene(os.path.join(ROOT, "scenes", "mis_test.txt"))
    r = ptk.Renderer(s, 2, in_flight=4, graph=True)
    for it in range(1, 1201):
        if it % 40 == 0:
            ptk.lib.ptk_set_sample_mode((it // 40) % 3)
        r.iterate(it)
    imgs.append(r.image()); r.close(); s.close()
ok = np.array_equal(imgs[0], imgs[1]) and np.isfinite(imgs[0]).all()
print(f"mis_test.txt, sample mode switched every 40 of 1200 iterations, 4 lanes + graph replay: {'deterministic' if ok else 'MISMATCH'}, mean radiance {imgs[0].mean() / 1200:.5f}")
assert ok
