import importlib, os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
for scn, mode in (("bunny_cornell.txt", 0), ("bunny_cornell.txt", 2), ("cornell.txt", 0)):
    for lanes, graph in ((3, True), (3, False), (1, False), (1, True), (2, True), (4, True)):
        s = ptk.Scene(os.path.join(ROOT, "scenes", scn))
        r = ptk.Renderer(s, mode, in_flight=lanes, graph=graph)
        for it in range(1, 33): r.iterate(it)
        r.sync(); r0 = r.ray_total(); r.timer_start()
        for it in range(33, 33 + 96): r.iterate(it)
        ms = r.timer_stop()
        print(f"{scn} mode {mode} lanes {lanes} graph {graph}: {(r.ray_total() - r0) / ms / 1e6:.3f} Grays/s {ms / 96:.4f} ms/iteration", flush=True)
        r.close(); s.close()
