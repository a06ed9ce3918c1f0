"""Quick A/B timing helper (GPU box): ms/iteration and Mrays/s of a scene for given lanes / pipeline."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")


def run(scn, mode=0, lanes=1, pipeline=0, steps=32, res=None, depth=-1, graph=True):
    scene = ptk.Scene(os.path.join(ROOT, "scenes", scn), res=res, depth=depth)
    r = ptk.Renderer(scene, mode, pipeline=pipeline, in_flight=lanes, graph=graph)
    for it in range(1, 9): r.iterate(it)
    r.sync(); r0 = r.ray_total(); r.timer_start()
    for it in range(9, 9 + steps): r.iterate(it)
    ms = r.timer_stop(); rays = r.ray_total() - r0
    r.close()
    return round(ms / steps, 4), round(rays / ms / 1e3, 1)


if __name__ == "__main__":
    pipes = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0]
    for scn, mode in (("cornell.txt", 0), ("cornell.txt", 2), ("mis_test.txt", 2), ("bunny_cornell.txt", 0), ("metallic_analytic.txt", 0)):
        for pipe in pipes:
            for lanes in (1, 2, 3, 4):
                print(scn, "mode", mode, "pipe", pipe, "L", lanes, run(scn, mode, lanes, pipeline=pipe), flush=True)
