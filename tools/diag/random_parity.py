"""GPU box: randomised parity sweep (diagnostic, beyond the fixed cases of tests/).  Random analytic scenes — a room with a random subset
of its walls (open and closed scenes), 3-60 random cubes / spheres (rotated, overlapping), materials of all five types, sphere and cube
lights — rendered in a random mode with random resolution, depth, iterations in flight, graph replay, tile partition and samples per launch;
image and live counts of every iteration must equal the CPU oracle's bit for bit.   usage: [MESHES=1] random_parity.py [n_cases] [first_seed]   (MESHES=1 adds 1-3 OBJ meshes — cube / sphere / bunny — to most scenes)"""
import importlib, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
import oracle_lib as orc
orc.set_trig_mode(orc.TRIG_PTK)
MESHES = int(os.environ.get("MESHES", "0"))
try: orc.set_threads(min(16, os.cpu_count() or 1))
except Exception: pass


def scene_text(rng):
    L = []
    types = ["Lambertian", "MetallicWorkflow", "Dielectric", "Microfacet"]
    n_mats = int(rng.integers(2, 24))
    for k in range(n_mats):
        t = types[int(rng.integers(0, 4))]
        L += [f"MATERIAL m{k}", f"TYPE {t}", "ALBEDO %.4f %.4f %.4f" % tuple(rng.uniform(0.15, 0.98, 3)), "METALLIC %.3f" % rng.uniform(0, 1),
              "ROUGHNESS %.3f" % rng.uniform(0.0, 0.9), "IOR %.3f" % rng.uniform(1.1, 1.9), ""]
    n_lights = int(rng.integers(0, 4))
    for k in range(max(n_lights, 1)):
        L += [f"MATERIAL l{k}", "TYPE Light", "ALBEDO %.3f %.3f %.3f" % tuple(rng.uniform(1, 9, 3)), ""]
    W, H = int(rng.integers(5, 21)) * 8, int(rng.integers(5, 21)) * 8
    depth = int(rng.integers(1, 11))
    L += ["CAMERA", f"RES {W} {H}", "FOVY %.1f" % rng.uniform(20, 60), "ITERATIONS 4", f"DEPTH {depth}", "FILE rnd",
          "EYE %.3f %.3f %.3f" % (rng.uniform(-2, 2), rng.uniform(3, 7), rng.uniform(8, 15)), "LOOKAT %.3f %.3f 0" % (rng.uniform(-1, 1), rng.uniform(4, 6)), "UP 0 1 0", ""]
    objs = []
    # the room spans z = -6 .. 15.6 and holds the camera (round 5: with the front wall at z = 6.5 in front of the camera every closed room rendered black —
    # half of the sweep's images; the live counts were compared all the same)
    walls = [((0, 0, 4.8), (12, 0.02, 21.6)), ((0, 10, 4.8), (12, 0.02, 21.6)), ((-6, 5, 4.8), (0.02, 10, 21.6)), ((6, 5, 4.8), (0.02, 10, 21.6)), ((0, 5, -6), (12, 10, 0.02)),
             ((0, 5, 15.6), (12, 10, 0.02))]
    p_wall = rng.choice([0.0, 0.5, 1.0])
    for pos, sc in walls:
        if rng.uniform() < p_wall: objs.append(("cube", f"m{int(rng.integers(0, n_mats))}", pos, (0, 0, 0), sc))
    for k in range(n_lights):
        kind = "sphere" if rng.uniform() < 0.7 else "cube"
        objs.append((kind, f"l{k}", tuple(rng.uniform([-4, 6, -4], [4, 9.5, 3])), (0, 0, 0), tuple(np.repeat(rng.uniform(0.5, 2.5), 3))))
    for k in range(int(rng.choice([3, 8, 20, 60]))):
        kind = "cube" if rng.uniform() < 0.5 else "sphere"
        mat = f"m{int(rng.integers(0, n_mats))}" if rng.uniform() < 0.93 else f"l{int(rng.integers(0, max(n_lights, 1)))}"
        objs.append((kind, mat, tuple(rng.uniform([-5, 0.5, -5], [5, 9, 4])), tuple(rng.uniform(0, 90, 3) * (rng.uniform() < 0.5)), tuple(rng.uniform(0.3, 2.2, 3))))
    # meshes (MESHES=1): cube.obj (12 triangles: up to two of them stay inside the fused kernel's inline walk), sphere.obj (320), bunny10k.obj —
    # sometimes with a light material (triangle lights: light_sample / light_pdf on triangles)
    if MESHES and rng.uniform() < 0.8:
        for k in range(int(rng.integers(1, 4))):
            f, scale = [("cube.obj", rng.uniform(0.8, 2.5)), ("sphere.obj", rng.uniform(0.5, 1.5)), ("bunny10k.obj", rng.uniform(2, 5))][int(rng.choice([0, 0, 1, 2]))]
            mat = f"m{int(rng.integers(0, n_mats))}" if rng.uniform() < 0.8 else f"l{int(rng.integers(0, max(n_lights, 1)))}"
            objs.append((os.path.join(ROOT, "scenes", "model", f), mat, tuple(rng.uniform([-3.5, 0.5, -3.5], [3.5, 6, 2.5])), tuple(rng.uniform(0, 360, 3) * (rng.uniform() < 0.6)),
                         tuple(np.repeat(scale, 3))))
    for i, (kind, mat, pos, rot, sc) in enumerate(objs):
        L += [f"OBJECT o{i}", kind, f"material {mat}", "TRANS %.6g %.6g %.6g" % tuple(pos), "ROTAT %.6g %.6g %.6g" % tuple(rot), "SCALE %.6g %.6g %.6g" % tuple(sc), ""]
    return "\n".join(L) + "\n", (W, H, depth, len(objs), n_lights)


def one_case(seed, tmp):
    rng = np.random.default_rng(seed)
    txt, info = scene_text(rng)
    path = os.path.join(tmp, f"rnd{seed}.txt"); open(path, "w").write(txt)
    mode = int(rng.integers(0, 3)); lanes = int(rng.integers(1, 5)); graph = bool(rng.integers(0, 2)) or lanes > 1
    part = (1, 0, 1); spp = 1
    if rng.uniform() < 0.3:
        n = int(rng.integers(2, 5)); part = (n, int(rng.integers(0, n)), int(rng.choice([1, 4, 8]))); spp = int(rng.integers(1, 4))
    pipeline = int(rng.choice([0, 0, 0, 10, 8, 1, 11] if MESHES else [0, 0, 0, 10, 8, 11, 11]))      # (11, round 5: k_trace_shade where the scene has no triangles)
    iters = int(rng.integers(3, 13))
    # round-5 variants that are read when the context is initialised: wave windows (k_shade_trace_w), batched ordered merges, the pooled wide walk
    knobs = {"PTK_WAVE_WINDOWS": str(int(rng.integers(0, 2))), "PTK_MERGE_BATCH": str(int(rng.choice([1, 1, 2, 3]))), "PTK_WIDE_POOL": str(int(rng.choice([0, 0, 64, 512]))),
             "PTK_WIDE_WIDTH": str(int(rng.choice([8, 8, 16])))}                 # (round 5: the 16-wide variant of the collapsed tree; read at upload)
    os.environ.update(knobs)
    s = ptk.Scene(path); o = orc.Scene(path)
    tiles = int(rng.choice([1, 1, 2, 3])) if (lanes == 1 and pipeline in (0, 10) and part[0] == 1) else 1      # tiles in flight: only alone, on the fused launches
    kw = dict(in_flight=lanes, graph=graph or tiles > 1, pipeline=pipeline, tiles=tiles)
    if part[0] > 1: kw.update(partition=part, spp=spp)
    r = ptk.Renderer(s, mode, **kw)
    acc = orc.Accumulator(o.P, lanes, spp, mode)          # the product's accumulation (ordered; planes per bounce in MIS)
    for it in range(1, iters + 1):
        r.iterate(it)
        if part[0] > 1: _, live = acc.batch(o, mode, it, partition=part)
        else: _, live = acc.iteration(o, mode, it)
        if it % 3 == 0 or it == iters:
            counts, _ = r.live_counts()
            assert np.array_equal(counts, live), (seed, it, counts, live)
    ref = acc.img
    img = r.image(); r.close(); s.close(); o.close()
    bad = int((img != ref).any(axis=1).sum())
    for k in knobs: os.environ.pop(k, None)
    return bad, info, dict(mode=mode, lanes=lanes, tiles=tiles, graph=graph, part=part, spp=spp, pipeline=pipeline, iters=iters, ww=knobs["PTK_WAVE_WINDOWS"], mb=knobs["PTK_MERGE_BATCH"],
                           pool=knobs["PTK_WIDE_POOL"], width=knobs["PTK_WIDE_WIDTH"], lit=float(ref.sum()))


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    fails = 0; lit = 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(first, first + n):
            bad, info, cfg = one_case(seed, tmp)
            fails += bad > 0; lit += cfg["lit"] > 0
            print(f"seed {seed} {'OK ' if bad == 0 else 'FAIL'} res {info[0]}x{info[1]} depth {info[2]} prims {info[3]} lights {info[4]} {cfg}" + (f" {bad} pixels differ" if bad else ""), flush=True)
    print(f"{n} random cases, {fails} failures, {lit} with non-black images")
    sys.exit(1 if fails else 0)
