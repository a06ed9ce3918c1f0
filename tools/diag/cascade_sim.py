"""CPU model (no GPU): a CAPPED, multi-stage wide BVH walk.  Stage s walks every ray of its list for at most K_s loop iterations; the rays that are not
done are compacted (in list order) into the next stage's list with their stack and hit state.  A wave serves G = 64 / LPR rays and runs for its slowest
one, so cost = sum over stages of sum over groups of min(max(remaining), K_s), against sum over groups of max(visits) for the single launch.
Per-ray loop iterations come from tools/diag/wide_sim.cpp, rays from the oracle (the reference's order).   usage: cascade_sim.py [scene] [res] [iteration]"""
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc                                    # noqa: E402
ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
so = "/tmp/libwidesim.so"
subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "diag", "wide_sim.cpp")])
W = C.CDLL(so)
W.wide_sim.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
scn = sys.argv[1] if len(sys.argv) > 1 else "bunny_cornell.txt"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 400
it = int(sys.argv[3]) if len(sys.argv) > 3 else 1
path = os.path.join(ROOT, "scenes", scn)
orc.set_trig_mode(orc.TRIG_PTK); orc.set_threads(8)
S = orc.Scene(path, res=(res, res)); ps = ptk.Scene(path, res=(res, res))
nodes, leaves, depth, stack = ps.wide_bvh()
leaves_f = np.ascontiguousarray(leaves).view(np.float32).reshape(-1, 16)
bv = S.bvh(); lo, hi = bv[0]["pMin"].astype(np.float64), bv[0]["pMax"].astype(np.float64)


def group_cost(v, G, cap=None):
    n = len(v); pad = (-n) % G
    vv = np.concatenate([v, np.zeros(pad, v.dtype)]).reshape(-1, G).max(axis=1)
    return int((np.minimum(vv, cap) if cap else vv).sum())


def cascade(v, G, caps):
    total = 0; rem = v.copy(); sizes = []
    for k in list(caps) + [None]:
        if len(rem) == 0: break
        sizes.append(len(rem))
        total += group_cost(rem, G, k)
        if k is None: break
        rem = rem[rem > k] - k
    return total, sizes


paths = S.generate_rays(it)
acc = {}
for d in range(S.trace_depth):
    isects, valid = S.intersect(paths)
    o = paths["o"].astype(np.float64); dd = paths["d"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dd; t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
    t_an = np.where((isects["triangleID"] < 0) & (isects["t"] > 0), isects["t"], np.float32(3.0e38)).astype(np.float32)
    idx = np.nonzero((tn <= tf) & (tf > 0) & ~(tn > t_an))[0]
    n = len(idx)
    if n:
        ro = np.ascontiguousarray(paths["o"][idx]); rd = np.ascontiguousarray(paths["d"][idx]); ti = np.ascontiguousarray(t_an[idx])
        nv = np.zeros(n, np.int32); nl = np.zeros(n, np.int32); tt = np.zeros(n, np.float32)
        W.wide_sim(nodes.ctypes.data, len(nodes), leaves_f.ctypes.data, n, ro.ctypes.data, rd.ctypes.data, ti.ctypes.data, nv.ctypes.data, nl.ctypes.data, tt.ctypes.data)
        v = (nv + nl).astype(np.int64)
        row = f"bounce {d}: {n:7d} listed rays, iterations mean {v.mean():5.1f} p50 {np.percentile(v, 50):3.0f} p90 {np.percentile(v, 90):3.0f} p99 {np.percentile(v, 99):3.0f} max {v.max():3d}"
        for G in (32, 16):
            base = group_cost(v, G)
            acc.setdefault((G, "single"), 0); acc[(G, "single")] += base
            acc.setdefault((G, "ideal"), 0); acc[(G, "ideal")] += int(np.ceil(v.sum() / G))
            for caps in ((8,), (12,), (16,), (24,), (8, 16), (12, 24), (16, 32), (8, 16, 32)):
                c, sizes = cascade(v, G, caps)
                acc.setdefault((G, caps), 0); acc[(G, caps)] += c
        print(row, flush=True)
    valid2 = S.shade(0, it, d, isects, paths, np.zeros((S.P, 3), np.float32)) if False else None
    # advance the paths one bounce with the oracle (BSDF)
    img = np.zeros((S.P, 3), np.float32)
    keep = valid.astype(bool); paths = paths[keep].copy(); isects = isects[keep].copy()
    if len(paths) == 0: break
    v2 = S.shade(0, it, d, isects, paths, img)
    paths = paths[v2.astype(bool)].copy()
    if len(paths) == 0: break
print("wave-iterations of one iteration's walks (all bounces), relative to the single launch:")
for G in (32, 16):
    b = acc[(G, "single")]
    print(f"  G = {G} rays per wave: single {b}, ideal (uniform waves) {acc[(G, 'ideal')] / b:.3f}")
    for k, c in acc.items():
        if k[0] == G and k[1] not in ("single", "ideal"): print(f"     caps {str(k[1]):14s} {c / b:.3f}")
