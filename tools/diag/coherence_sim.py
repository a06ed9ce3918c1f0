"""CPU model (no GPU): how much would ordering the BVH work list buy k_traverse_wide?

Rays of every bounce of one iteration of a mesh scene come from the oracle (the reference's order = the order k_list produces); the
per-ray visit counts of the wide walk come from tools/diag/wide_sim.cpp over the host copy of the 8-wide tree.  A wave of the walk
serves G = 64 / LPR rays and runs for its slowest one: cost(order) = sum over consecutive groups of G rays of max(visits), against
sum(visits) / G for perfectly uniform groups.  Prints that cost for several sort keys, globally sorted and sorted inside chunks of the
list (block-local sort), relative to the unsorted list.
usage: coherence_sim.py [scene] [res] [iteration]"""
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
W.wide_pred.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]

scn = sys.argv[1] if len(sys.argv) > 1 else "bunny_cornell.txt"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 400
it = int(sys.argv[3]) if len(sys.argv) > 3 else 1
CHUNK = int(os.environ.get("CHUNK", "1024"))
path = os.path.join(ROOT, "scenes", scn)
orc.set_trig_mode(orc.TRIG_PTK); orc.set_threads(8)
S = orc.Scene(path, res=(res, res))
ps = ptk.Scene(path, res=(res, res))
nodes, leaves, depth, stack = ps.wide_bvh()
leaves_f = np.ascontiguousarray(leaves).view(np.float32).reshape(-1, 16)
bv = S.bvh()
lo, hi = bv[0]["pMin"].astype(np.float64), bv[0]["pMax"].astype(np.float64)
level_of = np.zeros(len(nodes), np.int32)          # wide level of every node (breadth-first array: children come later)
for i in range(len(nodes)):
    for c in range(int(nodes[i][3] >> 24)):
        ref = int(nodes[i][4 + 4 * c + 3])
        if not ref & 0x80000000: level_of[ref] = level_of[i] + 1
print(f"{scn} {res}x{res}: {len(nodes)} wide nodes, {len(leaves)} leaves, root box {lo} .. {hi}")


def morton3(q, bits):
    out = np.zeros(len(q), np.int64)
    for b in range(bits):
        for a in range(3):
            out |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return out


def group_cost(v, G):
    n = len(v); pad = (-n) % G
    vv = np.concatenate([v, np.zeros(pad, v.dtype)]).reshape(-1, G)
    return vv.max(axis=1).sum()


def chunk_sorted(key, chunk):
    """order = stable sort by key inside consecutive chunks of the list"""
    n = len(key); idx = np.arange(n)
    return np.lexsort((idx, key, idx // chunk))


# occupancy grids over the root box: triangles (leaf boxes) overlapping every cell
def build_grid(R):
    g = np.zeros((R, R, R), np.int32)
    ext_ = hi - lo
    a = np.clip(np.floor((leaves["lo"].astype(np.float64) - lo) / ext_ * R).astype(int), 0, R - 1)
    b = np.clip(np.floor((leaves["hi"].astype(np.float64) - lo) / ext_ * R).astype(int), 0, R - 1)
    for k in range(len(leaves)):
        g[a[k, 0]:b[k, 0] + 1, a[k, 1]:b[k, 1] + 1, a[k, 2]:b[k, 2] + 1] += 1
    return g
GRIDS = {R: build_grid(R) for R in (8, 16, 32)}
for R, g in GRIDS.items(): print(f"grid {R}^3: {int((g > 0).sum())} of {R ** 3} cells occupied")


def grid_pred(R, ns, o_, d_, t0_, t1_, mode):
    g = GRIDS[R]; acc = np.zeros(len(o_), np.float64)
    for k in range(ns):
        t = t0_ + (t1_ - t0_) * ((k + 0.5) / ns)
        p = o_ + d_ * t[:, None]
        c = np.clip(np.floor((p - lo) / (hi - lo) * R).astype(int), 0, R - 1)
        cnt = g[c[:, 0], c[:, 1], c[:, 2]]
        acc += (cnt > 0) if mode == "occ" else np.log2(1.0 + cnt)
    return acc


paths = S.generate_rays(it)
img = np.zeros((S.P, 3), np.float32)
tot = {}
for d in range(S.trace_depth):
    isects, valid = S.intersect(paths)
    o = paths["o"].astype(np.float64); dd = paths["d"].astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / dd
        t0 = (lo - o) * inv; t1 = (hi - o) * inv
    tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
    t_an = np.where((isects["triangleID"] < 0) & (isects["t"] > 0), isects["t"], np.float32(3.0e38)).astype(np.float32)
    heavy = (tn <= tf) & (tf > 0) & ~(tn > t_an)
    hi_idx = np.nonzero(heavy)[0]
    n = len(hi_idx)
    ro = np.ascontiguousarray(paths["o"][hi_idx]); rd = np.ascontiguousarray(paths["d"][hi_idx]); ti = np.ascontiguousarray(t_an[hi_idx])
    nv = np.zeros(n, np.int32); nl = np.zeros(n, np.int32); tt = np.zeros(n, np.float32)
    W.wide_sim(nodes.ctypes.data, len(nodes), leaves_f.ctypes.data, n, ro.ctypes.data, rd.ctypes.data, ti.ctypes.data, nv.ctypes.data, nl.ctypes.data, tt.ctypes.data)
    v = (nv + nl).astype(np.int64)
    pred = np.zeros((n, 4), np.int32)
    W.wide_pred(nodes.ctypes.data, level_of.ctypes.data, leaves_f.ctypes.data, n, ro.ctypes.data, rd.ctypes.data, ti.ctypes.data, pred.ctypes.data)
    # candidate keys
    sgn = (rd > 0).astype(np.int64); octant = sgn[:, 0] | (sgn[:, 1] << 1) | (sgn[:, 2] << 2)
    ad = np.abs(rd); dom = np.where((ad[:, 0] > ad[:, 1]) & (ad[:, 0] > ad[:, 2]), 0, np.where(ad[:, 1] > ad[:, 2], 1, 2))
    tne = np.maximum(tn[hi_idx], 0.0); tfe = np.minimum(tf[hi_idx], ti.astype(np.float64))
    entry = ro + rd * tne[:, None]; ext = hi - lo
    mid = ro + rd * (0.5 * (tne + tfe))[:, None]
    chord = np.clip((tfe - tne) / np.linalg.norm(ext), 0, 0.999)

    def cell(p, bits): return np.clip(((p - lo) / ext * (1 << bits)).astype(np.int64), 0, (1 << bits) - 1)
    keys = {
        "unsorted": None,
        "octant": octant,
        "oct+entry3": (octant << 9) | morton3(cell(entry, 3), 3),
        "oct+entry2": (octant << 6) | morton3(cell(entry, 2), 2),
        "oct+mid3": (octant << 9) | morton3(cell(mid, 3), 3),
        "mid3+oct": (morton3(cell(mid, 3), 3) << 3) | octant,
        "mid4": morton3(cell(mid, 4), 4),
        "entry4": morton3(cell(entry, 4), 4),
        "chord16": (chord * 16).astype(np.int64),
        "chord8+oct+mid2": ((chord * 8).astype(np.int64) << 9) | (octant << 6) | morton3(cell(mid, 2), 2),
        "oct+mid2+chord4": (octant << 8) | (morton3(cell(mid, 2), 2) << 2) | (chord * 4).astype(np.int64),
        "oct+dom+mid3": (((octant * 3 + dom) << 9) | morton3(cell(mid, 3), 3)),
        "root_hits": pred[:, 0].astype(np.int64),
        "lvl1_hits": pred[:, 1].astype(np.int64),
        "lvl2_hits": np.minimum(pred[:, 3], 63).astype(np.int64),
        "lvl1+mid2": (np.minimum(pred[:, 1], 31).astype(np.int64) << 6) | morton3(cell(mid, 2), 2),
        "lvl2/4+mid2": ((np.minimum(pred[:, 3], 63).astype(np.int64) // 4) << 6) | morton3(cell(mid, 2), 2),
        "root_mask": (pred[:, 2] >> 8).astype(np.int64),
        "hits+mask": (pred[:, 0].astype(np.int64) << 8) | (pred[:, 2] >> 8).astype(np.int64),
        "hits+mid2": (pred[:, 0].astype(np.int64) << 6) | morton3(cell(mid, 2), 2),
        "hits+g16x8": (pred[:, 0].astype(np.int64) << 4) | grid_pred(16, 8, ro, rd, tne, tfe, "occ").astype(np.int64),
        "hits+chord4": (pred[:, 0].astype(np.int64) << 2) | (chord * 4).astype(np.int64),
        "hits+g16x16": (pred[:, 0].astype(np.int64) << 5) | grid_pred(16, 16, ro, rd, tne, tfe, "occ").astype(np.int64),
        "hits+g32x16": (pred[:, 0].astype(np.int64) << 5) | grid_pred(32, 16, ro, rd, tne, tfe, "occ").astype(np.int64),
        "g16x16+hits": (grid_pred(16, 16, ro, rd, tne, tfe, "occ").astype(np.int64) << 4) | pred[:, 0].astype(np.int64),
        "2hits+g16x16": (2 * pred[:, 0].astype(np.int64) + grid_pred(16, 16, ro, rd, tne, tfe, "occ").astype(np.int64)),
        "hits+g16x8log": (4 * pred[:, 0].astype(np.int64) + (grid_pred(16, 8, ro, rd, tne, tfe, "log")).astype(np.int64)),
        "3hits+g32x16log": (3 * pred[:, 0].astype(np.int64) + (grid_pred(32, 16, ro, rd, tne, tfe, "log")).astype(np.int64)),
        "g8x8occ": grid_pred(8, 8, ro, rd, tne, tfe, "occ").astype(np.int64),
        "g16x8occ": grid_pred(16, 8, ro, rd, tne, tfe, "occ").astype(np.int64),
        "g16x16occ": grid_pred(16, 16, ro, rd, tne, tfe, "occ").astype(np.int64),
        "g16x8log": (2 * grid_pred(16, 8, ro, rd, tne, tfe, "log")).astype(np.int64),
        "g16x16log": (2 * grid_pred(16, 16, ro, rd, tne, tfe, "log")).astype(np.int64),
        "g32x16log": (2 * grid_pred(32, 16, ro, rd, tne, tfe, "log")).astype(np.int64),
        "g32x32log": (2 * grid_pred(32, 32, ro, rd, tne, tfe, "log")).astype(np.int64),
        "oracle_log2": np.floor(np.log2(np.maximum(v, 1))).astype(np.int64),
        "oracle(visits)": v,
    }
    print(f"bounce {d}: {len(paths)} rays, {n} listed ({100.0 * n / max(len(paths), 1):.1f} %), visits/ray mean {v.mean():.2f} (nodes {nv.mean():.2f} leaves {nl.mean():.2f}) max {v.max()}")
    for G in (16, 32):
        base = group_cost(v, G)
        ideal = v.sum() / G
        row = []
        for name, key in keys.items():
            if key is None:
                c = base; cb = base
            else:
                order = np.argsort(key, kind="stable"); c = group_cost(v[order], G)
                cb = group_cost(v[chunk_sorted(key, CHUNK)], G)
            row.append((name, c / base, cb / base))
            t = tot.setdefault((G, name), [0, 0, 0]); t[0] += c; t[1] += cb; t[2] += base
        print(f"  G={G}: ideal {ideal / base:.3f} | " + " ".join(f"{nm} {a:.3f}/{b:.3f}" for nm, a, b in row))
    # next bounce
    valid_s = S.shade(orc.MODE_BSDF, it, d, isects, paths, img)
    paths, _ = orc.compact(valid_s, paths)
    if len(paths) == 0:
        break
print("---- whole iteration: cost relative to the unsorted list (global sort / sort inside 1024-ray chunks)")
for (G, name), (c, cb, base) in tot.items():
    print(f"  G={G:2d} {name:18s} {c / base:.3f} / {cb / base:.3f}")
