"""Fits, on gallery/MIS20.png, the compacted-slot offset the reference's GPU build ran with at depth >= 1 (the fixture's `slot_shift` is the
depth-0 offset, fitted on Direct20.png, where every path dies after one bounce).  Coordinate descent over the 20 iterations: iteration `it`
is re-rendered with every candidate k1 for depth >= 1 (depth 0 keeps the fixture's table), the 20-iteration sum is re-formed with that
iteration replaced, and per 4-row band the k1 with the most bit-identical lit plate pixels wins.  Writes /tmp/deep_shift.npy [21, 1080].
CPU only (oracle).  usage: fit_deep_shift.py [kmin kmax] [passes]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
import test_gallery as tg
W, R0, R1 = tg.W, tg.R0, tg.R1
BAND = 4
kmin, kmax = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (-12, 12)
passes = int(sys.argv[3]) if len(sys.argv) > 3 else 1
orc.L.orc_set_debug_slot_shift.argtypes = [orc.ci]; orc.L.orc_set_debug_slot_shift_deep.argtypes = [orc.ci, orc.ci]
orc.set_trig_mode(orc.TRIG_LIBM); orc.set_threads(os.cpu_count() or 1)
g = tg._fixture(); table = g["slot_shift"].astype(np.int32)
ref = g["rows_MIS20"].astype(np.int32)
lit = g["lit"] & (ref.sum(axis=2) > 0)
s = orc.Scene(tg.SCENE)


def render_it(it, k1_rows):
    """iteration `it` alone: depth-0 offset = table[it] per row, depth >= 1 offset = k1_rows per row"""
    one = np.zeros((W, W, 3), np.float32)
    pairs = np.unique(np.stack([table[it], k1_rows], axis=1), axis=0)
    for k0, k1 in pairs:
        orc.L.orc_set_debug_slot_shift(int(k0)); orc.L.orc_set_debug_slot_shift_deep(1, int(k1))
        img = np.zeros((W * W, 3), np.float32)
        s.pathtrace(2, it, img)
        rows = (table[it] == k0) & (k1_rows == k1)
        one[rows] = img.reshape(W, W, 3)[rows]
    return one


def matches(acc):
    out = tg.save_transform(acc, 20)
    return (np.abs(out[R0:R1].astype(np.int32) - ref).max(axis=2) == 0) & lit


deep = table.copy()                                   # start: the same offset at every depth (what tests/test_gallery.py does today)
per_it = [None] + [render_it(it, deep[it]) for it in range(1, 21)]
acc = np.zeros((W, W, 3), np.float32)
for it in range(1, 21): acc += per_it[it]
print("start: exact %.4f" % (matches(acc).sum() / lit.sum()), flush=True)
for p in range(passes):
    for it in range(1, 21):
        rest = np.zeros((W, W, 3), np.float32)
        for j in range(1, 21):
            if j != it: rest += per_it[j]
        best = np.full((R1 - R0) // BAND, -1, np.int64); best_k = np.zeros((R1 - R0) // BAND, np.int64); cand_img = {}
        for k1 in range(kmin, kmax + 1):
            one = render_it(it, np.full(W, k1, np.int32))
            m = matches(rest + one).reshape(-1, BAND, W).sum(axis=(1, 2))
            better = m > best
            best[better] = m[better]; best_k[better] = k1
            cand_img[k1] = one
        # mode filter over 21 rows like the depth-0 fit, then adopt
        rows_k = np.repeat(best_k, BAND)
        sm = rows_k.copy()
        for r in range(len(rows_k)):
            w = rows_k[max(0, r - 10):r + 11]; vals, cnt = np.unique(w, return_counts=True); sm[r] = vals[cnt.argmax()]
        deep[it, R0:R1] = sm; deep[it, R1:] = sm[-1]
        new = per_it[it].copy()
        for k1 in np.unique(sm):
            rr = np.zeros(W, bool); rr[R0:R1] = sm == k1
            new[rr] = cand_img[int(k1)][rr]
        per_it[it] = new
        acc = rest + new
        print(f"pass {p} iteration {it}: deep offsets {dict(zip(*[x.tolist() for x in np.unique(sm, return_counts=True)]))} (depth 0: {dict(zip(*[x.tolist() for x in np.unique(table[it, R0:R1], return_counts=True)]))}) -> exact {matches(acc).sum() / lit.sum():.4f}", flush=True)
np.save("/tmp/deep_shift.npy", deep)
# final, in iteration order
acc = np.zeros((W, W, 3), np.float32)
for it in range(1, 21): acc += render_it(it, deep[it])
m = matches(acc)
print("final: exact %.4f of %d lit plate pixels" % (m.sum() / lit.sum(), lit.sum()))
orc.L.orc_set_debug_slot_shift(0); orc.L.orc_set_debug_slot_shift_deep(0, 0)
