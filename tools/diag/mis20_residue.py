"""Where are the lit plate pixels of gallery/MIS20.png that the oracle (slot table, libm trig, no FMA) does NOT reproduce bit for bit?
(VERDICT r3 #4.)  CPU only; needs tests/golden/gallery_mis_test2.npz.  Prints, for MIS20 (and Direct20 / BSDF20 for comparison):
  * the histogram of the level deltas of the mismatching pixels;
  * the mismatch rate per plate (roughness 0.25 / 0.1 / 0.05 / 0.01; the plate of a pixel = the primitive its camera ray hits);
  * MIS only: the mismatch rate against which of MisPTkernel's two terms carries the pixel — the BSDF-sampled light hits weighted by
    powerHeuristic(prevPdf, lightPDF) (pathtrace.cu:545-556) or the next-event samples weighted by powerHeuristic(lightPdf, bsdfPdf)
    (:558-571) — rendered separately with the oracle (orc_set_debug_mis_terms);
  * the mismatch rate against the distance of the pixel's (pre-quantisation) value from the nearest 8-bit level boundary: rounding noise
    flips only values that sit next to a boundary, a formula slip would not care."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
import test_gallery as tg

W, R0, R1 = tg.W, tg.R0, tg.R1
orc.L.orc_set_debug_slot_shift.argtypes = [orc.ci]; orc.L.orc_set_debug_mis_terms.argtypes = [orc.ci]
orc.set_trig_mode(orc.TRIG_LIBM); orc.set_threads(os.cpu_count() or 1)
g = tg._fixture()
table = g["slot_shift"].astype(np.int32)

# plate of every pixel (image coordinates after saveImage's x-flip): geomID of the camera ray's hit
s = orc.Scene(tg.SCENE)
isects, valid = s.intersect(s.generate_rays(1))
geom = np.where(valid > 0, isects["geomID"], -1).reshape(W, W)[:, ::-1]
mats = s.materials(); geoms = s.geoms()
s.close()
plates = {int(i): float(mats[int(geoms[i]["materialid"])]["rough_s"]) for i in range(len(geoms)) if geoms[i]["type"] == 1}
print("plates (geom -> roughness):", plates)


def pre_quant(acc):
    """saveImage's transform up to the float that is cast to uchar"""
    x = acc.reshape(W, W, 3) / np.float32(20)
    a, b, c, d, e = (np.float32(v) for v in (2.51, 0.03, 2.43, 0.59, 0.14))
    y = np.clip((x * (a * x + b)) / (x * (c * x + d) + e), 0, 1).astype(np.float32)
    y = np.power(y, np.float32(1 / 2.2)).astype(np.float32)[:, ::-1, :]
    return np.clip(y, 0, 1) * np.float32(255)


for name in ("Direct", "BSDF", "MIS"):
    acc = tg._render_with_table(orc, tg.MODES[name], table)
    out = tg.save_transform(acc, 20)
    ref = g["rows_" + name + "20"].astype(np.int32)
    d = out[R0:R1].astype(np.int32) - ref
    lit = g["lit"] & (ref.sum(axis=2) > 0)
    bad = (np.abs(d).max(axis=2) > 0) & lit
    print(f"\n== {name}20: {int(lit.sum())} lit plate pixels, {int(bad.sum())} differ ({100.0 * bad.sum() / lit.sum():.2f} %)")
    dd = d[bad]; worst = dd[np.arange(len(dd)), np.abs(dd).argmax(axis=1)]
    print("   level delta (ours - gallery) of the differing pixels:", {int(k): int(v) for k, v in zip(*np.unique(np.clip(worst, -4, 4), return_counts=True))})
    gp = geom[R0:R1]
    for gi, rough in sorted(plates.items(), key=lambda kv: -kv[1]):
        m = lit & (gp == gi)
        if m.sum(): print(f"   plate roughness {rough:5.3f}: {int(m.sum()):7d} lit pixels, {100.0 * (bad & m).sum() / m.sum():6.2f} % differ")
    # distance of the pre-quantisation value from the nearest level boundary (any channel that differs / the closest channel otherwise)
    pq = pre_quant(acc)[R0:R1]
    frac = pq - np.floor(pq); dist = np.minimum(frac, 1 - frac).min(axis=2)
    for lo_, hi_ in ((0, 0.001), (0.001, 0.01), (0.01, 0.05), (0.05, 0.2), (0.2, 0.51)):
        m = lit & (dist >= lo_) & (dist < hi_)
        if m.sum(): print(f"   distance to a level boundary in [{lo_}, {hi_}): {int(m.sum()):7d} pixels, {100.0 * (bad & m).sum() / m.sum():6.2f} % differ")
    if name == "MIS":
        terms = {}
        for mask, tn in ((1, "bsdf_hit"), (2, "nee")):
            orc.L.orc_set_debug_mis_terms(mask)
            terms[tn] = tg._render_with_table(orc, tg.MODES["MIS"], table).reshape(W, W, 3)[:, ::-1][R0:R1].sum(axis=2)
        orc.L.orc_set_debug_mis_terms(3)
        share = terms["bsdf_hit"] / np.maximum(terms["bsdf_hit"] + terms["nee"], 1e-20)
        for lo_, hi_ in ((0, 0.01), (0.01, 0.25), (0.25, 0.75), (0.75, 0.99), (0.99, 1.01)):
            m = lit & (share >= lo_) & (share < hi_)
            if m.sum(): print(f"   share of the BSDF-sampled light hits in the pixel's radiance in [{lo_}, {hi_}): {int(m.sum()):7d} pixels, {100.0 * (bad & m).sum() / m.sum():6.2f} % differ")
        # number of additions the pixel's sum went through (iterations that contributed either term are not separable here; use the count of non-zero terms)
