"""Builds tests/golden/gallery_mis_test2.npz from the ONLY outputs the reference itself holds for this path: the six
1080x1080 renders gallery/{Direct,BSDF,MIS}{20,2000}.png of scenes/mis_test2.txt (three sphere lights over four glossy
plates; camera mis_test2.txt:69-74), one per SampleMode (preview.h:13), written by saveImage (main.cpp:146-170: sum / spp,
ACESFilm, gamma 1/2.2, x-flip, (unsigned char)(clamp * 255)).  Runs only where /root/reference exists (the build
container); the .npz is DATA (pixel bytes, block means and one fitted integer table), no reference source.

    python tests/golden/make_gallery_fixture.py

What goes in, per image:
  blocks8_<name>   float32 [135,135,3]  means of 8x8 pixel blocks (8-bit levels) of the whole image
  rows_<name>      uint8   [304,1080,3] raw pixel rows 636..939 (all four plates + their reflections); 20-spp images only
  plate4_<name>    float32 [76,270,3]   4x4 block means of the same rows; 2000-spp images only
  crop_<name>      uint8   [64,64,3]    raw crop rows 650..713 x cols 500..563 (medium light's reflection, top plate)
and once:
  slot_shift       int8    [21,1080]    see below
  slot_shift_deep  int8    [21,1080]    the same offset for depth >= 1 (fit_deep_shifts; fitted on MIS20.png)
  slot_shift_deep_left int8 [21,1080]   the same fit with only columns 0..539 of MIS20.png voting (hold-out: tests score columns 540..1079)
  lit              bool    [304,1080]   plate pixels (Direct2000 sum of channels > 30) in rows 636..939

**slot_shift: what it is and why it exists.**  The reference keys its shading RNG by the COMPACTED slot of a path
(pathtrace.cu:363,458,532), i.e. by the number of hits among all earlier pixels.  Its GPU build (nvcc: fused multiply-adds,
CUDA sinf/cosf/sqrtf) decides a handful of the ~430 sphere-silhouette pixels per iteration differently from any non-FMA
evaluation of the same formulas (the sphere radicand cancels catastrophically at 177 units distance), and every such flip
moves all later slots by one.  So iteration i of the gallery image was shaded with slot + k_i for a small integer k_i
(constant over a plate, sometimes stepping by one at a plate edge).  k is fitted here, per iteration and per 4-row band, from
Direct20.png alone: the oracle renders iteration i with every k in [-12,12] and the k whose noise residual (against
Direct2000.png as the noise-free image) correlates best with the gallery's residual wins (the peak is 1/sqrt(20) = 0.22
against a 0.05-0.12 floor, >10 sigma per band); a 21-row mode filter removes single-band outliers.  MIS20.png and
BSDF20.png are NOT used in the fit — tests/test_gallery.py checks them out of sample with the same table.
Result (recorded in DESIGN.md §3): with the table the oracle reproduces the gallery pixel for pixel — Direct 99.4 %,
BSDF 99.5 %, MIS 95.3 % (98.9 % within 1 level) of the lit plate pixels; without it 8-40 %; with the other sample2D
draw order and any k: none (this is what settled the draw order, oracle/pt_oracle.cpp sample2D)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
GALLERY = "/root/reference/gallery"
OUT = os.path.join(ROOT, "tests", "golden", "gallery_mis_test2.npz")
SCENE = os.path.join(ROOT, "scenes", "mis_test2.txt")
W = 1080
R0, R1 = 636, 940
KS = list(range(-12, 13))
BAND = 4


def save_transform(img_sum, spp):
    """main.cpp:146-170 + image.cpp:56-67 in float32: sum/spp, ACESFilm, pow(1/2.2), x-flip, (uchar)(clamp*255)"""
    x = img_sum.reshape(W, W, 3) / np.float32(spp)
    a, b, c, d, e = (np.float32(v) for v in (2.51, 0.03, 2.43, 0.59, 0.14))
    y = np.clip((x * (a * x + b)) / (x * (c * x + d) + e), 0, 1).astype(np.float32)
    y = np.power(y, np.float32(1 / 2.2)).astype(np.float32)[:, ::-1, :]
    return (np.clip(y, 0, 1) * np.float32(255)).astype(np.uint8)


def block_means(a, b):
    h, w = a.shape[:2]
    return a.astype(np.float64).reshape(h // b, b, w // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


def fit_slot_shifts(orc, direct20, direct2000):
    conv = direct2000.astype(np.float64)
    ref = direct20.astype(np.float64)
    m = (conv.sum(axis=2) > 30) & (conv.max(axis=2) < 250)
    s = orc.Scene(SCENE)
    raw = np.full((21, W), -128, np.int32)
    for it in range(1, 21):
        cor = np.full((len(KS), W), -1.0)
        for ki, k in enumerate(KS):
            orc.L.orc_set_debug_slot_shift(k)
            img = np.zeros((W * W, 3), np.float32)
            s.pathtrace(orc.MODE_DIRECTLI, it, img)
            out = save_transform(img, 1).astype(np.float64)
            for r in range(R0, R1, BAND):
                mm = m[r:r + BAND]
                if mm.sum() < 200:
                    continue
                cor[ki, r:r + BAND] = np.corrcoef((out[r:r + BAND] - conv[r:r + BAND])[mm].ravel(),
                                                  (ref[r:r + BAND] - conv[r:r + BAND])[mm].ravel())[0, 1]
        has = cor.max(axis=0) > -1
        raw[it, has] = np.array(KS)[cor.argmax(axis=0)][has]
        print("fit iteration", it, {int(v): int(c) for v, c in zip(*np.unique(raw[it, R0:R1], return_counts=True))}, flush=True)
    orc.L.orc_set_debug_slot_shift(0)
    s.close()
    sm = np.zeros((21, W), np.int8)
    for it in range(1, 21):
        for r in range(R0, R1):
            w = raw[it, max(R0, r - 10):min(R1, r + 11)]
            w = w[w != -128]
            if len(w):
                vals, cnt = np.unique(w, return_counts=True)
                sm[it, r] = vals[cnt.argmax()]
        sm[it, R1:] = sm[it, R1 - 1]
    return sm


def fit_deep_shifts(orc, table, mis20_rows, lit_rows, kmin=-16, kmax=16, cols=None):
    """The compacted-slot offset of the reference's run at DEPTH >= 1 (round 4; tools/diag/mis20_residue.py found that the pixels of MIS20.png the
    depth-0 table does not reproduce are exactly those that carry a contribution keyed by a depth >= 1 slot — the survivors of bounce 0 are
    compacted again, and the reference's build flipped other hit / miss decisions there).  Fitted on MIS20.png (Direct20 has no second bounce):
    coordinate descent over the iterations — iteration `it` is re-rendered with every candidate k1 for depth >= 1 (depth 0 keeps `table`), the
    20-iteration sum is re-formed with that iteration replaced, and per 4-row band the k1 with the most bit-identical lit plate pixels wins;
    ties (bands no depth >= 1 contribution reaches) keep the depth-0 offset.  BSDF20.png is NOT used: tests/test_gallery.py checks it out of sample.
    cols = (c0, c1): only the saved image's columns c0..c1-1 vote (round 5: the hold-out table `slot_shift_deep_left` is fitted on the left half and
    MIS20 is scored on the right half, which shares the band's offset but none of the pixels — out-of-sample evidence for the MIS-only terms)."""
    orc.L.orc_set_debug_slot_shift_deep.argtypes = [orc.ci, orc.ci]
    ref = mis20_rows.astype(np.int32)
    lit = lit_rows & (ref.sum(axis=2) > 0)
    if cols is not None:
        lit = lit.copy(); lit[:, :cols[0]] = False; lit[:, cols[1]:] = False
    s = orc.Scene(SCENE)

    def render_it(it, k1_rows):
        one = np.zeros((W, W, 3), np.float32)
        for k0, k1 in np.unique(np.stack([table[it], k1_rows], axis=1), axis=0):
            orc.L.orc_set_debug_slot_shift(int(k0)); orc.L.orc_set_debug_slot_shift_deep(1, int(k1))
            img = np.zeros((W * W, 3), np.float32)
            s.pathtrace(orc.MODE_MIS, it, img)
            rows = (table[it] == k0) & (k1_rows == k1)
            one[rows] = img.reshape(W, W, 3)[rows]
        return one

    def matches(acc):
        return (np.abs(save_transform(acc, 20)[R0:R1].astype(np.int32) - ref).max(axis=2) == 0) & lit

    deep = table.astype(np.int32).copy()
    per_it = [None] + [render_it(it, deep[it]) for it in range(1, 21)]
    for it in range(1, 21):
        rest = np.zeros((W, W, 3), np.float32)
        for j in range(1, 21):
            if j != it:
                rest += per_it[j]
        nb = (R1 - R0) // BAND
        base = matches(rest + per_it[it]).reshape(nb, BAND, W).sum(axis=(1, 2))       # the depth-0 offset at every depth
        best, best_k = base.copy(), np.full(nb, 127, np.int64)                        # 127 = keep table[it]
        cand = {}
        for k1 in range(kmin, kmax + 1):
            one = render_it(it, np.full(W, k1, np.int32))
            m = matches(rest + one).reshape(nb, BAND, W).sum(axis=(1, 2))
            better = m > best
            best[better] = m[better]; best_k[better] = k1
            cand[k1] = one
        rows_k = np.repeat(best_k, BAND)
        rows_k = np.where(rows_k == 127, table[it, R0:R1], rows_k)
        sm = rows_k.copy()
        for r in range(len(rows_k)):                   # 21-row mode filter, as for the depth-0 table
            w = rows_k[max(0, r - 10):r + 11]; vals, cnt = np.unique(w, return_counts=True); sm[r] = vals[cnt.argmax()]
        deep[it, R0:R1] = sm; deep[it, R1:] = sm[-1]
        per_it[it] = render_it(it, deep[it])
        acc = rest + per_it[it]
        print("deep fit iteration", it, {int(v): int(c) for v, c in zip(*np.unique(sm, return_counts=True))}, "exact %.4f" % (matches(acc).sum() / lit.sum()), flush=True)
    orc.L.orc_set_debug_slot_shift(0); orc.L.orc_set_debug_slot_shift_deep(0, 0)
    s.close()
    return deep.astype(np.int8)


def main():
    from PIL import Image
    import oracle_lib as orc
    orc.L.orc_set_debug_slot_shift.argtypes = [orc.ci]
    orc.set_trig_mode(orc.TRIG_LIBM)
    orc.set_threads(os.cpu_count() or 1)
    g = {}
    imgs = {}
    for mode in ("Direct", "BSDF", "MIS"):
        for spp in (20, 2000):
            name = f"{mode}{spp}"
            a = np.asarray(Image.open(os.path.join(GALLERY, name + ".png")).convert("RGB"))
            assert a.shape == (W, W, 3) and a.dtype == np.uint8
            imgs[name] = a
            g["blocks8_" + name] = block_means(a, 8)
            g["crop_" + name] = a[650:714, 500:564].copy()
            if spp == 20:
                g["rows_" + name] = a[R0:R1].copy()
            else:
                g["plate4_" + name] = block_means(a[R0:R1], 4)
    g["lit"] = imgs["Direct2000"][R0:R1].astype(np.int32).sum(axis=2) > 30
    if "--keep-slot-shift" in sys.argv and os.path.exists(OUT):      # re-use the committed depth-0 table (its fit takes ~10 minutes)
        g["slot_shift"] = np.load(OUT, allow_pickle=False)["slot_shift"]
    else:
        g["slot_shift"] = fit_slot_shifts(orc, imgs["Direct20"], imgs["Direct2000"])
    if "--keep-deep" in sys.argv and os.path.exists(OUT):
        g["slot_shift_deep"] = np.load(OUT, allow_pickle=False)["slot_shift_deep"]
    else:
        g["slot_shift_deep"] = fit_deep_shifts(orc, g["slot_shift"].astype(np.int32), g["rows_MIS20"], g["lit"])
    # hold-out: the same fit with only the LEFT half of the saved image (columns 0..539) voting
    g["slot_shift_deep_left"] = fit_deep_shifts(orc, g["slot_shift"].astype(np.int32), g["rows_MIS20"], g["lit"], cols=(0, W // 2))
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
