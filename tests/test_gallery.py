"""Parity against the only outputs the reference itself holds for this path: its six 1080x1080 renders of
scenes/mis_test2.txt, gallery/{Direct,BSDF,MIS}{20,2000}.png (one per SampleMode, preview.h:13; written by saveImage,
main.cpp:146-170).  tests/golden/gallery_mis_test2.npz holds their pixels / block means (generator + the meaning of the
fitted `slot_shift` table: tests/golden/make_gallery_fixture.py).

CPU tier (oracle, libm trig, the reference's nvcc draw order):
  * with the slot-shift tables the oracle reproduces the 20-spp gallery images PIXEL FOR PIXEL on the lit plate pixels:
    Direct >= 99 %, BSDF >= 99 %, MIS >= 98.5 % exact.  `slot_shift` (depth 0) was fitted on Direct20, `slot_shift_deep` (depth >= 1, round 4)
    on MIS20; BSDF20 is out of sample for both.  Measured: 99.39 / 99.40 / 99.01 % (MIS with the depth-0 table alone: 95.06 %).
  * with the OTHER sample2D draw order (the g++ host order) the same table matches nothing (< 20 %): this is what settles
    the draw order.
  * without the table (what the product renders): statistical agreement of 8x8 block means.
GPU tier: the HIP path (ptk_pathtrace + ptk_tonemap through the C ABI) at 20 and 2000 spp against the gallery's block
means / plate pixels with the stated tolerances, and bit-exact against the oracle at 1080x1080 depth 8 in all three modes."""
import os

import numpy as np
import pytest
from conftest import host_threads

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NPZ = os.path.join(ROOT, "tests", "golden", "gallery_mis_test2.npz")
SCENE = os.path.join(ROOT, "scenes", "mis_test2.txt")
W = 1080
R0, R1 = 636, 940
MODES = {"Direct": 1, "BSDF": 0, "MIS": 2}          # SampleMode values of the C ABI / oracle


def _fixture():
    return np.load(NPZ, allow_pickle=False)


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


@pytest.fixture()
def gallery_oracle(orc):
    orc.L.orc_set_debug_slot_shift.argtypes = [orc.ci]
    orc.L.orc_set_sample2d_order.argtypes = [orc.ci]
    orc.set_trig_mode(orc.TRIG_LIBM)
    orc.set_threads(host_threads())
    yield orc
    orc.L.orc_set_debug_slot_shift(0)
    orc.L.orc_set_sample2d_order(1)
    orc.set_threads(1)
    orc.set_trig_mode(orc.TRIG_PTK)


def _render_with_table(orc, mode, table, deep=None):
    """20 iterations; iteration `it`, pixel row r rendered with shading slot + table[it, r] at depth 0 and + deep[it, r] at depth >= 1
    (deep = None: the depth-0 offset at every depth); per-pixel sum in iteration order"""
    orc.L.orc_set_debug_slot_shift_deep.argtypes = [orc.ci, orc.ci]
    s = orc.Scene(SCENE)
    assert (s.width, s.height, s.trace_depth) == (W, W, 8)
    acc = np.zeros((W, W, 3), np.float32)
    dp = table if deep is None else deep
    for it in range(1, 21):
        one = np.zeros((W, W, 3), np.float32)
        for k0, k1 in np.unique(np.stack([table[it], dp[it]], axis=1), axis=0):
            orc.L.orc_set_debug_slot_shift(int(k0)); orc.L.orc_set_debug_slot_shift_deep(1, int(k1))
            img = np.zeros((W * W, 3), np.float32)
            s.pathtrace(mode, it, img)
            rows = (table[it] == k0) & (dp[it] == k1)
            one[rows] = img.reshape(W, W, 3)[rows]
        acc += one
    orc.L.orc_set_debug_slot_shift(0); orc.L.orc_set_debug_slot_shift_deep(0, 0)
    s.close()
    return acc


def _plate_match(out_u8, g, name, cols=None):
    """cols = (c0, c1): score only the saved image's columns c0..c1-1 (the held-out half of MIS20)"""
    ref = g["rows_" + name].astype(np.int32)
    d = np.abs(out_u8[R0:R1].astype(np.int32) - ref).max(axis=2)
    lit = g["lit"] & (ref.sum(axis=2) > 0)
    if cols is not None:
        lit = lit.copy(); lit[:, :cols[0]] = False; lit[:, cols[1]:] = False
    return float((d[lit] == 0).mean()), float((d[lit] <= 1).mean()), int(lit.sum())


@pytest.mark.parametrize("name,min_exact,min_within1", [("Direct", 0.99, 0.995), ("BSDF", 0.99, 0.994), ("MIS", 0.985, 0.993)])
def test_oracle_reproduces_gallery_pixels(gallery_oracle, name, min_exact, min_within1):
    orc = gallery_oracle
    g = _fixture()
    acc = _render_with_table(orc, MODES[name], g["slot_shift"].astype(np.int32), g["slot_shift_deep"].astype(np.int32))
    out = save_transform(acc, 20)
    exact, within1, n = _plate_match(out, g, name + "20")
    print(f"{name}20: {n} lit plate pixels, exact {exact:.4f}, within 1 level {within1:.4f}")
    assert exact >= min_exact and within1 >= min_within1, (exact, within1)
    # the 64x64 raw crop of the fixture (medium light's reflection on the top plate, rows 650..713 x cols 500..563)
    dc = np.abs(out[650:714, 500:564].astype(np.int32) - g["crop_" + name + "20"].astype(np.int32)).max(axis=2)
    print(f"{name}20 crop: exact {(dc == 0).mean():.4f}, within 2 levels {(dc <= 2).mean():.4f}")
    assert (dc <= 2).mean() >= (0.985 if name == "MIS" else 0.99), (dc <= 2).mean()
    # whole image through 8x8 block means (includes the three lights and the black background)
    db = np.abs(block_means(out, 8) - g["blocks8_" + name + "20"])
    assert db.mean() < 0.05 and np.percentile(db, 99.9) < 4.0, (db.mean(), db.max())


def test_mis20_held_out_half(gallery_oracle):
    """VERDICT r4 #6 — out-of-sample evidence for the MIS-only code (power heuristic, next-event add; pathtrace.cu:545-572).  `slot_shift_deep` is
    fitted on MIS20 itself, so MIS20's 99.0 % is in-sample.  `slot_shift_deep_left` is the same fit with ONLY the saved image's columns 0..539
    voting (tests/golden/make_gallery_fixture.py: the offset is one integer per iteration and 4-row band, i.e. 6 080 integers against 164 160
    pixel values in a half); MIS20 is scored here on columns 540..1079, none of whose pixels the fit has seen.  Measured: printed below."""
    orc = gallery_oracle
    g = _fixture()
    acc = _render_with_table(orc, MODES["MIS"], g["slot_shift"].astype(np.int32), g["slot_shift_deep_left"].astype(np.int32))
    out = save_transform(acc, 20)
    fit = _plate_match(out, g, "MIS20", cols=(0, W // 2))
    held = _plate_match(out, g, "MIS20", cols=(W // 2, W))
    print(f"MIS20, depth >= 1 offsets fitted on the left half: fitted half exact {fit[0]:.4f} ({fit[2]} px); HELD-OUT half exact {held[0]:.4f}, within 1 level {held[1]:.4f} ({held[2]} px)")
    assert held[2] > 80000 and held[0] >= 0.98 and held[1] >= 0.992, held


def test_mis20_residue_is_the_depth1_slot_offset(gallery_oracle):
    """Round 3 left MIS20 at 95.06 % exact with the depth-0 table alone against 99.4 % for Direct20 / BSDF20.  tools/diag/mis20_residue.py: the
    differing pixels are those that carry a contribution keyed by a compacted slot of depth >= 1 (15.7 % of the roughness-0.25 plate, whose
    BSDF samples reach the other plates; 0.7 % of the pixels with no such contribution = Direct20's rate) — the survivors of bounce 0 are compacted
    again and the reference's build flipped other decisions there.  With the depth >= 1 offset fitted on MIS20 (fixture `slot_shift_deep`) the
    match is 99.0 %; without it, the old figure."""
    orc = gallery_oracle
    g = _fixture()
    acc = _render_with_table(orc, MODES["MIS"], g["slot_shift"].astype(np.int32))
    exact, within1, n = _plate_match(save_transform(acc, 20), g, "MIS20")
    print(f"MIS20 with the depth-0 offset at every depth: exact {exact:.4f}, within 1 level {within1:.4f}")
    assert 0.94 <= exact <= 0.965, exact


def test_other_draw_order_matches_nothing(gallery_oracle):
    """y-first (the g++ host build's order) with the same table: the noise is unrelated to the gallery's"""
    orc = gallery_oracle
    g = _fixture()
    orc.L.orc_set_sample2d_order(0)
    acc = _render_with_table(orc, MODES["Direct"], g["slot_shift"].astype(np.int32))
    exact, within1, _ = _plate_match(save_transform(acc, 20), g, "Direct20")
    print(f"y-first: exact {exact:.4f}, within 1 level {within1:.4f}")
    assert exact < 0.2


@pytest.mark.parametrize("name", ["Direct", "BSDF", "MIS"])
def test_oracle_unshifted_is_statistically_the_gallery(gallery_oracle, name):
    """what the product computes (no table): same image up to which silhouette pixels flipped -> different noise, same mean"""
    orc = gallery_oracle
    g = _fixture()
    s = orc.Scene(SCENE)
    img = np.zeros((W * W, 3), np.float32)
    for it in range(1, 21):
        s.pathtrace(MODES[name], it, img)
    s.close()
    out = save_transform(img, 20)
    _check_statistical(out, g, name + "20")


# tolerances of the statistical comparison (8-bit levels), per image: (mean |block delta| over lit blocks, global mean delta)
STAT_TOL = {"Direct20": (3.0, 0.05), "MIS20": (3.0, 0.05), "BSDF20": (12.0, 0.1),     # measured (oracle): 1.15/0.006, 1.00/0.005, 5.56/0.015
            "Direct2000": (1.2, 0.15), "MIS2000": (1.2, 0.15), "BSDF2000": (2.5, 0.2)}


def _check_statistical(out_u8, g, name):
    ref_b = g["blocks8_" + name]
    b = block_means(out_u8, 8)
    lit = ref_b.sum(axis=2) > 3
    mad = float(np.abs(b - ref_b)[lit].mean())
    gm = float(np.abs(b.mean(axis=(0, 1)) - ref_b.mean(axis=(0, 1))).max())
    dark = float(np.abs(b - ref_b)[~lit].max()) if (~lit).any() else 0.0
    print(f"{name}: lit-block mean |delta| {mad:.3f} levels, global mean delta {gm:.4f}, darkest-region max delta {dark:.3f}")
    tol_mad, tol_gm = STAT_TOL[name]
    assert mad < tol_mad and gm < tol_gm, (name, mad, gm)
    return mad, gm


# ------------------------------------------------------------------------------------------------------------------
# GPU tier
# ------------------------------------------------------------------------------------------------------------------
def _gpu_render(ptk, mode, iters, **kw):
    import torch
    s = ptk.Scene(SCENE)
    assert (s.width, s.height, s.trace_depth) == (W, W, 8)
    r = ptk.Renderer(s, mode, **kw)
    for it in range(1, iters + 1):
        r.iterate(it)
    img = r.image()
    dev = torch.from_numpy(img).cuda()
    pbo = torch.zeros((W * W, 4), dtype=torch.uint8, device="cuda")
    ptk.tonemap(dev, W, W, iters, pbo)                              # sendImageToPBO's transform (pathtrace.cu:32-65), on the device
    torch.cuda.synchronize()
    out = pbo.cpu().numpy().reshape(W, W, 4)[:, ::-1, :3]            # saveImage's x-flip
    r.close(); s.close()
    return img, np.ascontiguousarray(out)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Direct", "BSDF", "MIS"])
def test_gpu_gallery_20spp(ptk, orc, name):
    g = _fixture()
    img, out = _gpu_render(ptk, MODES[name], 20)
    _check_statistical(out, g, name + "20")
    # device tonemap == saveImage's host transform within 1 level (powf)
    host = save_transform(img, 20)
    assert np.abs(host.astype(np.int32) - out.astype(np.int32)).max() <= 1
    # and the radiance itself is the oracle's, bit for bit (ptk trig), at the gallery's own size
    orc.set_trig_mode(orc.TRIG_PTK); orc.set_threads(host_threads())
    try:
        s = orc.Scene(SCENE)
        ref = np.zeros((W * W, 3), np.float32)
        for it in range(1, 21):
            s.pathtrace(MODES[name], it, ref)
        s.close()
    finally:
        orc.set_threads(1)
    assert np.array_equal(img, ref), int((img != ref).any(axis=1).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Direct", "BSDF", "MIS"])
def test_gpu_gallery_2000spp(ptk, name):
    """converged renders: the HIP path against the reference's 2000-spp images (different silhouette flips -> different
    noise realisation, so the comparison is statistical: block means, and plate pixels through 4x4 means)"""
    g = _fixture()
    img, out = _gpu_render(ptk, MODES[name], 2000, in_flight=3, graph=True)
    _check_statistical(out, g, name + "2000")
    p4 = block_means(out[R0:R1], 4)
    d = np.abs(p4 - g["plate4_" + name + "2000"])
    lit = g["plate4_" + name + "2000"].sum(axis=2) > 3
    frac2 = float((d[lit].max(axis=1) <= 2.0).mean())
    print(f"{name}2000: 4x4 plate blocks within 2 levels: {frac2:.4f}; mean |delta| {d[lit].mean():.3f}")
    # measured: Direct 0.993 / 0.23 levels; BSDF sampling alone rarely finds the small lights, its 2000-spp image is still
    # noisy (the gallery's own too): 0.45 / 2.96 levels
    assert frac2 > (0.35 if name == "BSDF" else 0.97) and d[lit].mean() < (4.5 if name == "BSDF" else 0.6), frac2


@pytest.mark.gpu
@pytest.mark.parametrize("name,min_exact,min_within1", [("Direct", 0.985, 0.995), ("BSDF", 0.985, 0.99), ("MIS", 0.975, 0.99)])
def test_gpu_reproduces_gallery_pixels(tmp_path, name, min_exact, min_within1):
    """VERDICT r2 #7 / r3 #4a: the HIP path against the reference's own pixels DIRECTLY (not through the oracle), in all three sample modes.
    gallery/{Direct,BSDF,MIS}20.png, iteration by iteration, with the slot offsets the reference's GPU build ran with (fixture `slot_shift` at
    depth 0 — fitted on Direct20 with the oracle — and `slot_shift_deep` at depth >= 1, fitted on MIS20; BSDF20 is out of sample for both) —
    summed in iteration order, saveImage's transform in numpy.  The offsets are test hooks (include/ptk_test_hooks.h): the render runs in a
    child process on libptk_hooks.so = the product sources + those entry points.  The device computes sin / cos by the ptk specification, the
    reference by CUDA's sinf / cosf (both <= 2 ulp): a different last bit survives the 8-bit quantisation only on level boundaries.
    Measured on MI355X: printed below (the CPU oracle with libm trig: 99.39 / 99.40 / 99.01 %)."""
    import subprocess, sys
    out_npz = str(tmp_path / f"gallery_{name}.npz")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hooks_child.py"), "gallery", out_npz, name],
                       env=dict(os.environ, PTK_FLAVOUR="hooks"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    acc = np.load(out_npz, allow_pickle=False)["acc"]
    g = _fixture()
    out = save_transform(acc, 20)
    exact, within1, n = _plate_match(out, g, name + "20")
    print(f"HIP path vs gallery/{name}20.png: {n} lit plate pixels, exact {exact:.4f}, within 1 level {within1:.4f}")
    assert exact >= min_exact and within1 >= min_within1, (exact, within1)
    if name == "MIS":
        # out of sample (test_mis20_held_out_half): depth >= 1 offsets fitted on the LEFT half of MIS20 only, the HIP path scored on the right half
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hooks_child.py"), "gallery", out_npz, name, "slot_shift_deep_left"],
                           env=dict(os.environ, PTK_FLAVOUR="hooks"), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        out = save_transform(np.load(out_npz, allow_pickle=False)["acc"], 20)
        held = _plate_match(out, g, "MIS20", cols=(W // 2, W))
        print(f"HIP path vs gallery/MIS20.png, HELD-OUT right half (offsets fitted on the left half): {held[2]} px, exact {held[0]:.4f}, within 1 level {held[1]:.4f}")
        assert held[0] >= 0.975 and held[1] >= 0.99, held
