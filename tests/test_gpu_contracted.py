"""The opt-in "reference-device numerics" flavour, libptk_fma.so (csrc/Makefile: the same sources with -ffp-contract=fast, IEEE divide
and sqrt kept) — what nvcc's default -fmad=true did to the reference's own device build.  It is reported BESIDE the bit-exact default,
never instead of it (bench.py `contracted`).  Fused multiply-adds change last bits, a changed last bit can flip a hit/miss decision,
and the reference keys its shading RNG by the compacted slot (pathtrace.cu:458,532): after one flip every later path of that bounce
draws different random numbers.  So this flavour is NOT bit-comparable with the oracle; it is held to the tolerances SURVEY 8(d)
states, written here:
  * 1 spp, same seeds: the fraction of pixels within |d| <= 1e-3 (1 + |ref|) and the first pixel outside it are REPORTED ONLY (printed;
    a floor of 50 % is asserted as a sanity check) — after the first flipped hit / miss the comparison is meaningless by construction,
    and which pixel flips first is a property of the contraction, not an error bound;
  * 64 spp: relative L2 of the 8x8-block means <= 2 %, global mean within 0.5 %;
  * 2000 spp of scenes/mis_test2.txt against the reference's own gallery/MIS2000.png and Direct2000.png: the same statistical check as
    the default flavour (tests/test_gallery.py::_check_statistical)."""
import os
import subprocess
import sys

import numpy as np
import pytest
from conftest import host_threads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = os.path.join(ROOT, "tests", "flavour_child.py")


def _child(tmp_path, flavour, *args):
    out = str(tmp_path / f"{flavour}_{args[0].replace('.txt', '')}_{args[4]}.npz")
    env = dict(os.environ, PTK_FLAVOUR=flavour)
    r = subprocess.run([sys.executable, CHILD, out, *[str(a) for a in args]], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out, allow_pickle=False)
    assert str(d["flavour"]) == flavour
    return d


def _blocks(img, W, H, b=8):
    return img.reshape(H // b, b, W // b, b, 3).astype(np.float64).mean(axis=(1, 3))


@pytest.mark.parametrize("scene,W,H,depth,mode", [("cornell.txt", 800, 800, 8, 0), ("mis_test.txt", 512, 512, 8, 2), ("bunny_cornell.txt", 400, 400, 6, 0)])
def test_contracted_flavour_within_stated_tolerance(tmp_path, orc, scene, W, H, depth, mode):
    d = _child(tmp_path, "fma", scene, W, H, depth, mode, "1,64")
    orc.set_trig_mode(orc.TRIG_PTK); orc.set_threads(host_threads())
    try:
        o = orc.Scene(os.path.join(ROOT, "scenes", scene), res=(W, H), depth=depth)
        ref = np.zeros((o.P, 3), np.float32)
        o.pathtrace(mode, 1, ref)
        ref1 = ref.copy()
        for it in range(2, 65):
            o.pathtrace(mode, it, ref)
        o.close()
    finally:
        orc.set_threads(1)
    g1, g64 = d["img1"], d["img64"]
    ok = np.abs(g1 - ref1).max(axis=1) <= 1e-3 * (1 + np.abs(ref1).max(axis=1))
    frac = float(ok.mean())
    first_bad = int(np.argmin(ok)) if not ok.all() else len(ok)
    print(f"{scene}: 1 spp, {frac:.4%} of the pixels within 1e-3 (1 + |ref|); first pixel outside: {first_bad} of {len(ok)}")
    bg, br = _blocks(g64, W, H), _blocks(ref, W, H)
    rel_l2 = float(np.sqrt(((bg - br) ** 2).sum() / (br ** 2).sum()))
    mean_rel = float(abs(g64.astype(np.float64).mean() - ref.astype(np.float64).mean()) / ref.astype(np.float64).mean())
    print(f"{scene}: 64 spp, 8x8-block relative L2 {rel_l2:.4%}, global mean off by {mean_rel:.4%}")
    assert rel_l2 <= 0.02 and mean_rel <= 0.005, (rel_l2, mean_rel)
    assert frac >= 0.5        # sanity only (see the module docstring): a flip-free frame gives ~100 %, one early flip can leave ~50 %


@pytest.mark.parametrize("name", ["Direct", "MIS"])
def test_contracted_flavour_against_the_gallery(tmp_path, name):
    import test_gallery as tg
    d = _child(tmp_path, "fma", "mis_test2.txt", -1, -1, -1, tg.MODES[name], "2000", "tonemap")
    out = np.ascontiguousarray(d["pbo"].reshape(tg.W, tg.W, 4)[:, ::-1, :3])
    tg._check_statistical(out, tg._fixture(), name + "2000")
