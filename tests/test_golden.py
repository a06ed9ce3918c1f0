"""Committed golden fixtures (tests/golden/oracle_golden.npz; generator: tests/golden/make_golden.py).
CPU tier: the oracle still computes exactly these values.  GPU tier: the HIP path reproduces them through the C ABI without
the oracle in the loop (camera rays, first-bounce hit records, per-bounce live counts, block-averaged radiance)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "scenes")
NPZ = os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")


def _golden():
    return np.load(NPZ, allow_pickle=False)


def _same(a, b, what):
    assert a.dtype == b.dtype and a.shape == b.shape, (what, a.dtype, b.dtype, a.shape, b.shape)
    if a.dtype.names:
        for name in a.dtype.names:
            assert np.array_equal(a[name], b[name], equal_nan=(a[name].dtype.kind == "f")), f"{what}.{name}"
    else:
        assert np.array_equal(a, b, equal_nan=(a.dtype.kind == "f")), what


def test_oracle_reproduces_the_golden_vectors(orc):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
    try:
        fresh = mg.build()
    finally:
        orc.set_trig_mode(orc.TRIG_LIBM); orc.set_light_fix(False)
    ref = _golden()
    assert set(fresh) == set(ref.files)
    for k in ref.files:
        _same(np.asarray(fresh[k]), ref[k], k)


def _block_average(img, W, H, b):
    a = img.reshape(H, W, 3).astype(np.float64)
    return a.reshape(H // b, b, W // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


@pytest.mark.gpu
def test_gpu_camera_rays_match_golden(ptk):
    import torch
    assert torch.cuda.is_available()
    g = _golden()
    s = ptk.Scene(os.path.join(SC, "cornell.txt"))
    pix = g["g2_pixels"].astype(np.int64)
    bufs = ptk.new_paths(s.width * s.height)
    for k, it in enumerate((1, 2)):
        ptk.generate_rays(s, it, bufs)
        dirs = bufs[1].cpu().numpy().reshape(-1, 4)[pix, :3]
        assert np.array_equal(dirs, g["g2_dirs"][k]), it
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tag,scn,res,depth", [("cornell", "cornell.txt", None, -1), ("bunny", "bunny_cornell.txt", None, -1),
                                               ("objtest", "obj_test.txt", None, -1), ("mis", "mis_test.txt", (1024, 1024), 16)])
def test_gpu_first_bounce_hits_match_golden(ptk, tag, scn, res, depth):
    import torch
    g = _golden()
    s = ptk.Scene(os.path.join(SC, scn), res=res, depth=depth).upload()
    full = ptk.new_paths(s.width * s.height)
    ptk.generate_rays(s, 1, full)
    sel = torch.from_numpy(g[f"g9_{tag}_pixels"].astype(np.int64)).cuda()
    n = int(sel.numel())
    rays = [x[sel].contiguous() for x in full]
    hits = ptk.new_paths(n)
    valid = torch.zeros(n, dtype=torch.int32, device="cuda")
    ptk.intersect(s, n, rays, hits, valid)
    ref = g[f"g9_{tag}_isects"]; rv = g[f"g9_{tag}_valid"]
    assert np.array_equal(valid.cpu().numpy(), rv)
    hit = rv != 0
    tn, pm, ui = (x.cpu().numpy().reshape(-1, 4) for x in hits)
    assert np.array_equal(tn[hit, 0], ref["t"][hit]) and np.array_equal(tn[hit, 1:], ref["n"][hit])
    assert np.array_equal(pm[hit, :3], ref["p"][hit]) and np.array_equal(pm[hit, 3].view(np.int32), ref["materialId"][hit])
    assert np.array_equal(ui[hit, :2], ref["uv"][hit])
    assert np.array_equal(ui[hit, 2].view(np.int32), ref["geomID"][hit]) and np.array_equal(ui[hit, 3].view(np.int32), ref["triangleID"][hit])
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,mname", [(0, "bsdf"), (2, "mis")])
def test_gpu_cornell400_counts_and_radiance_match_golden(ptk, mode, mname):
    g = _golden()
    s = ptk.Scene(os.path.join(SC, "cornell.txt"), res=(400, 400), depth=4)
    r = ptk.Renderer(s, mode)
    lives = []
    for it in range(1, 17):
        r.iterate(it)
        lives.append(r.live_counts()[0].copy())
    img = r.image(); r.close()
    assert np.array_equal(np.stack(lives), g[f"g7_cornell400_{mname}_live"])
    assert np.array_equal(_block_average(img / np.float32(16), 400, 400, 8), g[f"g8_cornell400_{mname}_blocks"])
    assert np.array_equal((img.astype(np.float64).mean(0) / 16).astype(np.float32), g[f"g8_cornell400_{mname}_mean"])


@pytest.mark.gpu
def test_gpu_full_size_counts_match_golden(ptk):
    g = _golden()
    s = ptk.Scene(os.path.join(SC, "cornell.txt"))
    r = ptk.Renderer(s, 0, in_flight=3, graph=True)
    r.iterate(1)
    assert np.array_equal(r.live_counts()[0], g["g7_cornell800_live"])
    assert np.array_equal(_block_average(r.image(), 800, 800, 16), g["g8_cornell800_blocks16"])
    r.close()
    s = ptk.Scene(os.path.join(SC, "mis_test.txt"), res=(512, 512), depth=16)
    r = ptk.Renderer(s, 2)
    lives = []
    for it in (1, 2):
        r.iterate(it); lives.append(r.live_counts()[0].copy())
    assert np.array_equal(np.stack(lives), g["g7_mis512_live"])
    assert np.array_equal(_block_average(r.image() / np.float32(2), 512, 512, 8), g["g8_mis512_blocks"])
    r.close()
