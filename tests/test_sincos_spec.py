"""The "ptk sincos" specification (oracle/pt_oracle.cpp ptk_sincos, csrc/ptk_device.hpp sincos_ptk): accuracy against
float64 sin/cos on the range the path tracer uses ([0, 2*pi], phi = 2*pi*u), and agreement with glibc sinf/cosf."""
import numpy as np


def _ulp(x):
    return np.spacing(np.abs(x).astype(np.float32)).astype(np.float64)


def test_ptk_sincos_accuracy(orc):
    rng = np.random.default_rng(7)
    x = np.concatenate([rng.uniform(0, 2 * np.pi, 400000), np.linspace(0, 2 * np.pi, 100001),
                        np.arange(0, 9) * (np.pi / 4), [0.0, 6.2831855]]).astype(np.float32)
    s = np.zeros_like(x); c = np.zeros_like(x)
    orc.L.orc_ptk_sincos_array(x.ctypes.data, len(x), s.ctypes.data, c.ctypes.data)
    rs, rc = np.sin(x.astype(np.float64)), np.cos(x.astype(np.float64))
    # absolute error in units of the last place of values near 1 (the direction components are O(1) after scaling)
    assert np.abs(s - rs).max() < 2.0 * 2.0 ** -24 and np.abs(c - rc).max() < 2.0 * 2.0 ** -24
    # relative accuracy away from the zeros of sin/cos: <= 2 ulp
    big = np.abs(rs) > 1e-3
    assert (np.abs(s - rs)[big] / _ulp(rs[big].astype(np.float32))).max() <= 2.0
    big = np.abs(rc) > 1e-3
    assert (np.abs(c - rc)[big] / _ulp(rc[big].astype(np.float32))).max() <= 2.0
    assert np.abs(s * s + c * c - 1).max() < 4e-7
    # agreement with the platform libm the SURVEY probes were produced with
    assert np.abs(s - np.sin(x)).max() <= 3 * 2.0 ** -24 and np.abs(c - np.cos(x)).max() <= 3 * 2.0 ** -24


def test_trig_mode_switch_changes_nothing_structural(orc):
    """Rendering with the ptk spec instead of libm keeps the per-bounce live counts of the pinned cornell probe
    and moves the mean radiance by < 1e-5 relative (the two are the same algorithm up to <= 2 ulp in sin/cos)."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in (orc.TRIG_LIBM, orc.TRIG_PTK):
        orc.set_trig_mode(mode)
        o = orc.Scene(os.path.join(root, "scenes", "cornell.txt"), res=(200, 200), depth=4)
        img = np.zeros((o.P, 3), np.float32); tot = 0
        for it in (1, 2, 3, 4):
            r, live = o.pathtrace(orc.MODE_BSDF, it, img); tot += r
        res[mode] = (tot, img.astype(np.float64).mean(0))
    orc.set_trig_mode(orc.TRIG_PTK)
    assert abs(res[0][0] - res[1][0]) <= 8
    assert np.allclose(res[0][1], res[1][1], rtol=2e-3)
