"""The kernels' exact fast paths (csrc/ptk_device.hpp: div_exact, inv_exact, sqrt_exact, inv_sqrt_exact, div3_exact, the optimistic box_test / sphere_test, and
the packed dot / normalize / length / xform_*) return the SAME BITS as the plain IEEE forms the reference's arithmetic is restated in
(`/`, sqrtf, scalar products and sums in glm's order).  Checked on the device itself — the helpers exist only there — over 2^28
pseudo-random operand sets per seed: arbitrary bit patterns (NaN, inf, denormals, zeros), moderate magnitudes, the edges of the
fast-path ranges (2^-40 / 2^40 for divides, 2^-96 / 2^-80 / 2^80 for square roots), and cube-slab numerators one to three ulps
from zero.  This is what lets the bit-exact parity tests stay bit-exact while divisions share reciprocals and ride packed fp32."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHECKS = ("div_exact", "inv_exact", "sqrt_exact", "inv_sqrt_exact", "div3_exact", "box_test/sphere_test", "dot/normalize/length", "xform_*")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_exact_fast_paths_equal_ieee(tmp_path, seed):
    """(the check kernel is a test hook: it lives in libptk_hooks.so — the product sources + include/ptk_test_hooks.h — and runs in a child)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"exact{seed}.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "hooks_child.py"), "exact_math", out, str(seed)],
                       env=dict(os.environ, PTK_FLAVOUR="hooks"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out, allow_pickle=False)
    bad, smp = d["bad"], d["smp"]
    assert int(d["rc"]) == 0, int(d["rc"])
    first = [(hex(int(r[0])), r[1:7].view(np.float32).tolist(), [hex(int(x)) for x in r[1:7]], int(r[7])) for r in smp[8:8 + 8 * min(int(smp[0]), 15)].reshape(-1, 8)]
    assert not bad.any(), ({k: int(v) for k, v in zip(CHECKS, bad) if v}, first[:6])
