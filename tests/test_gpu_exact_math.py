"""The kernels' exact fast paths (csrc/ptk_device.hpp: div_exact, inv_exact, sqrt_exact, inv_sqrt_exact, div3_exact, the optimistic box_test / sphere_test, and
the packed dot / normalize / length / xform_*) return the SAME BITS as the plain IEEE forms the reference's arithmetic is restated in
(`/`, sqrtf, scalar products and sums in glm's order).  Checked on the device itself — the helpers exist only there — over 2^28
pseudo-random operand sets per seed: arbitrary bit patterns (NaN, inf, denormals, zeros), moderate magnitudes, the edges of the
fast-path ranges (2^-40 / 2^40 for divides, 2^-96 / 2^-80 / 2^80 for square roots), and cube-slab numerators one to three ulps
from zero.  This is what lets the bit-exact parity tests stay bit-exact while divisions share reciprocals and ride packed fp32."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CHECKS = ("div_exact", "inv_exact", "sqrt_exact", "inv_sqrt_exact", "div3_exact", "box_test/sphere_test", "dot/normalize/length", "xform_*")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_exact_fast_paths_equal_ieee(ptk, seed):
    import torch
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    bad = np.zeros(8, np.uint64); smp = np.zeros(128, np.uint32)
    rc = ptk.lib.ptk_debug_exact_math(C.c_ulonglong(1 << 28), seed, bad.ctypes.data, smp.ctypes.data)
    assert rc == 0, rc
    first = [(hex(int(r[0])), r[1:7].view(np.float32).tolist(), [hex(int(x)) for x in r[1:7]], int(r[7])) for r in smp[8:8 + 8 * min(int(smp[0]), 15)].reshape(-1, 8)]
    assert not bad.any(), ({k: int(v) for k, v in zip(CHECKS, bad) if v}, first[:6])
