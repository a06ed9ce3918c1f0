"""CPU tier: the restatement of the product's accumulation policies that the GPU parity tests compare with (oracle_lib.Accumulator) against the
oracle's plain sequential render — the reference's own running sum (pathtrace.cu:398,486,556,572,600)."""
import os

import numpy as np
import pytest


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_ordered_accumulation_is_the_plain_running_sum(orc, scenes_dir, mode):
    """Ordered (the default): with L iterations in flight the image is the plain sequential render in every sample mode — MIS through one
    plane per bounce, which the restatement expresses by adding into the image itself; BSDF / DIRECT initialised contexts that switch to MIS
    add the iteration's own sum (documented exception)."""
    orc.set_trig_mode(orc.TRIG_PTK)
    s = orc.Scene(os.path.join(scenes_dir, "mis_test.txt"), res=(40, 32), depth=5)
    seq = np.zeros((s.P, 3), np.float32)
    acc = orc.Accumulator(s.P, lanes=3, spp=1, mode=mode)
    for it in range(1, 8):
        s.pathtrace(mode, it, seq)
        acc.iteration(s, mode, it)
    assert seq.sum() > 0
    assert np.array_equal(acc.img, seq)
    if mode != orc.MODE_MIS:
        # a context initialised in BSDF / DIRECT has one plane per lane: MIS iterations on it fold their own sums
        sw = orc.Accumulator(s.P, lanes=3, spp=1, mode=mode)
        ref = np.zeros((s.P, 3), np.float32)
        for it in range(1, 6):
            sw.iteration(s, orc.MODE_MIS, it)
            tmp = np.zeros((s.P, 3), np.float32); s.pathtrace(orc.MODE_MIS, it, tmp); ref += tmp
        assert np.array_equal(sw.img, ref)
    s.close()


def test_accumulating_planes_sum_in_lane_order(orc, scenes_dir):
    """ptk_set_ordered_accumulation(0): lane l's plane sums the calls l+1, l+1+L, ...; the image is ((p0 + p1) + p2)."""
    orc.set_trig_mode(orc.TRIG_PTK)
    s = orc.Scene(os.path.join(scenes_dir, "cornell.txt"), res=(40, 32), depth=4)
    lanes = 3
    acc = orc.Accumulator(s.P, lanes=lanes, spp=1, mode=0, ordered=False)
    planes = np.zeros((lanes, s.P, 3), np.float32)
    for it in range(1, 8):
        acc.iteration(s, 0, it)
        s.pathtrace(0, it, planes[(it - 1) % lanes])
    assert np.array_equal(acc.img, (planes[0] + planes[1]) + planes[2])
    assert planes.sum() > 0
    s.close()


def test_samples_per_launch_fold_in_sample_order(orc, scenes_dir):
    """S samples per launch: one plane per sample, folded into the image in sample order when the launch is done (also in MIS: planes per bounce are a
    one-sample-per-launch feature)."""
    orc.set_trig_mode(orc.TRIG_PTK)
    s = orc.Scene(os.path.join(scenes_dir, "cornell.txt"), res=(40, 32), depth=4)
    spp = 3
    for mode in (0, 2):
        acc = orc.Accumulator(s.P, lanes=2, spp=spp, mode=mode)
        ref = np.zeros((s.P, 3), np.float32)
        for step in range(1, 5):
            acc.batch(s, mode, step)
            tmp = np.zeros((spp, s.P, 3), np.float32)
            s.pathtrace_batch(mode, step, spp, tmp)
            for k in range(spp):
                ref += tmp[k]
        assert np.array_equal(acc.img, ref)
    s.close()
