"""Randomised parity (tools/diag/random_parity.py: seeded random scenes — open and closed rooms, rotated overlapping cubes / spheres, all
five material types, sphere / cube / triangle lights, OBJ meshes of 12 / 320 / 10 000 triangles — in a random mode, resolution, depth, with
random iterations in flight, graph replay, tile partition, samples per launch and pipeline): image and live counts == the CPU oracle bit
for bit.  The sweeps recorded in profiles/r3_random_parity.txt ran 1 470 seeds (r5: 9 000 more, profiles/r5_random_parity.txt); this keeps a slice of them in the GPU tier."""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X; there is no CPU fallback"
    torch.cuda.set_device(0)
    return torch


@pytest.fixture(autouse=True)
def _ptk_trig(orc):
    orc.set_trig_mode(orc.TRIG_PTK)
    yield
    orc.set_threads(1)


def _load(meshes):
    os.environ["MESHES"] = "1" if meshes else "0"
    spec = importlib.util.spec_from_file_location("random_parity_%d" % meshes, os.path.join(ROOT, "tools", "diag", "random_parity.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("meshes,first,n", [(0, 1000, 40), (1, 5000, 30)])
def test_random_scenes_equal_the_oracle(ptk, orc, torch_cuda, tmp_path, meshes, first, n):
    m = _load(meshes)
    lit = 0
    for seed in range(first, first + n):
        bad, info, cfg = m.one_case(seed, str(tmp_path))
        assert bad == 0, (seed, info, cfg, bad)
        lit += cfg["lit"] > 0
    assert lit >= 3 * n // 4      # the generator must keep producing scenes that are not black (round 5: its room holds the camera; ~90 % lit)
