"""Rehearsal of bench.py's N > 1 data path on ONE GPU: two processes share cuda:0, each owns the interleaved row tiles
t % 2 == rank and traces 2 samples of them per launch with 2 iterations in flight (the weak-scaling configuration), then
the full-frame radiance buffers are SUM-reduced to rank 0 (gloo here; RCCL refuses two ranks on one device).  Rank 0's
result must equal the oracle's emulation of exactly that job, bit for bit."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENE, RES, DEPTH, MODE, ROWS, SPP, LANES, STEPS = "cornell.txt", (96, 80), 4, 0, 8, 2, 2, 5


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
    scene = ptk.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
    r = ptk.Renderer(scene, MODE, partition=(world, rank, ROWS), in_flight=LANES, graph=True, spp=SPP)
    for step in range(1, STEPS + 1):
        r.iterate(step)
    red = torch.zeros((scene.width * scene.height, 3), dtype=torch.float32, device="cuda")
    r.image_to(red)
    rays = torch.tensor([r.ray_total()], dtype=torch.int64)
    r.close()
    host = red.cpu()
    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "img.npy"), host.numpy())
        np.save(os.path.join(out_dir, "rays.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_tiles_times_samples(tmp_path, orc):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    img = np.load(tmp_path / "img.npy"); rays = int(np.load(tmp_path / "rays.npy")[0])
    orc.set_trig_mode(orc.TRIG_PTK)
    o = orc.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
    total = np.zeros((o.P, 3), np.float32); ref_rays = 0
    for rank in range(world):
        planes = np.zeros((LANES, SPP, o.P, 3), np.float32)
        for step in range(1, STEPS + 1):
            rr, _ = o.pathtrace_batch(MODE, step, SPP, planes[(step - 1) % LANES], partition=(world, rank, ROWS)); ref_rays += rr
        flat = planes.reshape(LANES * SPP, o.P, 3)
        part = flat[0].copy()
        for p in flat[1:]:
            part = part + p
        total = total + part              # disjoint supports: exact for any reduce order
    assert rays == ref_rays
    assert np.array_equal(img, total)
    assert total.sum() > 0
