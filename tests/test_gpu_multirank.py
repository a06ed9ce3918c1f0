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
        acc = orc.Accumulator(o.P, LANES, SPP, MODE)
        for step in range(1, STEPS + 1):
            rr, _ = acc.batch(o, MODE, step, partition=(world, rank, ROWS)); ref_rays += rr
        total = total + acc.img           # disjoint supports: exact for any reduce order
    assert rays == ref_rays
    bad = np.flatnonzero((img != total).any(axis=1))
    assert bad.size == 0, f"{bad.size} pixels differ, first {bad[:8]}: {img[bad[:4]]} vs {total[bad[:4]]}"
    assert total.sum() > 0


# ---- iteration sharding: the exact multi-GPU validation mode of SURVEY 8e -----------------------------------------------------
IT_STEPS = 6


def _iter_worker(rank, world, port, out_dir, lanes):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
    scene = ptk.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
    r = ptk.Renderer(scene, MODE, in_flight=lanes, graph=True, shard=(world, rank))
    for step in range(1, IT_STEPS + 1):
        r.iterate(step)                      # = iteration (step - 1) * world + rank + 1 of the whole frame
    mine = r.image()
    rays = torch.tensor([r.ray_total()], dtype=torch.int64)
    r.close()
    np.save(os.path.join(out_dir, f"shard{rank}.npy"), mine)
    host = torch.from_numpy(mine.copy())
    dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
    dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    if rank == 0:
        np.save(os.path.join(out_dir, "img.npy"), host.numpy())
        np.save(os.path.join(out_dir, "rays.npy"), rays.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lanes", [1, 2])
def test_two_ranks_iteration_sharded_equals_single_process(tmp_path, ptk, orc, lanes):
    """Rank g of 2 renders iterations g+1, g+3, ... of the whole frame (ptk_set_iteration_sharding) with the reference's own RNG keys
    (pathtrace.cu:148,458,532).  Every shard equals the oracle's running sum of exactly its iterations, bit for bit (one or two of them in
    flight per rank: ordered accumulation), and the SUM-reduced image is shard0 + shard1 (two addends: any reduce order).  The single-process
    render traces the same paths (equal ray totals) and adds the same per-iteration radiances in the order 1, 2, 3, ... instead of
    (1 + 3 + 5 ...) + (2 + 4 + 6 ...): equal up to float association (1e-5 relative, written here)."""
    world = 2
    mp.spawn(_iter_worker, args=(world, _free_port(), str(tmp_path), lanes), nprocs=world, join=True)
    img = np.load(tmp_path / "img.npy"); rays = int(np.load(tmp_path / "rays.npy")[0])
    orc.set_trig_mode(orc.TRIG_PTK)
    o = orc.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
    ref_rays = 0
    shards = []
    for rank in range(world):
        acc = orc.Accumulator(o.P, lanes, 1, MODE)
        for step in range(1, IT_STEPS + 1):
            rr, _ = acc.iteration(o, MODE, (step - 1) * world + rank + 1, step=step); ref_rays += rr
        sh = acc.img
        shards.append(sh)
        assert np.array_equal(np.load(tmp_path / f"shard{rank}.npy"), sh)
    assert rays == ref_rays
    assert np.array_equal(img, shards[0] + shards[1])
    if lanes == 1:
        # the single-process run with iterations_in_flight = 2 (ordered accumulation: the reference's running sum)
        scene = ptk.Scene(os.path.join(ROOT, "scenes", SCENE), res=RES, depth=DEPTH)
        r = ptk.Renderer(scene, MODE, in_flight=2, graph=True)
        for it in range(1, world * IT_STEPS + 1):
            r.iterate(it)
        single = r.image(); single_rays = r.ray_total()
        r.close()
        assert single_rays == rays
        assert np.allclose(img, single, rtol=1e-5, atol=1e-6)


def test_bench_self_launches_two_ranks(tmp_path):
    """`python bench.py --gpus 2` from a plain shell (no torchrun, no WORLD_SIZE): the parent starts the ranks itself before touching
    the GPU, rank 0 prints ONE JSON line, rc 0 (VERDICT r2 missing #1).  gloo because RCCL refuses two ranks on one device."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1",
                        "--iters-per-step", "2", "--res", "128", "128", "--profile-steps", "1", "--cpu-seconds", "0", "--collective", "gather"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "weak"
    m = out["multi_gpu"]
    for k in (m["weak"], m["strong"], m["iteration_sharded"]["weak"], m["iteration_sharded"]["strong"]):
        assert k["value"] > 0
    assert m["iteration_sharded"]["strong"]["calls_per_rank"] == 2 and m["iteration_sharded"]["weak"]["calls_per_rank"] == 4
    # --collective gather: the tile-sharded jobs gathered their own row tiles, the iteration-sharded ones (whole frames everywhere) reduced
    assert m["collective_used"]["tiles/weak"] == "gather" and m["collective_used"]["iterations/weak"] == "reduce" and m["gather_ms"] > 0


def test_bench_nccl_without_enough_gpus_fails_fast():
    """`--backend nccl` with fewer visible GPUs than ranks exits non-zero with one clear line instead of hanging in RCCL (a rank per
    device is a hard requirement there; the one-GPU rehearsals use gloo)."""
    import subprocess
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--backend", "nccl", "--steps", "1", "--warmup", "1", "--res", "64", "64"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert f"needs {n} visible GPUs" in (r.stderr + r.stdout), (r.stderr + r.stdout)[-1500:]


def _gather_worker(rank, world, port, out_dir, res, depth, rows):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")
    scene = ptk.Scene(os.path.join(ROOT, "scenes", "bunny_cornell.txt"), res=res, depth=depth)
    r = ptk.Renderer(scene, 0, partition=(world, rank, rows), in_flight=2, graph=True)
    for step in (1, 2):
        r.iterate(step)
    red = torch.zeros((scene.width * scene.height, 3), dtype=torch.float32, device="cuda")
    r.image_to(red)
    r.close()
    # (a) the full-frame SUM-reduce, (b) the gather of the own row tiles, as bench.py does them
    host = red.cpu(); dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
    counts = [len(ptk.partition_pixels(scene.width, scene.height, world, q, rows).ravel()) for q in range(world)]
    idx = torch.from_numpy(ptk.partition_pixels(scene.width, scene.height, world, rank, rows).ravel().astype("int64"))
    part = torch.zeros((max(counts), 3), dtype=torch.float32); part[:len(idx)] = red.cpu().index_select(0, idx)
    bufs = [torch.zeros_like(part) for _ in range(world)] if rank == 0 else None
    dist.gather(part, bufs, dst=0)
    if rank == 0:
        full = torch.zeros_like(host)
        for q in range(world):
            iq = torch.from_numpy(ptk.partition_pixels(scene.width, scene.height, world, q, rows).ravel().astype("int64"))
            full.index_copy_(0, iq, bufs[q][:counts[q]])
        np.save(os.path.join(out_dir, "reduced.npy"), host.numpy()); np.save(os.path.join(out_dir, "gathered.npy"), full.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_five_rank_rehearsal_at_configs4_rank_size(tmp_path, ptk):
    """a one-GPU gloo rehearsal at the PER-RANK size of BASELINE configs[4] (2048 x 2048 / 8 ranks = 524 288 paths per rank, depth 16,
    8-row tiles, the bunny scene) — with five ranks, the most this pool lets share one card beside the test process itself (six processes per GPU):
    a 2048 x 1280 frame split five ways is the same 524 288 paths per rank.  The gathered own tiles equal the SUM-reduced full frames bit for
    bit (disjoint supports), and rank 3's part equals the single-process render of that part."""
    world, res, depth, rows = 5, (2048, 1280), 16, 8
    mp.spawn(_gather_worker, args=(world, _free_port(), str(tmp_path), res, depth, rows), nprocs=world, join=True)
    red = np.load(tmp_path / "reduced.npy"); gat = np.load(tmp_path / "gathered.npy")
    assert np.array_equal(red, gat) and red.sum() > 0
    scene = ptk.Scene(os.path.join(ROOT, "scenes", "bunny_cornell.txt"), res=res, depth=depth)
    assert ptk.lib.ptk_partition_count(res[0], res[1], world, 3, rows) == 2048 * 2048 // 8
    r = ptk.Renderer(scene, 0, partition=(world, 3, rows), in_flight=2, graph=True)
    for step in (1, 2):
        r.iterate(step)
    mine = r.image(); r.close()
    px = ptk.partition_pixels(res[0], res[1], world, 3, rows).ravel()
    assert np.array_equal(mine[px], red[px]) and mine[px].sum() > 0


def test_cpp_rccl_driver_one_rank(tmp_path, orc):
    """the C++ multi-GPU host path (csrc/main.cpp: ptk_headless --ranks N --rank R --id-file PATH): ncclGetUniqueId through a file,
    ncclCommInitRank, ptk_set_tile_partition, an ncclAllReduce as the barrier, ONE ncclReduce(sum, float32, 3 W H) of the device images and one of
    the ray counters.  On the one-GPU box N = 1: the communicator is created and both collectives EXECUTE (the first RCCL calls this code base
    ever made), and the frame rank 0 saves == the single-process image == the oracle, bit for bit.  A rank count above the visible GPUs is
    refused with one line (two ranks never share a device; RCCL itself rejects that)."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "ptk_headless")
    base = str(tmp_path / "rccl")
    W, H, D, N_IT = 120, 96, 5, 5
    env = {k: v for k, v in os.environ.items() if not k.startswith("PTK_")}
    cmd = [exe, os.path.join(ROOT, "scenes", "cornell.txt"), "--mode", "bsdf", "--iters", str(N_IT), "--res", str(W), str(H), "--depth", str(D), "--out", base,
           "--ranks", "1", "--rank", "0", "--id-file", str(tmp_path / "nccl_id"), "--tile-rows", "8", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("cpp_rccl ")]
    assert len(line) == 1, r.stdout
    rec = json.loads(line[0][len("cpp_rccl "):])
    assert rec["ranks"] == 1 and rec["rank"] == 0 and rec["iterations"] == N_IT - 1 and rec["reduce_ms"] > 0
    img = np.fromfile(base + f".{N_IT}samp.f32", np.float32).reshape(-1, 3)
    orc.set_trig_mode(orc.TRIG_PTK)
    o = orc.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(W, H), depth=D)
    ref = np.zeros((o.P, 3), np.float32); rays = 0
    for it in range(1, N_IT + 1):
        n, _ = o.pathtrace(orc.MODE_BSDF, it, ref)
        rays += n if it > 1 else 0                       # (the counter is read after the warm-up iteration)
    assert ref.sum() > 0 and np.array_equal(img, ref), int((img != ref).any(axis=1).sum())
    assert rec["rays"] == rays
    # more ranks than GPUs: one line, exit code 6, nothing rendered
    ndev = torch.cuda.device_count()
    bad = subprocess.run([*cmd[:cmd.index("--ranks")], "--ranks", str(ndev + 1), "--rank", "0", "--id-file", str(tmp_path / "nccl_id2")],
                         capture_output=True, text=True, timeout=120, env=env)
    assert bad.returncode == 6 and f"{ndev + 1} ranks need {ndev + 1} GPUs, {ndev} visible" in bad.stdout, bad.stdout + bad.stderr
