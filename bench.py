#!/usr/bin/env python3
"""bench.py — headline benchmark of the path-tracing hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): Mrays/s = (sum over iterations and bounces of live paths entering the intersection stage) /
wall time of the pathtrace() calls, on scenes/cornell.txt, 800x800, depth 8 (configs[1]; SampleMode BSDF).
A "step" is one batch of `--iters-per-step` (default 16) consecutive pathtrace() iterations of the whole frame, i.e. 16 samples
per pixel: the reference renders thousands of iterations per image (its gallery: 20 and 2000), and with iterations in flight a
handful of iterations would mostly time the fill and drain of the pipeline (measured: 64 iterations 16.8, 256 18.1, 1024
18.3 Grays/s).  `ms_per_iteration` is on the line beside `ms_per_step`.

N = 1 : one process, the full frame, one sample per pixel per step.
N > 1 : launched by torch.distributed.run, one rank per GPU.  The frame is sharded into interleaved row tiles
        (tile t of `--tile-rows` rows belongs to rank t % N; scene + BVH replicated).  Every rank runs its own bounce
        loop on its tiles and traces N samples of them per step, so each rank launches W*H paths per step like the single
        GPU does (per-GPU work fixed -> "scaling": "weak"; a step of the job is N samples per pixel of the whole frame).
        After the K steps ONE RCCL reduce (sum, float32, 3*W*H) brings the radiance buffer to rank 0 — inside the timed
        region.  `--scaling strong` keeps one sample per launch instead (per-rank launches shrink by N).

Timed region: barrier + synchronize, K x ptk_pathtrace (no host synchronisation inside), [reduce], synchronize + barrier;
time = MAX over ranks; value = rays of all ranks / time.  The scene, BVH and all state are resident in HBM before the
timed region starts (nothing crosses PCIe inside it).

Also on the JSON line:
  roofline     : the dominant kernel OF THE TIMED PIPELINE priced against the 8 TB/s HBM roofline with its ALGORITHMIC bytes, its device
                 time measured live with HIP events on the render stream in a separate, untimed profiling pass.  Since round 3 the timed
                 pipeline runs one fused launch per bounce, k_shade_trace (shade bounce d-1 + both compactions + trace bounce d): 224 B per
                 path-bounce, SURVEY §8d's whole-bounce figure (intersect 76 + 2 x 8 + shade 132).  Scenes that run the staged kernels
                 (larger meshes in BSDF / DIRECT) get k_intersect (+ the BVH stage) there instead.
  roofline_k_intersect : the staged intersection kernel alone (pipeline 8), 76 B/ray analytic (+ 40 B per BVH node visit + 124 B per
                 triangle test) — the ruler of rounds 1-2 (`roofline` then), kept so that the rounds stay comparable.
  cpu_baseline : the CPU oracle ("port", single thread) timed on rank 0's host on a bounded sample of the same workload.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_RAY_ANALYTIC = 76    # 24 B ray in + 48 B hit record + 4 B flag out
BYTES_PER_NODE, BYTES_PER_TRI = 40, 124


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--iters-per-step", type=int, default=16,
                    help="pathtrace() iterations (samples per pixel of the whole frame) in one step: a step is one batch of this many iterations")
    ap.add_argument("--scene", default="cornell.txt")
    ap.add_argument("--res", type=int, nargs=2, default=None)
    ap.add_argument("--depth", type=int, default=-1)
    ap.add_argument("--mode", default="bsdf", choices=["bsdf", "direct", "mis"])
    ap.add_argument("--tile-rows", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU time of the cpu_baseline sample (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--in-flight", type=int, default=4, help="iterations in flight (ptk_set_iterations_in_flight)")
    ap.add_argument("--graph", type=int, default=1, help="1: replay iterations from a hipGraph")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="N > 1: per-GPU work fixed (weak) or whole-job work fixed (strong)")
    ap.add_argument("--shard", default="tiles", choices=["tiles", "iterations"],
                    help="N > 1: what the headline run shards — interleaved row tiles of the frame (north star) or whole-frame iterations "
                         "(rank g renders iterations g+1, g+1+N, ...: the exact validation mode of SURVEY 8e); the other one is reported in multi_gpu")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--bvh-scene", default="bunny_cornell.txt", help="second roofline measurement on a BVH scene ('' = skip)")
    ap.add_argument("--pipeline", type=int, default=0, help="ptk_set_pipeline value (A/B comparisons)")
    ap.add_argument("--tiles", type=int, default=0, help="> 1: the headline run uses K tiles in flight (ptk_set_tiles_in_flight; bit-identical to one iteration at a time) instead of iterations in flight")
    ap.add_argument("--collective", default="auto", choices=["auto", "reduce", "gather"],
                    help="N > 1, tile sharding: how the radiance reaches rank 0 — SUM-reduce of the full frame, or gather of the owned row tiles (1/N of the bytes); "
                         "auto = reduce unless it costs more than 2 %% of a step")
    ap.add_argument("--cpp-rccl", type=int, default=1, help="1: also run the C++ / RCCL host path (csrc/ptk_headless --ranks N) as a child of every rank, after the headline")
    ap.add_argument("--lean", type=int, default=0, help="1: only the headline run and the roofline pass (no reference_order / cpu_baseline / BVH scene / contracted child)")
    return ap.parse_args()


def host_cores():
    """CPU threads this process may really use: affinity mask, cut by the cgroup quota when there is one, and by 16 (the CPU
    share of a one-GPU box on this pool) when the quota cannot be read."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            tok = open(f).read().split()
            if f.endswith("cpu.max"):
                if tok[0] != "max":
                    quota = int(tok[0]) / int(tok[1])
            else:
                q = int(tok[0])
                if q > 0:
                    quota = q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    if quota is not None:
        n = min(n, max(1, int(quota)))
    else:
        n = min(n, 16)
    return max(1, n)


def cpu_baseline(scene_path, res, depth, mode, seconds):
    """The CPU restatement (oracle/, kind "port") timed on this host on a bounded sample of the same workload: single thread (the
    definition of BASELINE configs[0]; this is `value`), then the same loops over all host cores this process may use (OpenMP over
    paths inside the intersect and shade loops, ordered serial compaction, identical results) as `all_cores`."""
    import numpy as np
    import oracle_lib as orc
    orc.set_trig_mode(orc.TRIG_PTK)
    o = orc.Scene(scene_path, res=res, depth=depth)

    def timed(budget, threads):
        orc.set_threads(threads)
        img = np.zeros((o.P, 3), np.float32)
        stage = np.zeros(4)
        rays = 0; it = 0
        t0 = time.perf_counter()
        while True:
            it += 1
            r, _ = o.pathtrace(mode, it, img, timing=stage)
            rays += r
            el = time.perf_counter() - t0
            if el >= budget or it >= 64:
                break
        orc.set_threads(1)
        return rays, it, el, stage

    rays, it, el, stage = timed(0.6 * seconds, 1)
    out = {"value": round(rays / el / 1e6, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
           "sample": f"{it} iteration(s) of the same workload ({rays} rays) in {el:.1f} s, oracle/pt_oracle.cpp, 1 thread",
           "stage_seconds": {k: round(float(v), 3) for k, v in zip(("generate", "intersect", "shade", "compact"), stage)}}
    cores = host_cores()
    if cores > 1 and seconds > 0.5:
        rays, it, el, _ = timed(0.4 * seconds, cores)
        out["all_cores"] = {"value": round(rays / el / 1e6, 4), "unit": "Mrays/s", "cores": cores,
                            "sample": f"{it} iteration(s) ({rays} rays) in {el:.1f} s, {cores} OpenMP threads, same results"}
    return out


def cpu_baseline_fixed(scene_path, res, depth, mode, iterations, threads):
    """the oracle on a FIXED number of iterations of a workload (BASELINE.md's C1 / C3 definitions), `threads` OpenMP threads"""
    import numpy as np
    import oracle_lib as orc
    orc.set_trig_mode(orc.TRIG_PTK)
    o = orc.Scene(scene_path, res=res, depth=depth)
    orc.set_threads(threads)
    img = np.zeros((o.P, 3), np.float32)
    rays = 0
    t0 = time.perf_counter()
    for it in range(1, iterations + 1):
        r, _ = o.pathtrace(mode, it, img)
        rays += r
    el = time.perf_counter() - t0
    orc.set_threads(1)
    o.close()
    return {"value": round(rays / el / 1e6, 4), "iterations": iterations, "rays": int(rays), "seconds": round(el, 2)}


def measure_kernels(ptk, scene, mode, pipeline, steps):
    """Untimed diagnostic pass on ONE stream (no overlap): per-kernel device time from HIP events on the render stream, plus
    BVH traversal statistics for scenes with triangles."""
    r = ptk.Renderer(scene, mode, pipeline=pipeline, in_flight=1, graph=False)
    for it in (1, 2):
        r.iterate(it)
    r.sync()
    r.profile_begin()
    rays0 = r.ray_total()
    for it in range(3, 3 + steps):
        r.iterate(it)
    prof = r.profile_end()
    rays = r.ray_total() - rays0
    counts, _ = r.live_counts()
    nodes = tris = rays_s = 0
    if scene.n_triangles > 0:
        r.traversal_stats(enable=True)
        rays_s0 = r.ray_total()
        r.iterate(3 + steps)
        nodes, tris = r.traversal_stats()
        rays_s = r.ray_total() - rays_s0
        r.traversal_stats(enable=False)
    r.close()
    return prof, rays, counts, nodes, tris, rays_s


ROUND = next((r for r in ("r5", "r4", "r3") if os.path.exists(os.path.join(ROOT, "profiles", "%s_pmc.json" % r))), "r3")      # the committed profile set the line cites
PMC_FILE = "profiles/%s_pmc.json" % ROUND
PMC_TIMED_FILE = "profiles/%s_pmc_timed.json" % ROUND
BYTES_PER_PATH_BOUNCE = 224   # whole bounce (intersect 76 + 2 x 8 index-only compaction + shade 132) per path-bounce
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4      # wave-instructions / s: 1024 SIMDs, one wave64 instruction per 4 cycles at 2.4 GHz


def pmc_record(scene_file):
    """Committed rocprofv3 PMC passes of the intersect stage (profiles/r3_pmc.json says how they were taken: separate --pmc
    passes with --kernel-trace only; FETCH_SIZE doubled on gfx950 per MI355X_MICROARCH.md).  PMC collection cannot run inside
    this process, so the numbers are read from the committed summary."""
    try:
        return json.load(open(os.path.join(ROOT, PMC_FILE)))["scenes"].get(scene_file)
    except Exception:
        return None


def rocprof_avg_us(scene_file, kernel_substr):
    """average duration of a kernel in the committed rocprofv3 --kernel-trace --stats summaries of the single-stream configuration
    (profiles/<round>_<scene>_kernel_stats_inflight1_p<pipeline>.csv, produced by tools/refresh_profiles.sh with `bench.py --in-flight 1
    --graph 0 --pipeline 0 | 8`: the fused default and the staged kernels); where both summaries hold the kernel, the one with the most
    calls (the 40-call sample of the profiling pass used to shadow the 8 744-call one)"""
    import csv
    best = None
    for p in (0, 8):
        try:
            f = os.path.join(ROOT, "profiles", "%s_%s_kernel_stats_inflight1_p%d.csv" % (ROUND, (scene_file or "").replace(".txt", ""), p))
            for row in csv.DictReader(open(f)):
                if kernel_substr in row["Name"] and (best is None or int(row["Calls"]) > best[0]):
                    best = (int(row["Calls"]), float(row["AverageNs"]) / 1e3)
        except Exception:
            pass
    return best[1] if best else None


def roofline_of(scene, prof, rays_p, counts, nodes, tris, rays_s, steps, scene_file=None):
    ms_int, _ = prof["intersect"]
    has_bvh = scene.n_triangles > 0
    bytes_per_ray = BYTES_PER_RAY_ANALYTIC + ((BYTES_PER_NODE * nodes + BYTES_PER_TRI * tris) / max(rays_s, 1) if has_bvh else 0.0)
    achieved = rays_p * bytes_per_ray / (ms_int * 1e-3) / 1e9 if ms_int > 0 else 0.0
    live_launches = max(int((counts > 0).sum()), 1)          # launches per iteration that had paths to process
    pmc = pmc_record(scene_file) if (scene.width, scene.height) == (800, 800) else None
    traffic = pmc["traffic_bytes_per_launch"] if pmc else None
    avg_us = 1e3 * ms_int / (steps * live_launches)
    rp_us = rocprof_avg_us(scene_file, "k_intersect<false, false>") if (scene.width, scene.height) == (800, 800) else None
    hbm = {"achieved_hbm": round(traffic / (avg_us * 1e-6) / 1e9, 1), "frac_hbm": round(traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
           "achieved_hbm_note": "PMC traffic / this run's average launch time: the rate the HBM really sees (node / triangle re-reads of a "
                                "BVH scene are served by L2); `achieved` is the ALGORITHMIC rate at the reference's struct sizes (SURVEY 8d)"} if traffic else {}
    b_ctr, b_note = bound_from_counters((pmc or {}).get("counters"), traffic, rp_us or avg_us)
    return {
        "bound": b_ctr,
        "bound_note": b_note + ("  (~14 vector instructions per byte moved: see roofline_valu; the walk of a BVH scene adds the dependent chain of its slowest rays, DESIGN.md section 5.2)"),
        "kernel": "k_intersect (the staged pipeline 8's intersection kernel, the kernel SURVEY 8d's 76 B/ray is defined on)" + (" + k_list + k_traverse_wide + k_finalize (the intersect stage)" if has_bvh else ""),
        "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 5), "frac_measured": round(traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None, "traffic": traffic,
        "traffic_unit": "bytes per launch (rocprofv3 PMC, %s)" % PMC_FILE, **hbm,
        "counters": (pmc or {}).get("counters"),
        **({"avg_launch_us_rocprof": round(rp_us, 2), "frac_rocprof": round(bytes_per_ray * rays_p / (steps * live_launches) / (rp_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
            "rocprof_note": "k_intersect's average duration in the committed rocprofv3 summary of the same single-stream configuration "
                            "(profiles/%s_%s_kernel_stats_inflight1_p8.csv); the live HIP-event figure above brackets every launch with two event "
                            "records and so includes ~3 us of dispatch / event latency per launch" % (ROUND, (scene_file or "").replace(".txt", ""))}
           if (rp_us and not has_bvh) else {}),
        "algorithmic_bytes_per_launch": round(bytes_per_ray * rays_p / (steps * live_launches)),
        "bytes_per_ray": round(bytes_per_ray, 2), "rays_per_launch": round(rays_p / (steps * live_launches), 1),
        "avg_launch_us": round(avg_us, 2),
        "nodes_per_ray": round(nodes / max(rays_s, 1), 3) if has_bvh else 0, "tris_per_ray": round(tris / max(rays_s, 1), 3) if has_bvh else 0,
        "note": "algorithmic bytes (SURVEY 8d) x rays / device time of k_intersect, HIP events on the render stream, %d untimed "
                "iterations on one stream; launches with zero live paths excluded from the average; per-kernel ms/iter: %s" % (
                    steps, {k: round(v[0] / steps, 4) for k, v in prof.items()}),
    }


def self_launch(a):
    """`python bench.py --gpus N` from a plain shell: start the N ranks ourselves, BEFORE torch is imported or the GPU is touched in
    this process (the parent never initialises HIP), relay their output (rank 0 prints the JSON line) and return their exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def roofline_fused(scene, prof, rays_p, steps, scene_file=None):
    """The dominant kernel of the benchmarked pipeline: k_shade_trace (shade bounce d-1 + trace bounce d in one launch; its first
    launch of an iteration generates the camera rays instead of shading).  Algorithmic bytes = SURVEY 8(d)'s whole-bounce figure,
    224 B per path-bounce, x the rays the launches traced, / the device time of those launches (HIP events on the render stream)."""
    ms = prof["shade"][0] + prof["generate"][0]
    launches = prof["shade"][1] + prof["generate"][1]
    if ms <= 0 or launches <= 0:
        return None
    achieved = rays_p * BYTES_PER_PATH_BOUNCE / (ms * 1e-3) / 1e9
    avg_us = 1e3 * ms / launches
    rp = rocprof_avg_us(scene_file, "k_shade_trace<0, false, false") if (scene.width, scene.height) == (800, 800) else None
    pm = ((pmc_record(scene_file) or {}).get("k_shade_trace") or {}) if (scene.width, scene.height) == (800, 800) else {}
    alg_per_launch = rays_p * BYTES_PER_PATH_BOUNCE / launches
    # the contract figure is the rocprofv3 one (the committed summary of the same single-stream command); the live HIP-event figure —
    # which brackets every launch with two event records, ~3 us of dispatch / event latency per launch — stays beside it as frac_hip_events
    frac_rp = alg_per_launch / (rp * 1e-6) / 1e9 / HBM_PEAK_GBS if rp else None
    ctr = dict(pm.get("counters") or {})
    try:          # round 5: where the wait goes (tools/pmc_wait.sh)
        wj = json.load(open(os.path.join(ROOT, "profiles", "%s_pmc_wait.json" % ROUND)))["scenes"][scene_file]
        k, gk = wj["k_shade_trace"], wj.get("k_shade_trace_gen", {})
        ctr["wait_split"] = {"source": "profiles/%s_pmc_wait.json" % ROUND, "data_or_barrier": k["waiting_on_data_or_barrier_share_of_wave_cycles"], "to_issue": k["waiting_to_issue_share_of_wave_cycles"],
                             "to_issue_lds": k.get("waiting_to_issue_lds_share_of_wave_cycles", 0), "gen_data_or_barrier": gk.get("waiting_on_data_or_barrier_share_of_wave_cycles", 0),
                             "l2_latency": k.get("mean_l1_to_l2_read_latency_cycles", 0), "l2_hit": k.get("l2_hit_rate", 0)}
    except Exception:
        pass
    bound, bound_note = bound_from_counters(ctr, pm.get("traffic_bytes_per_launch"), rp or avg_us)
    return {"bound": bound, "bound_note": bound_note,
            "kernel": "k_shade_trace (pipeline 0: shade + trace per launch)", "achieved": round((frac_rp * HBM_PEAK_GBS) if frac_rp else achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(frac_rp if frac_rp else achieved / HBM_PEAK_GBS, 5),
            "frac_source": ("algorithmic bytes per launch / k_shade_trace's average duration in profiles/%s_%s_kernel_stats_inflight1_p0.csv (rocprofv3 --kernel-trace --stats of "
                            "`bench.py --lean 1 --in-flight 1 --graph 0`)" % (ROUND, (scene_file or "").replace(".txt", ""))) if frac_rp else "HIP events of this run (no committed rocprofv3 summary for this workload)",
            "frac_hip_events": round(achieved / HBM_PEAK_GBS, 5), "achieved_hip_events": round(achieved, 2),
            "traffic": pm.get("traffic_bytes_per_launch"),
            "traffic_unit": "bytes per launch (rocprofv3 PMC, %s)" % PMC_FILE,
            "frac_rocprof": round(frac_rp, 4) if frac_rp else None,
            # `frac` prices the launch at the reference's struct sizes (224 B per path-bounce); the compact streams move less.  The same launch
            # time against the HBM bytes the counters saw = what the HBM really carries
            "frac_measured": round(pm["traffic_bytes_per_launch"] / ((rp or avg_us) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if pm.get("traffic_bytes_per_launch") else None,
            "frac_measured_note": "PMC traffic per launch / (rocprofv3 average launch time, else this run's) / 8 TB/s",
            "rocprof_note": "the same algorithmic bytes over k_shade_trace's average duration in the committed rocprofv3 summary of the single-stream configuration",
            "algorithmic_bytes_per_launch": round(rays_p * BYTES_PER_PATH_BOUNCE / launches), "bytes_per_path_bounce": BYTES_PER_PATH_BOUNCE,
            "rays_per_launch": round(rays_p / launches, 1), "avg_launch_us": round(avg_us, 2), "avg_launch_us_rocprof": round(rp, 2) if rp else None,
            "counters": pm.get("counters"),
            "note": "HIP events on the render stream, %d untimed iterations on one stream (launches of the sparse tail excluded); per-kernel ms/iter: %s" % (
                steps, {k: round(v[0] / steps, 4) for k, v in prof.items()})}


def bound_from_counters(c, traffic, launch_us):
    """`bound` and its note from the committed counters of the kernel (profiles/<round>_pmc.json), not from a fixed string: the
    HBM share (PMC traffic / launch time / 8 TB/s), the vector-ALU share (SQ_ACTIVE_INST_VALU x 4 / (SQ_BUSY_CYCLES-derived SIMD cycles)) and the
    wave-time split (issuing / waiting on data or a barrier / waiting to issue; with the round-5 pass, the LDS and vector-memory parts of the wait)."""
    if not c:
        return "latency", "no committed counters for this workload; alone on one stream the launch is bound by its fixed latency (DESIGN.md section 5)"
    split = c.get("wave_time_split") or {}
    hbm = traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS if (traffic and launch_us) else None
    valu = c.get("valu_busy_frac_at_2p4GHz")
    wait = split.get("waiting_on_data_or_barrier")
    parts = []
    if hbm is not None: parts.append("HBM carries %.2f of 8 TB/s (counters)" % hbm)
    if valu is not None: parts.append("vector ALUs busy %.2f of the launch at 2.4 GHz" % valu)
    if split: parts.append("a wave's lifetime: issuing %.2f, waiting on data or a barrier %.2f, waiting to issue %.2f" % (split.get("issuing", 0), wait or 0, split.get("waiting_to_issue", 0)))
    ws = c.get("wait_split")
    if ws:
        parts.append("round-5 wait attribution (%s): waiting on data or at a barrier %.2f of the wave's cycles, waiting to issue %.2f (LDS part %.3f); the camera launch — the same trace phase and "
                     "barriers, no dependent loads at its head — waits on data or barriers for %.2f only: the shading launches' wait is their dependent load chain (window tables -> map word -> records; "
                     "L1 -> L2 read latency %.0f cycles at an L2 hit rate of %.2f) and the barrier skew it causes, not the barriers themselves" % (
                         ws.get("source"), ws.get("data_or_barrier", 0), ws.get("to_issue", 0), ws.get("to_issue_lds", 0), ws.get("gen_data_or_barrier", 0), ws.get("l2_latency", 0), ws.get("l2_hit", 0)))
    if hbm is not None and hbm >= 0.6: b = "hbm"
    elif valu is not None and valu >= 0.6: b = "valu"
    else: b = "latency"
    return b, ("priced against the HBM roofline as the contract asks; by the committed counters (%s, one launch alone on one stream): " % PMC_FILE) + "; ".join(parts) + \
              (" — neither pipe is saturated: the launch is bound by stalls (dependent loads and block barriers) at this occupancy" if b == "latency" else "")


def roofline_valu(scene_file, roof):
    """Vector-ALU issue roofline of the kernel `roof` describes: wave-instructions per second (committed PMC pass: SQ_INSTS_VALU per launch /
    this run's average launch time) against 1024 SIMDs x 2.4 GHz / 4 cycles.  tools/diag/issue_cost.hip measured what an instruction
    really costs on gfx950: v_fma / v_mul / v_add_f32 ~3.1 cycles per wave64, v_pk_* f32 ~4.7 (two results), compares / selects / integer
    ~4.3, v_rcp / v_sqrt ~8.2 — so the instruction count is not proportional to issue time (profiles/r3_issue_cost.txt)."""
    c = (roof or {}).get("counters") or {}
    if not c.get("SQ_INSTS_VALU") or not roof.get("avg_launch_us"):
        return None
    rate = c["SQ_INSTS_VALU"] / (roof["avg_launch_us"] * 1e-6)
    return {"achieved": round(rate / 1e9, 2), "peak": round(VALU_ISSUE_PEAK / 1e9, 1), "unit": "G wave-instructions/s", "frac": round(rate / VALU_ISSUE_PEAK, 4),
            "insts_valu_per_launch": c["SQ_INSTS_VALU"], "kernel": roof.get("kernel"),
            "note": "one kernel alone on one stream (its launch also ramps up and drains); with several iterations in flight the issue slots are still not what the pipeline "
                    "is short of (valu_issue_timed.issue_sensitivity).  Peak = 1024 SIMDs x 2.4 GHz nominal / 4; packed instructions counted once; what an instruction really costs: profiles/r3_issue_cost.txt"}


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))
    import torch
    import torch.distributed as dist
    ptk = importlib.import_module("project3-cuda-path-tracer-1_amd")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    N = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if N > 1 and a.backend == "nccl" and torch.cuda.device_count() < N:
        # RCCL needs one device per rank; sharing a card is only possible with the gloo rehearsal backend (fail fast, one clear line)
        raise SystemExit(f"bench.py --gpus {N} --backend nccl needs {N} visible GPUs, this node shows {torch.cuda.device_count()} (use --backend gloo for a one-GPU rehearsal)")
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    if N > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(a.backend, rank=rank, world_size=N)

    mode = {"bsdf": ptk.MODE_BSDF, "direct": ptk.MODE_DIRECTLI, "mis": ptk.MODE_MIS}[a.mode]
    scene_path = os.path.join(ROOT, "scenes", a.scene)
    res = tuple(a.res) if a.res else None
    scene = ptk.Scene(scene_path, res=res, depth=a.depth)
    P = scene.width * scene.height
    red = torch.zeros((P, 3), dtype=torch.float32, device="cuda") if N > 1 else None
    IPS = max(1, a.iters_per_step)
    cdev = "cuda" if a.backend == "nccl" else "cpu"

    def run_job(shard, scaling):
        """One timed job of K steps.  shard "tiles": rank r owns the row tiles t % N == r and traces N (weak) or 1 (strong) samples
        of them per launch.  shard "iterations": rank r renders iterations r+1, r+1+N, ... of the whole frame — K*IPS calls per rank
        (weak: the job renders N*K*IPS iterations) or K*IPS/N calls per rank (strong: the job renders K*IPS iterations).
        Timed region: barrier + synchronize, the calls (no host synchronisation inside), ONE reduce(SUM) of the float32[3*W*H]
        radiance buffer to rank 0, synchronize + barrier; returns rays of all ranks, MAX over ranks of the time, and this rank's
        renderer-side figures."""
        weak = scaling == "weak"
        if N == 1:
            if a.tiles > 1:
                r = ptk.Renderer(scene, mode, pipeline=a.pipeline, tiles=a.tiles, graph=bool(a.graph))
            else:
                r = ptk.Renderer(scene, mode, pipeline=a.pipeline, in_flight=a.in_flight, graph=bool(a.graph))
            calls, spp = a.steps * IPS, 1
        elif shard == "tiles":
            spp = N if weak else 1
            r = ptk.Renderer(scene, mode, partition=(N, rank, a.tile_rows), pipeline=a.pipeline, in_flight=a.in_flight, graph=bool(a.graph), spp=spp)
            calls = a.steps * IPS
        else:
            spp = 1
            r = ptk.Renderer(scene, mode, pipeline=a.pipeline, in_flight=a.in_flight, graph=bool(a.graph), shard=(N, rank))
            calls = a.steps * IPS if weak else max(1, (a.steps * IPS + N - 1) // N)

        def barrier():
            r.sync(); torch.cuda.synchronize()
            if N > 1:
                dist.barrier()

        own = None
        if N > 1 and shard == "tiles":
            # pixels of this rank's row tiles, padded to the largest rank's count (ragged last tile): what a gather moves instead of the full frame
            counts_px = [len(ptk.partition_pixels(scene.width, scene.height, N, q, a.tile_rows).ravel()) for q in range(N)]
            mine = torch.from_numpy(ptk.partition_pixels(scene.width, scene.height, N, rank, a.tile_rows).ravel().astype("int64")).to(red.device)
            own = {"idx": mine, "n": counts_px, "pad": max(counts_px)}

        def reduce_radiance(how="reduce"):
            """the job's one collective.  "reduce": SUM-reduce of the full-frame float32 radiance buffer to rank 0 (zeros outside the own tiles).
            "gather" (tile sharding only): every rank sends the pixels of its own row tiles — 1/N of the bytes — and rank 0 places them."""
            r.image_to(red)
            if how == "gather" and own is not None:
                part = torch.zeros((own["pad"], 3), dtype=torch.float32, device=red.device)
                part[:len(own["idx"])] = red.index_select(0, own["idx"])
                if a.backend != "nccl":
                    part = part.cpu()
                bufs = [torch.zeros_like(part) for _ in range(N)] if rank == 0 else None
                dist.gather(part, bufs, dst=0)
                if rank == 0:
                    for q in range(1, N):
                        iq = torch.from_numpy(ptk.partition_pixels(scene.width, scene.height, N, q, a.tile_rows).ravel().astype("int64")).to(red.device)
                        red.index_copy_(0, iq, bufs[q][:own["n"][q]].to(red.device))
                return
            if a.backend == "nccl":
                dist.reduce(red, dst=0, op=dist.ReduceOp.SUM)
            else:                                  # rehearsal backends: stage through the host
                h = red.cpu(); dist.reduce(h, dst=0, op=dist.ReduceOp.SUM)

        it = 0
        tw0 = time.perf_counter()
        for _ in range(a.warmup * IPS):
            it += 1; r.iterate(it)
        r.sync()
        t_step_est = (time.perf_counter() - tw0) / max(a.warmup, 1)      # one step's time, from the warm-up (first-launch costs included: an upper bound)
        how = "reduce"
        t_red = t_gat = 0.0
        if N > 1:                      # warm the communicator outside the timed region, and time both collectives alone
            for kind in (("reduce", "gather") if own is not None else ("reduce",)):
                reduce_radiance(kind); barrier()
                tt0 = time.perf_counter()
                for _ in range(3):
                    reduce_radiance(kind)
                barrier()
                tt = (time.perf_counter() - tt0) / 3
                if kind == "reduce": t_red = tt
                else: t_gat = tt
            # every rank must call the SAME collective, so the choice is made here — before the timed region, from figures that are
            # identical on every rank (MAX over ranks) — never from a rank-local clock inside it.  "auto": the full-frame reduce stays unless it is
            # worth more than 2 % of a step (estimated from the warm-up) and the gather of the own tiles is cheaper
            tl = torch.tensor([t_red, t_gat, t_step_est], dtype=torch.float64, device=cdev); dist.all_reduce(tl, op=dist.ReduceOp.MAX)
            t_red, t_gat, t_step_est = float(tl[0]), float(tl[1]), float(tl[2])
            if own is not None and a.collective == "gather":
                how = "gather"
            elif own is not None and a.collective == "auto" and t_red > 0.02 * t_step_est and t_gat < t_red:
                how = "gather"
        barrier()
        rays0 = r.ray_total()
        barrier()
        t0 = time.perf_counter()
        r.timer_start()
        for _ in range(calls):
            it += 1; r.iterate(it)
        dev_ms = r.timer_stop()
        if N > 1:
            reduce_radiance(how)           # (the job's one collective; which one was settled before the clock started)
        barrier()
        elapsed = time.perf_counter() - t0
        rays = r.ray_total() - rays0
        counts, _ = r.live_counts()
        r.close()
        if N > 1:
            t = torch.tensor([elapsed, t_red, t_gat], dtype=torch.float64, device=cdev); dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, t_red, t_gat = float(t[0].item()), float(t[1].item()), float(t[2].item())
            q = torch.tensor([rays], dtype=torch.int64, device=cdev); dist.all_reduce(q, op=dist.ReduceOp.SUM); rays = int(q.item())
        return {"rays": rays, "elapsed": elapsed, "dev_ms": dev_ms, "counts": counts, "spp": spp, "calls": calls, "reduce_s": t_red, "gather_s": t_gat, "collective": how}

    head = run_job(a.shard, a.scaling)
    rays, elapsed, dev_ms, counts, spp = head["rays"], head["elapsed"], head["dev_ms"], head["counts"], head["spp"]
    weak = N > 1 and a.scaling == "weak"

    multi = None
    if N > 1:
        # the other three (shard, scaling) combinations and the collective alone, measured after the headline (every rank takes part)
        def line(j):
            return {"value": round(j["rays"] / j["elapsed"] / 1e6, 3), "samples_per_launch": j["spp"], "calls_per_rank": j["calls"],
                    "ms": round(1e3 * j["elapsed"], 3)}
        jobs = {(a.shard, a.scaling): head}
        for sh in ("tiles", "iterations"):
            for sc in ("weak", "strong"):
                if (sh, sc) not in jobs:
                    jobs[(sh, sc)] = run_job(sh, sc)
        multi = {"weak": line(jobs[("tiles", "weak")]), "strong": line(jobs[("tiles", "strong")]),
                 "iteration_sharded": {"weak": line(jobs[("iterations", "weak")]), "strong": line(jobs[("iterations", "strong")]),
                                       "note": "rank g renders iterations g+1, g+1+N, ... of the WHOLE frame with the reference's own RNG keys "
                                               "(ptk_set_iteration_sharding): every launch keeps the single-GPU width; the reduced image is the "
                                               "single-GPU image with N iterations in flight up to the order of the float additions in the reduce"},
                 "unit": "Mrays/s", "headline": f"{a.shard}/{a.scaling}",
                 "reduce_ms": round(1e3 * head["reduce_s"], 4), "gather_ms": round(1e3 * jobs[("tiles", "weak")]["gather_s"], 4),
                 "collective_used": {f"{k[0]}/{k[1]}": v["collective"] for k, v in jobs.items()},
                 "reduce_note": "reduce_ms: one dist.reduce(SUM) of the float32[3*W*H] radiance buffer to rank 0 (+ the device copy into the send buffer); "
                                "gather_ms: dist.gather of the own row tiles only (1/N of the bytes; tile sharding); mean of 3, max over ranks.  One of them is inside "
                                "every timed region once: the reduce, unless it costs more than 2 % of a step and the gather is cheaper (--collective auto)",
                 "note": "tiles (weak / strong): every rank traces N / 1 samples of its interleaved row tiles per launch (per-GPU work fixed / "
                         "whole-frame work fixed, per-rank launches shrink by N); iteration_sharded (weak / strong): K*IPS / K*IPS/N "
                         "whole-frame iterations per rank"}
    # The C++ multi-GPU host path (csrc/main.cpp): one ptk_headless process per rank — ncclCommInitRank through an id file, the tile
    # partition, the pathtrace() calls, ONE ncclReduce of the device images — as a child of every rank, after the headline (never inside its timed
    # region).  Fixed work (1 sample of the own tiles per launch = tiles/strong).  Optional: a failure is recorded, never raised.
    cpp_rccl = None
    exe = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "ptk_headless")
    if a.cpp_rccl and not a.lean and os.path.exists(exe) and (N == 1 or a.backend == "nccl"):
        import subprocess
        idf = "/tmp/ptk_nccl_id_%s_%d" % (os.environ.get("MASTER_PORT", "0"), N)
        try:
            if N > 1: dist.barrier()
            if rank == 0 and os.path.exists(idf): os.remove(idf)
            if N > 1: dist.barrier()
            c = subprocess.run([exe, scene_path, "--mode", a.mode, "--iters", str((a.warmup + a.steps) * IPS), "--warmup", str(a.warmup * IPS), "--no-save",
                                "--ranks", str(N), "--rank", str(local if N > 1 else 0), "--id-file", idf, "--tile-rows", str(a.tile_rows)]
                               + (["--res", str(res[0]), str(res[1])] if res else []) + (["--depth", str(a.depth)] if a.depth >= 0 else []),
                               capture_output=True, text=True, timeout=300 if N == 1 else 120)      # (N > 1 has never run on hardware: a communicator that does not come up costs two minutes, not the run)
            j = json.loads([l for l in c.stdout.splitlines() if l.startswith("cpp_rccl ")][-1][len("cpp_rccl "):])
            cpp_rccl = {"value": round(j["rays"] / j["seconds"] / 1e6, 3), "unit": "Mrays/s", "ranks": j["ranks"], "reduce_ms": j["reduce_ms"], "iterations": j["iterations"],
                        "config": "csrc/ptk_headless --ranks N --rank r --id-file F: C++ host, one process per GPU, RCCL (ncclCommInitRank, ncclAllReduce barrier, one "
                                  "ncclReduce(sum, float32, 3*W*H) of the device images + one of the ray counters); interleaved row tiles, 1 sample per launch (fixed work); "
                                  "the clock covers the pathtrace() calls and the reduce"}
        except Exception as e:
            cpp_rccl = {"error": str(e)[:200]}
        if N > 1:
            try: dist.barrier()
            except Exception: pass
    if rank == 0:
        # what ptk_set_pipeline(0) picks (include/ptk.h, pathtrace.hip use_trace_pipeline): the fused launches for triangle-free scenes, for meshes
        # of up to 1 024 triangles (walked inside the kernel) and, on larger meshes, in MIS mode
        fused = a.pipeline == 10 or (a.pipeline == 0 and (scene.n_triangles <= 1024 or a.mode == "mis"))
        staged = 8 if fused else a.pipeline        # k_intersect on its own: the kernel SURVEY 8d's 76 B / ray figure is defined on
        prof, rays_p, counts_p, nodes, tris, rays_s = measure_kernels(ptk, scene, mode, staged, a.profile_steps)
        roofline = roofline_of(scene, prof, rays_p, counts_p, nodes, tris, rays_s, a.profile_steps, a.scene)
        extra = {}
        if fused:
            pf = measure_kernels(ptk, scene, mode, 10, a.profile_steps)
            # `roofline` = the dominant kernel of the pipeline that was timed; the staged k_intersect keeps its own key (rounds 1-2's `roofline`)
            extra["roofline_k_intersect"] = roofline
            extra["roofline_valu_k_intersect"] = roofline_valu(a.scene, roofline)
            roofline = roofline_fused(scene, pf[0], pf[1], a.profile_steps, a.scene) or roofline
        extra["roofline_valu"] = roofline_valu(a.scene, roofline)
        if N == 1 and not a.lean and a.bvh_scene and a.bvh_scene != a.scene:
            # the HBM-fraction target is only meaningful where the kernel streams BVH data: same measurement
            # on the BASELINE configs[2] timed scene (bunny_cornell.txt, 800x800 depth 8)
            bs = ptk.Scene(os.path.join(ROOT, "scenes", a.bvh_scene))
            rb = ptk.Renderer(bs, mode, pipeline=a.pipeline, in_flight=a.in_flight, graph=bool(a.graph))
            for i in range(1, 5):
                rb.iterate(i)
            rb.sync(); rb0 = rb.ray_total(); rb.timer_start()
            for i in range(5, 5 + 128):
                rb.iterate(i)
            bms = rb.timer_stop(); brays = rb.ray_total() - rb0
            rb.close()
            pb = measure_kernels(ptk, bs, mode, staged, a.profile_steps)
            extra["roofline_bvh_scene"] = dict(roofline_of(bs, *pb, a.profile_steps, a.bvh_scene), workload=f"scenes/{a.bvh_scene} {bs.width}x{bs.height} depth {bs.trace_depth}",
                                               mrays_per_s=round(brays / bms / 1e3, 1))
        ref_order = None
        if N == 1 and not a.lean and (a.in_flight != 1 or a.graph):
            # the reference's own schedule: one iteration at a time, eager launches.  (The headline overlaps `in_flight` iterations, each computed
            # unchanged into its own planes, which are folded into the image in the reference's order of adds: the image is bit-identical to this
            # schedule's in every sample mode, tests/test_gpu_parity.py::test_iterations_in_flight_keep_the_reference_order)
            rr = ptk.Renderer(scene, mode, pipeline=a.pipeline, in_flight=1, graph=False)
            for i in range(1, a.warmup * IPS + 1):
                rr.iterate(i)
            rr.sync(); q0 = rr.ray_total(); tt0 = time.perf_counter()
            for i in range(a.warmup * IPS + 1, (a.warmup + a.steps) * IPS + 1):
                rr.iterate(i)
            rr.sync(); tt = time.perf_counter() - tt0; qr = rr.ray_total() - q0
            rr.close()
            ref_order = {"value": round(qr / tt / 1e6, 3), "unit": "Mrays/s", "ms_per_step": round(1e3 * tt / a.steps, 4), "ms_per_iteration": round(1e3 * tt / (a.steps * IPS), 4),
                         "config": "iterations_in_flight 1, graph_replay false: the reference's accumulation order, bit-identical image"}
        if ref_order is not None and a.tiles <= 1:
            # the same accumulation order — the image is bit-identical to reference_order's — with three bands of rows in flight (ptk_set_tiles_in_flight)
            rt = ptk.Renderer(scene, mode, pipeline=a.pipeline, tiles=3, graph=True)
            for i in range(1, a.warmup * IPS + 1):
                rt.iterate(i)
            rt.sync(); q0 = rt.ray_total(); tt0 = time.perf_counter()
            for i in range(a.warmup * IPS + 1, (a.warmup + a.steps) * IPS + 1):
                rt.iterate(i)
            rt.sync(); tt = time.perf_counter() - tt0; qr = rt.ray_total() - q0
            rt.close()
            ref_order["tiles_in_flight"] = {"value": round(qr / tt / 1e6, 3), "unit": "Mrays/s", "ms_per_iteration": round(1e3 * tt / (a.steps * IPS), 4),
                                            "config": "tiles_in_flight 3, graph_replay true: the reference's accumulation order and RNG keys, bit-identical image (tests/test_gpu_parity.py::test_tiles_in_flight*)"}
        if ref_order is not None and a.tiles <= 1 and a.in_flight > 1:
            # what the ordered accumulation costs: the headline's configuration with planes that accumulate instead (ptk_set_ordered_accumulation(0):
            # image = ((p0 + p1) + p2), float association differs from the reference's) — rounds 2 and 3 timed THIS form
            ru = ptk.Renderer(scene, mode, pipeline=a.pipeline, in_flight=a.in_flight, graph=bool(a.graph), ordered=0)
            for i in range(1, a.warmup * IPS + 1):
                ru.iterate(i)
            ru.sync(); q0 = ru.ray_total(); tt0 = time.perf_counter()
            for i in range(a.warmup * IPS + 1, (a.warmup + a.steps) * IPS + 1):
                ru.iterate(i)
            ru.sync(); tt = time.perf_counter() - tt0; qr = ru.ray_total() - q0
            ru.close()
            ref_order["accumulating_planes"] = {"value": round(qr / tt / 1e6, 3), "unit": "Mrays/s", "ms_per_iteration": round(1e3 * tt / (a.steps * IPS), 4),
                                                "config": f"iterations_in_flight {a.in_flight}, graph_replay {bool(a.graph)}, ptk_set_ordered_accumulation(0): the form rounds 2-3 timed; image = ((p0 + p1) + p2), NOT the reference's association"}
        cpp_api = None
        exe = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "ptk_headless")
        if N == 1 and not a.lean and os.path.exists(exe):
            # the drop-in C++ API (pathtraceInit / pathtrace / pathtraceFree of src/pathtrace.h:6-9, csrc/pathtrace.h) driven by the headless replay of
            # main.cpp's loop, own process: wall time of the pathtrace() calls + ONE pathtraceResolve() (the host image for saveImage), SURVEY 8(d)'s
            # definition of the metric.  `value`: the API's defaults — four iterations in flight + graph replay with ordered accumulation (image
            # bit-identical to one iteration at a time); `one_at_a_time`: --in-flight 1 --graph 0, the
            # reference's own schedule; `overlapped`: the knobs set explicitly; `eager`: a host image after EVERY call like pathtrace.cu:809-810
            import subprocess
            def headless(*flags):
                c = subprocess.run([exe, scene_path, "--mode", a.mode, "--iters", str((a.warmup + a.steps) * IPS), "--warmup", str(a.warmup * IPS), "--no-save", *flags]
                                   + (["--res", str(res[0]), str(res[1])] if res else []) + (["--depth", str(a.depth)] if a.depth >= 0 else []),
                                   capture_output=True, text=True, timeout=600)
                j = json.loads([l for l in c.stdout.splitlines() if l.startswith("cpp_api ")][-1][len("cpp_api "):])
                return {"value": round(j["rays"] / j["seconds"] / 1e6, 3), "ms_per_iteration": round(1e3 * j["seconds"] / j["iterations"], 4)}
            try:
                cpp_api = dict(headless(), unit="Mrays/s",
                               config=("csrc/ptk_headless = the reference's loop, unedited: pathtraceInit + %d x pathtrace(nullptr, 0, it), the last call makes state.image current; API defaults (" % (a.steps * IPS))
                                      + "4 iterations in flight + graph replay, ordered accumulation = the reference's running sum, bit-identical image)",
                               one_at_a_time=headless("--in-flight", "1", "--graph", "0"),
                               overlapped=headless("--in-flight", str(a.in_flight), "--graph", str(a.graph)),
                               eager=headless("--in-flight", "1", "--graph", "0", "--eager"),
                               eager_note="a resolve (7.7 MB D2H + host wait) after every call = what the reference's pathtrace() does (pathtrace.cu:809-810)")
            except Exception as e:
                cpp_api = {"error": str(e)[:200]}
        cpu = None
        cpu_c1 = cpu_bvh = None
        if N == 1 and not a.lean and a.cpu_seconds > 0:
            cpu = cpu_baseline(scene_path, res, a.depth, mode, a.cpu_seconds)
            # BASELINE.md section 2, C1 = BASELINE.json configs[0]: cornell.txt 400x400 depth 4, 16 iterations, single thread, BSDF and MIS
            c1 = {}
            for mname, mval in (("bsdf", ptk.MODE_BSDF), ("mis", ptk.MODE_MIS)):
                c1[mname] = cpu_baseline_fixed(os.path.join(ROOT, "scenes", "cornell.txt"), (400, 400), 4, mval, 16, 1)
            cpu_c1 = dict(c1, unit="Mrays/s", cores=1, kind="port",
                          workload="scenes/cornell.txt 400x400 depth 4, 16 iterations (BASELINE configs[0]), oracle/pt_oracle.cpp, 1 thread; MIS counts the same "
                                   "paths x bounces (shadow rays not counted, SURVEY 8d)")
            if a.bvh_scene:
                # C3: the BVH scene on all host cores (and one iteration on a single thread)
                cores = host_cores()
                cpu_bvh = {"all_cores": dict(cpu_baseline_fixed(os.path.join(ROOT, "scenes", a.bvh_scene), None, -1, mode, 4, cores), cores=cores),
                           "one_thread": dict(cpu_baseline_fixed(os.path.join(ROOT, "scenes", a.bvh_scene), None, -1, mode, 1, 1), cores=1),
                           "unit": "Mrays/s", "kind": "port", "workload": f"scenes/{a.bvh_scene} as shipped in this repo (800x800 depth 8), SampleMode {a.mode.upper()}"}
        contracted = None
        # (never under a profiler: the child would inherit the preload, contend for the counters and land in the trace — ADVICE r3)
        profiled = any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
        if N == 1 and not a.lean and not profiled and ptk.FLAVOUR == "exact" and os.path.exists(os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "libptk_fma.so")):
            # the opt-in "reference-device numerics" flavour (libptk_fma.so: -ffp-contract=fast like nvcc's -fmad=true), same workload, own
            # process (one flavour per process), reported beside the bit-exact headline, never instead of it
            import subprocess
            try:
                c = subprocess.run([sys.executable, os.path.abspath(__file__), "--lean", "1", "--steps", str(a.steps), "--warmup", str(a.warmup), "--scene", a.scene,
                                    "--mode", a.mode, "--in-flight", str(a.in_flight), "--graph", str(a.graph), "--iters-per-step", str(IPS), "--profile-steps", str(a.profile_steps)]
                                   + (["--res", str(res[0]), str(res[1])] if res else []) + (["--depth", str(a.depth)] if a.depth >= 0 else []),
                                   env=dict(os.environ, PTK_FLAVOUR="fma"), capture_output=True, text=True, timeout=600)
                cj = json.loads([l for l in c.stdout.splitlines() if l.startswith("{")][-1])
                pm = (pmc_record(a.scene) or {}).get("contracted_fma") or {}
                contracted = {"value": cj["value"], "unit": "Mrays/s", "ms_per_iteration": cj["ms_per_iteration"], "frac": cj["roofline"]["frac"],
                              "frac_note": "same kernel, algorithmic bytes and HIP-event launch time as roofline.frac",
                              "valu_per_wave": pm.get("valu_per_launched_wave"), "valu_per_wave_exact": pm.get("valu_per_launched_wave_exact"),
                              "flavour": cj.get("flavour"),
                              "note": "libptk_fma.so = the same sources with -ffp-contract=fast (IEEE divide / sqrt kept); not bit-comparable with the oracle, "
                                      "held to SURVEY 8(d)'s tolerances by tests/test_gpu_contracted.py"}
            except Exception as e:       # the flavour is optional: its absence never fails the headline
                contracted = {"error": str(e)[:200]}
        out = {
            "metric": "Mrays/s (paths x bounces / s)", "value": round(rays / elapsed / 1e6, 3), "unit": "Mrays/s",
            "n_gpus": N, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
            "higher_is_better": True, "scaling": "weak" if (weak or N == 1) else "strong", "vs_baseline": None, "dtype": "f32",
            "data": f"the reference's own scene file scenes/{a.scene} (verbatim copy, parsed by this build's loader); no synthetic inputs, no dataset",
            "config": {"workload": f"scenes/{a.scene} {scene.width}x{scene.height} depth {scene.trace_depth}, SampleMode {a.mode.upper()}"
                                   + (" (BASELINE configs[1])" if a.scene == "cornell.txt" and res is None and a.depth < 0 and a.mode == "bsdf" else ""),
                       "iterations_per_step": IPS, "rays_per_step": rays // max(a.steps, 1), "live_paths_per_bounce_last_step_rank0": [int(c) for c in counts],
                       "sharding": "none" if N == 1 else ((f"interleaved row tiles of {a.tile_rows} rows (tile % {N} == rank), {spp} sample(s) of the owned pixels per launch"
                                                           if a.shard == "tiles" else f"whole-frame iterations, rank g renders iterations g+1, g+1+{N}, ...")
                                                          + ", one RCCL reduce(SUM) of the float32[3*W*H] radiance buffer in the timed region"),
                       "iterations_total": head["calls"] * (N if a.shard == "iterations" else 1) * spp,
                       "iterations_in_flight": a.in_flight if a.tiles <= 1 else 1, "tiles_in_flight": max(a.tiles, 1), "graph_replay": bool(a.graph),
                       "accumulation": ("in place: the reference's own order" if (a.in_flight <= 1 and spp <= 1) or a.tiles > 1 else
                                        "ordered: every iteration adds into planes of its own (one; one per bounce in MIS) that are folded into the image in the reference's "
                                        "order of adds (one k_merge_planes launch per iteration, inside the timed region): the image is bit-identical to one iteration at a "
                                        "time (reference_order)" + ("" if spp <= 1 or a.mode != "mis" else " — except MIS with samples per launch: one plane per sample")),
                       "pipeline": ("one launch per bounce: k_shade_trace shades bounce d-1 and traces bounce d (segmented streams, physical-slot hit records, window hit "
                                    "prefixes); device-resident counts, no host sync") if fused else
                                   "per bounce: intersect stage / shade with block-local compaction / window prefix (segmented streams); device-resident counts, no host sync",
                       "step_definition": "one step = iterations_per_step pathtrace() iterations (1 in round 1, 16 since round 2): compare ms_per_iteration across rounds, not ms_per_step"},
            "device_ms_per_step": round(dev_ms / a.steps, 4), "ms_per_iteration": round(1e3 * elapsed / (a.steps * IPS), 4),
            "overlap_note": (f"{a.in_flight} iterations in flight on {a.in_flight} streams: ms_per_step is smaller than the sum of the "
                             "per-kernel times of one iteration (roofline.note) because iterations overlap") if a.in_flight > 1 else None,
            "reference_order": ref_order, "cpp_api": cpp_api, "cpp_rccl": cpp_rccl,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_c1": cpu_c1, "cpu_baseline_bvh_scene": cpu_bvh, "contracted": contracted, "flavour": ptk.FLAVOUR,
        }
        if multi is not None:
            out["multi_gpu"] = multi
            # north_star's ">= 6x at 8 GPUs on cornell.txt 800x800 depth 8" is a FIXED-WORK statement.  `value` is the headline the
            # contract's "scaling" key describes (weak by default: per-GPU work fixed); the fixed-work figure of the same job sits beside it
            out["strong_scaling"] = {"value": multi["strong"]["value"], "unit": "Mrays/s", "what": "tiles/strong: the whole frame's work fixed, every rank traces 1 sample of "
                                     "its own row tiles per launch (launches shrink by N) — the curve north_star's '>= 6x at 8 GPUs' is about",
                                     "headline_is": multi["headline"]}
        if N == 1:
            # utilisation from counters taken IN the timed configuration (tools/pmc_timed.sh: one rocprofv3 --pmc pass over `bench.py --lean 1`,
            # summed over every dispatch), no assumed clock: VALU-active quad-cycles x 4 / (SIMDs x GRBM_GUI_ACTIVE / 8 XCDs) — a fraction of the cycles the
            # GPU was active in that very run, <= 1 by construction.  Read from the committed summary (PMC collection cannot run inside this process).
            try:
                pt = json.load(open(os.path.join(ROOT, PMC_TIMED_FILE)))
                out["valu_busy_timed"] = {k: pt[k] for k in ("frac", "valu_active_cycles", "gpu_active_cycles_per_xcd", "simds", "effective_clock_GHz", "dispatches", "note", "source") if k in pt}
            except Exception:
                pass
            # the roofline of the TIMED pipeline (not of one kernel alone): algorithmic and counter bytes over the timed wall time
            rec = pmc_record(a.scene) if (scene.width, scene.height) == (800, 800) and res is None and a.depth < 0 else None
            alg = rays * BYTES_PER_PATH_BOUNCE / elapsed / 1e9
            roofline["frac_timed"] = round(alg / HBM_PEAK_GBS, 4)
            roofline["frac_timed_note"] = "algorithmic bytes of ALL rays of the timed region (224 B per path-bounce) / its wall time / 8 TB/s: the whole pipeline with its overlap, where `frac` is one launch alone on one stream"
            if rec and fused:
                # How close the TIMED configuration is to vector-ALU ISSUE (rounds 3 and 5).  Vector instructions of one iteration's launches (committed
                # single-stream PMC pass: SQ_INSTS_VALU per dispatch of k_shade_trace x traceDepth + the camera launch) x cycles per wave64 instruction /
                # (1024 SIMDs x 2.4 GHz) against the timed ms_per_iteration — 0.94 at a flat 4 cycles, 0.73 at the 3.1 an fma / mul / add was measured at —
                # AND the direct experiment (tools/diag/issue_sensitivity.sh, profiles/<round>_issue_sensitivity.json): diagnostic builds with 10 / 19 / 39 %
                # more vector instructions (pure fma chains) run 0.9 / 4.2 / 10.9 % longer, the contracted flavour's 5 % fewer buy nothing.  An issue-bound
                # pipeline would pay for every added instruction; this one pays 0.14-0.39 of their issue time: the SIMDs have idle issue slots, what the timed
                # configuration is short of is wave slots x wave lifetime (seven blocks per CU, each a serial chain of dependent loads, barriers and its own
                # ~1 600 vector instructions).  Every figure recomputable from the named files.
                try:
                    vi = rec["k_shade_trace"]["counters"]["SQ_INSTS_VALU"] * scene.trace_depth + rec["k_shade_trace_gen"]["counters"]["SQ_INSTS_VALU"]
                    t_issue_us = vi * 4 / (1024 * 2.4e9) * 1e6
                    sens = None
                    try:
                        sj = json.load(open(os.path.join(ROOT, "profiles", "%s_issue_sensitivity.json" % ROUND)))
                        sens = {k: {f: {q: r[q] for q in ("instructions", "time", "paid_share_of_issue_time") if q in r} for f, r in sj[k].items() if f != "exact"} for k in ("timed", "one_at_a_time") if k in sj}
                        sens["source"] = "profiles/%s_issue_sensitivity.json (tools/diag/issue_sensitivity.sh: diagnostic builds, alternating child processes)" % ROUND
                    except Exception:
                        pass
                    out["valu_issue_timed"] = {"frac": round(t_issue_us / (1e3 * out["ms_per_iteration"]), 4), "frac_at_measured_fma_cost": round(t_issue_us * 3.1 / 4 / (1e3 * out["ms_per_iteration"]), 4),
                                               "bound": "not issue: wave slots x wave lifetime (see issue_sensitivity)",
                                               "vector_instructions_per_iteration": int(vi), "issue_time_us_per_iteration": round(t_issue_us, 1), "ms_per_iteration": out["ms_per_iteration"],
                                               "peak": "1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 614 G wave-instructions/s (frac); / 3.1 cycles, what v_fma / v_mul / v_add_f32 cost "
                                                       "in profiles/r3_issue_cost.txt (frac_at_measured_fma_cost)",
                                               "source": "%s: SQ_INSTS_VALU per dispatch of k_shade_trace x %d + k_shade_trace_gen" % (PMC_FILE, scene.trace_depth),
                                               "issue_sensitivity": sens,
                                               "note": "the instruction count priced at a flat cost says 0.73-0.94 of the issue slots; the direct experiment says the slots are NOT what the timed "
                                                       "configuration is short of: +10 / +19 / +39 %% vector instructions (fma chains, no memory) cost +0.9 / +4.2 / +10.9 %% time — 0.14-0.39 of the added issue "
                                                       "time, where an issue-bound pipeline pays all of it — and the contracted flavour's -5 %% instructions buy nothing (issue_sensitivity).  The limit is "
                                                       "occupancy x latency: seven blocks per CU (72 VGPRs, 22 KB of LDS), each a serial chain — four dependent load round trips at its head, 18 barriers, its "
                                                       "own instruction stream — so what pays is a shorter chain (k_trace_shade with a free prefix: +11 / +21 %%; instruction scheduling for memory clauses: "
                                                       "+2-5 %%) or another wave slot (5 -> 6 -> 7 waves per SIMD: +8 %%, +6 %%), not fewer instructions or bytes.  profiles/%s_pmc_timed.json's figure is the "
                                                       "SERIALISED pace: rocprofv3 runs one dispatch at a time while it collects counters" % ROUND}
                    roofline["bound_timed"] = "latency"
                    roofline["bound_note"] = ("TIMED configuration (%d iterations in flight): neither HBM (frac_timed_measured) nor vector-ALU issue (valu_issue_timed.issue_sensitivity) is "
                                              "saturated — wave slots x wave lifetime.  One launch ALONE on one stream (what frac prices): %s" % (a.in_flight, roofline["bound_note"]))
                except Exception as e:
                    out["valu_issue_timed"] = {"error": str(e)[:200]}
                per_iter = 0.0
                for k, mult in (("k_shade_trace", scene.trace_depth), ("k_shade_trace_gen", 1)):
                    t = (rec.get(k) or {}).get("traffic_bytes_per_launch") if isinstance(rec.get(k), dict) else None
                    per_iter += (t or 0) * mult
                if per_iter > 0:
                    roofline["frac_timed_measured"] = round(per_iter / (1e-3 * out["ms_per_iteration"]) / 1e9 / HBM_PEAK_GBS, 4)
                    roofline["frac_timed_measured_note"] = "PMC traffic of one iteration's launches (single-stream pass, %s) / the timed ms_per_iteration / 8 TB/s" % PMC_FILE
        out.update(extra)
        print(json.dumps(out))
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
