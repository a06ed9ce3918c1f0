"""Pins of the CPU oracle.  The reference has no tests, golden vectors or fixtures and cannot be built in
this image without stand-in CUDA/thrust headers, so parity is UNPINNED by reference-owned data.  What exists are the
[probe] values recorded in SURVEY.md (numbers the surveyor obtained from the reference's own host-callable code, g++
-O2, glibc sinf/cosf); this file checks the oracle (trig mode 0 = libm) against every one of them.
tests/golden/survey_probes.json holds those values with the SURVEY section each one comes from."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBES = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_probes.json")))


@pytest.fixture(autouse=True)
def _libm_trig(orc):
    """the SURVEY probes come from a g++ host build of the reference: glibc sinf/cosf and g++'s sample2D argument order
    (second argument drawn first).  The oracle's default order is the reference's nvcc order (tests/test_gallery.py)."""
    orc.L.orc_set_sample2d_order.argtypes = [orc.ci]
    orc.set_trig_mode(orc.TRIG_LIBM)
    orc.L.orc_set_sample2d_order(0)
    yield
    orc.L.orc_set_sample2d_order(1)
    orc.set_trig_mode(orc.TRIG_PTK)


def test_utilhash_and_seed(orc):
    for a, h in PROBES["utilhash"]:
        assert orc.L.orc_utilhash(a) == h
    for rec in PROBES["rng_keys"]:
        it, idx, d = rec["key"]
        assert orc.L.orc_seed_hash(it, idx, d) == int(rec["h"], 16)
        u = np.zeros(3, np.float32); orc.L.orc_rng_uniforms(it, idx, d, 3, u.ctypes.data)
        assert [float(x) for x in u] == [float(np.float32(x)) for x in rec["u"]]


def test_thrust_minstd_uniform(orc):
    for rec in PROBES["rng_seeds"]:
        u = np.zeros(len(rec["u"]), np.float32)
        orc.L.orc_rng_uniforms_seed(rec["seed"] & 0xFFFFFFFF, len(u), u.ctypes.data)
        assert [float(x) for x in u] == [float(np.float32(x)) for x in rec["u"]]


def test_unit_box_probe(orc):
    g = np.zeros(1, orc.GEOM_DT)
    tr, rot, sc, o, d = orc.f32(0, 0, 0), orc.f32(0, 0, 0), orc.f32(1, 1, 1), orc.f32(0, 0, 3), orc.f32(0, 0, -1)   # keep alive
    orc.L.orc_make_geom(1, 0, tr.ctypes.data, rot.ctypes.data, sc.ctypes.data, g.ctypes.data)
    P = np.zeros(3, np.float32); N = np.zeros(3, np.float32)
    t = orc.L.orc_box_test(g.ctypes.data, o.ctypes.data, d.ctypes.data, P.ctypes.data, N.ctypes.data)
    b = PROBES["unit_box"]
    assert np.float32(t) == np.float32(b["t"]) and P[2] == np.float32(b["pz"]) and list(N) == b["n"]


def test_scatter_sample_records(orc):
    rec = PROBES["scatter"]
    n = orc.f32(*rec["n"])
    v = np.array(rec["wo_unnormalized"], np.float32)
    d = np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    wo = (v * (np.float32(1) / np.sqrt(d))).astype(np.float32)          # glm::normalize
    for name, exp in rec["materials"].items():
        m = np.zeros(1, orc.MATERIAL_DT)
        m["type"] = exp["type"]; m["albedo"] = rec["albedo"]; m["albedo_s"] = rec["albedo"]
        m["rough_s"] = rec["roughness"]; m["metal_s"] = rec["metallic"]; m["ior"] = rec["ior"]
        out = np.zeros(7, np.float32)
        orc.L.orc_scatter_sample(m.ctypes.data, n.ctypes.data, wo.ctypes.data, *rec["key"], out.ctypes.data)
        got = [float("%.7g" % x) for x in out]
        want = exp["dir"] + exp["bsdf"] + [exp["pdf"]]
        assert np.allclose(got, want, rtol=2e-7, atol=0), (name, got, want)


def test_cube_obj_bvh_nodes(orc):
    o = orc.Scene(os.path.join(ROOT, "scenes", "obj_test.txt"))
    assert (o.n_geoms, o.n_materials, o.n_triangles, o.bvh_size, o.n_lights) == (7, 5, 12, 23, 0)
    b = o.bvh()
    for i, (st, en, hit, miss) in enumerate(PROBES["cube_obj_bvh_dir0"]):
        assert (b[i]["start"], b[i]["end"], b[i]["hit"], b[i]["miss"]) == (st, en, hit, miss)
    assert list(b[0]["pMin"]) == [-2, 4, -1] and list(b[0]["pMax"]) == [-1, 5, 0]


@pytest.mark.parametrize("case", PROBES["renders"], ids=lambda c: c["id"])
def test_render_probes(orc, case):
    """Per-bounce live-path counts (exact) and mean radiance (printed digits) of the SURVEY §6 / §8c-G7/G8 probes."""
    o = orc.Scene(os.path.join(ROOT, "scenes", case["scene"]), res=tuple(case["res"]) if case["res"] else None, depth=case["depth"])
    img = np.zeros((o.P, 3), np.float32)
    tot = np.zeros(o.trace_depth + 1, np.int64); rays = 0
    orc.traversal_stats(reset=True)
    for it in range(1, case["iters"] + 1):
        r, live = o.pathtrace(case["mode"], it, img)
        rays += r; tot += live
    nodes, tris = orc.traversal_stats()
    assert rays // case["iters"] == case["rays_per_iter"]                 # SURVEY quotes the truncated per-iteration mean
    if "live_mean" in case:
        k = len(case["live_mean"])
        assert [int(x) // case["iters"] for x in tot[:k]] == case["live_mean"] and not tot[k:].any()
    if "mean" in case:
        mean = img.astype(np.float64).mean(0) / case["iters"]
        assert np.allclose(mean, case["mean"], rtol=0, atol=6e-7), (mean, case["mean"])
    if "nodes_per_ray" in case:
        assert abs(nodes / rays - case["nodes_per_ray"]) < 5e-3 and abs(tris / rays - case["tris_per_ray"]) < 5e-4


def test_fixed_cube_lights_are_consistent(orc):
    """Opt-in extension (oracle side): with the cube-light fix, MIS and BSDF estimate the same image once the path-length cut
    no longer matters (depth 40): global means within 3 % at 32 spp, 96x96; without the fix MIS is several times darker."""
    orc.set_trig_mode(orc.TRIG_PTK)
    means = {}
    try:
        for fix in (False, True):
            orc.set_light_fix(fix)
            for mode in (0, 2):
                S = orc.Scene(os.path.join(ROOT, "scenes", "cornell.txt"), res=(96, 96), depth=40)
                img = np.zeros((S.P, 3), np.float32)
                for it in range(1, 33):
                    S.pathtrace(mode, it, img)
                means[(fix, mode)] = float(img.mean()) / 32
    finally:
        orc.set_light_fix(False); orc.set_trig_mode(orc.TRIG_LIBM)
    assert abs(means[(True, 2)] / means[(True, 0)] - 1) < 0.03, means
    assert means[(False, 2)] < 0.5 * means[(False, 0)], means
    assert means[(False, 0)] == means[(True, 0)]          # BSDF mode never samples lights


def test_threaded_oracle_is_identical(orc):
    """The optional OpenMP threading of the oracle's per-path loops (bench.py's all-cores CPU baseline) changes nothing: image,
    live counts and traversal counters == the single-thread run."""
    res = {}
    for th in (1, 4):
        orc.set_threads(th)
        try:
            S = orc.Scene(os.path.join(ROOT, "scenes", "bunny_cornell.txt"), res=(96, 96), depth=5)
            img = np.zeros((S.P, 3), np.float32)
            orc.traversal_stats(reset=True)
            lives = [S.pathtrace(2, it, img)[1].tolist() for it in (1, 2)]
            res[th] = (img, lives, orc.traversal_stats(reset=True))
        finally:
            orc.set_threads(1)
    assert np.array_equal(res[1][0], res[4][0]) and res[1][1] == res[4][1] and tuple(res[1][2]) == tuple(res[4][2])
