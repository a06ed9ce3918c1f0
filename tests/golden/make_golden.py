"""Generates tests/golden/oracle_golden.npz — the small golden fixtures of SURVEY.md §8c (G1-G9) — from the CPU oracle
(oracle/pt_oracle.cpp) in ptk-trig mode (the mode the HIP kernels are compared in).  Run from the repo root:

    python tests/golden/make_golden.py

The oracle itself is pinned to the SURVEY probe values of the reference by tests/test_oracle_pins.py (libm-trig mode); these
vectors freeze what it computes, so that (a) a change to the oracle is noticed, and (b) the GPU tests can be checked against
committed data as well as against the oracle built on the box.  Inputs that are random are drawn from numpy's PCG64 with the
seeds written below; everything is float32 / int32 and compared bit-exactly (tests/test_golden.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc  # noqa: E402

SC = os.path.join(ROOT, "scenes")
OUT = os.path.join(ROOT, "tests", "golden", "oracle_golden.npz")


def block_average(img, W, H, b):
    a = img.reshape(H, W, 3).astype(np.float64)
    return a.reshape(H // b, b, W // b, b, 3).mean(axis=(1, 3)).astype(np.float32)


def sample_rays(rng, n, box_lo, box_hi):
    o = rng.uniform(box_lo, box_hi, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d.astype(np.float32)


def build():
    """-> dict of arrays (what main() writes)"""
    orc.set_trig_mode(orc.TRIG_PTK)
    g = {}
    # G1 RNG table: (iter, index, depth) -> hash, first 4 uniforms
    keys = np.array([[1, 0, 0], [1, 12345, 3], [5000, 639999, 7], [2, 1, 1], [17, 640000 * 3, 15], [4096, 77, 0], [1, 0x7fffffff, 63], [0, 0, 0]], np.int32)
    g["g1_keys"] = keys
    g["g1_hash"] = np.array([orc.L.orc_seed_hash(int(a), int(b), int(c)) for a, b, c in keys], np.uint32)
    u = np.zeros((len(keys), 4), np.float32)
    for i, (a, b, c) in enumerate(keys):
        orc.L.orc_rng_uniforms(int(a), int(b), int(c), 4, u[i].ctypes.data)
    g["g1_uniforms"] = u
    # G2 camera: fixed-up camera + rays of 16 pixels x iterations {1, 2}, cornell 800x800
    S = orc.Scene(os.path.join(SC, "cornell.txt"))
    g["g2_camera"] = S.camera().reshape(1)
    pix = np.array([0, 1, 799, 800, 319999, 320400, 400 + 400 * 800, 639999, 12345, 54321, 100 + 700 * 800, 700 + 100 * 800, 2, 3, 638400, 639200], np.int64)
    g["g2_pixels"] = pix.astype(np.int32)
    g["g2_dirs"] = np.stack([S.generate_rays(it)["d"][pix] for it in (1, 2)])
    S.close()
    # G3 per-primitive hits: 64 rays x {cube, sphere} with a rotated, non-uniformly scaled transform; triangles and AABBs
    rng = np.random.default_rng(20240301)
    for name, ty in (("cube", 1), ("sphere", 0)):
        geom = np.zeros(1, orc.GEOM_DT)
        tr, rot, sc = orc.f32(0.5, -1.0, 2.0), orc.f32(25.0, -40.0, 70.0), orc.f32(2.0, 0.75, 1.25)
        orc.L.orc_make_geom(ty, 0, tr.ctypes.data, rot.ctypes.data, sc.ctypes.data, geom.ctypes.data)
        o, d = sample_rays(rng, 64, [-4, -4, -2], [4, 4, 6])
        t = np.zeros(64, np.float32); P = np.zeros((64, 3), np.float32); N = np.zeros((64, 3), np.float32)
        fn = orc.L.orc_box_test if ty == 1 else orc.L.orc_sphere_test
        for i in range(64):
            t[i] = fn(geom.ctypes.data, o[i].ctypes.data, d[i].ctypes.data, P[i].ctypes.data, N[i].ctypes.data)
        hit = t > 0
        P[~hit] = 0; N[~hit] = 0                       # outputs are unspecified on a miss
        g[f"g3_{name}_o"], g[f"g3_{name}_d"], g[f"g3_{name}_t"], g[f"g3_{name}_p"], g[f"g3_{name}_n"] = o, d, t, P, N
    o, d = sample_rays(rng, 64, [-2, -2, -2], [2, 2, 2])
    pmin, pmax = orc.f32(-0.5, -0.25, -1.0), orc.f32(0.75, 0.5, 0.25)
    ah = np.zeros(64, np.int32); at = np.zeros(64, np.float32)
    for i in range(64):
        tt = np.zeros(1, np.float32)
        ah[i] = orc.L.orc_aabb_test(pmin.ctypes.data, pmax.ctypes.data, o[i].ctypes.data, d[i].ctypes.data, tt.ctypes.data); at[i] = tt[0] if ah[i] else 0
    g["g3_aabb_o"], g["g3_aabb_d"], g["g3_aabb_hit"], g["g3_aabb_t"] = o, d, ah, at
    # G4 BVH: full node table x6 for cube.obj @ obj_test; counts + first/last 32 nodes per direction for the bunny scene
    S = orc.Scene(os.path.join(SC, "obj_test.txt"))
    g["g4_cube_bvh"] = S.bvh(); g["g4_cube_counts"] = np.array([S.n_triangles, S.bvh_size], np.int32)
    S.close()
    S = orc.Scene(os.path.join(SC, "bunny_cornell.txt"))
    b = S.bvh().reshape(6, S.bvh_size)
    g["g4_bunny_counts"] = np.array([S.n_triangles, S.bvh_size], np.int32)
    g["g4_bunny_first"] = b[:, :32].copy(); g["g4_bunny_last"] = b[:, -32:].copy()
    g["g4_bunny_leaf_count"] = np.array([int(((b[k]["end"] - b[k]["start"]) == 1).sum()) for k in range(6)], np.int32)
    # G9 first-bounce hit records of 1024 sampled pixels (and the per-bounce live counts of iteration 1) per config scene
    for tag, scn, res, depth in (("cornell", "cornell.txt", None, -1), ("bunny", "bunny_cornell.txt", None, -1), ("objtest", "obj_test.txt", None, -1),
                                 ("mis", "mis_test.txt", (1024, 1024), 16)):
        S2 = orc.Scene(os.path.join(SC, scn), res=res, depth=depth)
        p = S2.generate_rays(1)
        sel = np.random.default_rng(7).choice(S2.P, 1024, replace=False); sel.sort()
        h, v = S2.intersect(np.ascontiguousarray(p[sel]))
        g[f"g9_{tag}_pixels"] = sel.astype(np.int32); g[f"g9_{tag}_isects"] = h; g[f"g9_{tag}_valid"] = v
        S2.close()
    S.close()
    # G5 scatterSample / BSDF / pdf: 5 material types x 32 (n, wo, rng key)
    rng = np.random.default_rng(5)
    mats = np.zeros(5, orc.MATERIAL_DT)
    for k in range(5):
        mats[k]["type"] = k; mats[k]["albedo"] = (0.8, 0.6, 0.4); mats[k]["albedo_s"] = (0.8, 0.6, 0.4)
        mats[k]["rough_s"] = 0.3; mats[k]["metal_s"] = 0.5; mats[k]["ior"] = 1.5
        for f in ("albedoMapID", "metallicMapID", "roughnessMapID", "normalMapID"):
            mats[k][f] = -1
    n = rng.normal(size=(32, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    wo = rng.normal(size=(32, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    n = n.astype(np.float32); wo = wo.astype(np.float32)
    g["g5_n"], g["g5_wo"] = n, wo
    out = np.zeros((5, 32, 7), np.float32)
    for k in range(5):
        for i in range(32):
            orc.L.orc_scatter_sample(mats[k:k + 1].ctypes.data, n[i].ctypes.data, wo[i].ctypes.data, 3, 1000 + i, 2, out[k, i].ctypes.data)
    g["g5_scatter"] = np.nan_to_num(out, nan=-12345.0, posinf=3e38, neginf=-3e38)
    # G6 lightSample / lightPDF: sphere lights (mis_test), a cube light (cornell, reference behaviour: nothing assigned) and fixed
    for tag, scn, fix in (("mis", "mis_test.txt", 0), ("cornell", "cornell.txt", 0), ("cornell_fixed", "cornell.txt", 1)):
        orc.set_light_fix(fix)
        S = orc.Scene(os.path.join(SC, scn))
        vp = np.random.default_rng(11).uniform([-3, 1, -3], [3, 8, 3], (32, 3)).astype(np.float32)
        rec = np.zeros((32, 7), np.float32)
        for i in range(32):
            orc.L.orc_light_sample(S.h, vp[i].ctypes.data, 5, 200 + i, 1, 0, rec[i].ctypes.data)
        g[f"g6_{tag}_view"] = vp; g[f"g6_{tag}_rec"] = np.nan_to_num(rec, nan=-12345.0, posinf=3e38, neginf=-3e38)
        S.close()
    orc.set_light_fix(0)
    # G7 per-bounce live counts + G8 radiance (8x8 block averages + global means): cornell 400x400 d4, 16 spp, BSDF and MIS;
    #    live counts of cornell 800x800 d8 iteration 1
    for mode, mname in ((0, "bsdf"), (2, "mis")):
        S = orc.Scene(os.path.join(SC, "cornell.txt"), res=(400, 400), depth=4)
        img = np.zeros((S.P, 3), np.float32)
        lives = np.stack([S.pathtrace(mode, it, img)[1] for it in range(1, 17)])
        g[f"g7_cornell400_{mname}_live"] = lives
        g[f"g8_cornell400_{mname}_blocks"] = block_average(img / np.float32(16), 400, 400, 8)
        g[f"g8_cornell400_{mname}_mean"] = (img.astype(np.float64).mean(0) / 16).astype(np.float32)
        S.close()
    S = orc.Scene(os.path.join(SC, "cornell.txt"))
    img = np.zeros((S.P, 3), np.float32)
    g["g7_cornell800_live"] = S.pathtrace(0, 1, img)[1]
    g["g8_cornell800_blocks16"] = block_average(img, 800, 800, 16)
    S.close()
    S = orc.Scene(os.path.join(SC, "mis_test.txt"), res=(512, 512), depth=16)
    img = np.zeros((S.P, 3), np.float32)
    g["g7_mis512_live"] = np.stack([S.pathtrace(2, it, img)[1] for it in (1, 2)])
    g["g8_mis512_blocks"] = block_average(img / np.float32(2), 512, 512, 8)
    S.close()
    return g


def main():
    g = build()
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, os.path.getsize(OUT), "bytes,", len(g), "arrays")


if __name__ == "__main__":
    main()
