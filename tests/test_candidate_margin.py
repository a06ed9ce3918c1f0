"""The conservative candidate test of the HIP path (ptk_device.hpp: analytic_candidates) may only skip a primitive whose exact
test (oracle: boxIntersectionTest / sphereIntersectionTest) reports no accepted hit.  This test pins the margin argument on
the CPU: a float32 emulation of the device formula, with the margins read from the header, is run on adversarial rays
(origins on / within 1e-8..1e-2 of faces, edges and corners, grazing directions, zero direction components, far origins)
against the oracle's exact tests, and must never cull an exact hit — not even with the margins divided by 8."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "project3-cuda-path-tracer-1_amd", "csrc", "ptk_device.hpp")
f32 = np.float32


def _constants():
    txt = open(HDR).read()
    a = float(re.search(r"#define\s+PTK_CAND_ABS\s+([0-9.eE+-]+)f", txt).group(1))
    r = float(re.search(r"#define\s+PTK_CAND_REL\s+([0-9.eE+-]+)f", txt).group(1))
    q = float(re.search(r"#define\s+PTK_CAND_QUAD\s+([0-9.eE+-]+)f", txt).group(1))
    q2 = float(re.search(r"#define\s+PTK_CAND_QUAD2\s+([0-9.eE+-]+)f", txt).group(1))
    return a, r, q, q2


def _make_geom(orc, type_, tr, rot, sc):
    g = np.zeros(1, orc.GEOM_DT)
    tr, rot, sc = orc.f32(*tr), orc.f32(*rot), orc.f32(*sc)
    orc.L.orc_make_geom(type_, 0, tr.ctypes.data, rot.ctypes.data, sc.ctypes.data, g.ctypes.data)
    return g


def _world_box(g, pad_abs, pad_rel, quad=0.0):
    """dev_scene.hip: corners of the unit cube through `transform` in double, grown by abs + rel * max|coordinate|; for spheres
    also the coefficients of the per-ray pad qa + qb * |o|^2."""
    M = g["transform"][0].reshape(4, 4).T.astype(np.float64)
    if quad:
        sc = np.linalg.norm(M[:3, :3], axis=0)
        Q = quad * sc.max() / sc.min() ** 2
        return f32(2 * Q * (M[:3, 3] @ M[:3, 3])), f32(2 * Q)
    c = np.array([[x, y, z, 1] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)])
    w = (c @ M.T)[:, :3]
    pad = pad_abs + pad_rel * np.abs(w).max()
    lo = np.nextafter((w.min(0) - pad).astype(f32), f32(-np.inf))
    hi = np.nextafter((w.max(0) + pad).astype(f32), f32(np.inf))
    return lo, hi, M


def _candidate(lo, hi, o, d, rel, qa=0.0, qb=0.0):
    """float32 emulation of analytic_candidates for one primitive."""
    pr = (f32(rel) * (f32(1) + np.abs(o).max(1))).astype(f32)[:, None]
    if qb:
        oo = ((o[:, 0] * o[:, 0]).astype(f32) + (o[:, 1] * o[:, 1]).astype(f32) + (o[:, 2] * o[:, 2]).astype(f32)).astype(f32)
        pg = (f32(qa) + (f32(qb) * oo).astype(f32)).astype(f32)[:, None]
        lo, hi = (lo - pg).astype(f32), (hi + pg).astype(f32)
    with np.errstate(all="ignore"):
        inv = (f32(1) / d).astype(f32)
        t0 = ((lo - (o + pr).astype(f32)).astype(f32) * inv).astype(f32)
        t1 = ((hi - (o - pr).astype(f32)).astype(f32) * inv).astype(f32)
        tn = np.fmax(np.fmax(np.fmin(t0[:, 0], t1[:, 0]), np.fmin(t0[:, 1], t1[:, 1])), np.fmin(t0[:, 2], t1[:, 2]))
        tf = np.fmin(np.fmin(np.fmax(t0[:, 0], t1[:, 0]), np.fmax(t0[:, 1], t1[:, 1])), np.fmax(t0[:, 2], t1[:, 2]))
        return ~(tn > tf) & ~(tf <= 0)


def _round_sphere(g, pad_abs, pad_rel):
    """dev_scene.hip: a sphere under a similarity transform is tested as a world-space sphere: centre, Rp^2, 2 Rp (or None)."""
    M = g["transform"][0].reshape(4, 4).T.astype(np.float64)
    sc = np.linalg.norm(M[:3, :3], axis=0)
    G = M[:3, :3].T @ M[:3, :3]
    off = np.abs(G - np.diag(np.diag(G))).max()
    if sc.max() - sc.min() > 1e-6 * sc.max() or off > 1e-6 * sc.max() ** 2:
        return None
    c = np.array([[x, y, z, 1] for x in (-.5, .5) for y in (-.5, .5) for z in (-.5, .5)])
    pad = pad_abs + pad_rel * np.abs((c @ M.T)[:, :3]).max()
    Rp = 0.5 * sc.max() * (1 + 2e-6) + pad
    return M[:3, 3].astype(f32), np.nextafter(f32(Rp * Rp), f32(np.inf)), np.nextafter(f32(2 * Rp), f32(np.inf))


def _candidate_round(c, Rp2, twoRp, o, d, rel, quad):
    """float32 emulation of the world-space sphere branch of analytic_candidates."""
    pr = (f32(rel) * (f32(1) + np.abs(o).max(1))).astype(f32)
    with np.errstate(all="ignore"):
        oc = (c[None, :] - o).astype(f32)
        oc2 = ((oc[:, 0] * oc[:, 0]).astype(f32) + (oc[:, 1] * oc[:, 1]).astype(f32) + (oc[:, 2] * oc[:, 2]).astype(f32)).astype(f32)
        tca = ((oc[:, 0] * d[:, 0]).astype(f32) + (oc[:, 1] * d[:, 1]).astype(f32) + (oc[:, 2] * d[:, 2]).astype(f32)).astype(f32)
        dd = ((d[:, 0] * d[:, 0]).astype(f32) + (d[:, 1] * d[:, 1]).astype(f32) + (d[:, 2] * d[:, 2]).astype(f32)).astype(f32)
        idd = (f32(1) / dd).astype(f32)
        lim = ((Rp2 + (pr * (twoRp + pr).astype(f32)).astype(f32)).astype(f32) + (f32(quad) * oc2).astype(f32)).astype(f32)
        b2 = (oc2 - ((tca * tca).astype(f32) * idd).astype(f32)).astype(f32)
        cull = (b2 > lim) | ((tca < 0) & (oc2 > lim))
    return ~cull


def _adversarial_rays(rng, M, n):
    p = rng.uniform(-0.5, 0.5, (n, 3))
    ax = rng.integers(0, 3, n)
    p[np.arange(n), ax] = rng.choice([-0.5, 0.5], n)                 # on a face ...
    e = rng.random(n) < 0.5
    ax2 = (ax + 1 + rng.integers(0, 2, n)) % 3
    p[e, ax2[e]] = rng.choice([-0.5, 0.5], int(e.sum()))             # ... half of them on an edge
    c = rng.random(n) < 0.1
    p[c] = rng.choice([-0.5, 0.5], (int(c.sum()), 3))                # ... some on a corner
    p = p + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, -2, (n, 1))
    pw = (np.c_[p, np.ones(n)] @ M.T)[:, :3]
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    far = rng.random(n) < 0.5
    dist = 10.0 ** rng.uniform(-1, 4, (n, 1))
    o = np.where(far[:, None], pw - d * dist, pw)
    d = d + rng.normal(size=(n, 3)) * 10.0 ** rng.uniform(-8, -3, (n, 1))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    z = rng.random(n) < 0.1
    d[z, rng.integers(0, 3, int(z.sum()))] = 0.0                      # axis-parallel rays
    return np.ascontiguousarray(o, f32), np.ascontiguousarray(d, f32)


GEOMS = [  # type (1 cube, 0 sphere), TRANS, ROTAT, SCALE
    (1, (0, 10, 0), (0, 0, 0), (10, .01, 10)), (1, (0, 5, -5), (0, 90, 0), (.01, 10, 10)), (1, (-5, 5, 0), (0, 0, 0), (.01, 10, 10)),
    (1, (1, 2, 3), (30, 40, 50), (2, 1, .5)), (1, (100, -50, 30), (10, 200, -70), (5, 5, 5)), (1, (0, 0, 0), (45, 45, 45), (1e-2, 1e-2, 1e-2)),
    (0, (-2, 4, -2), (0, 0, 0), (3, 3, 3)), (0, (2, 5, 2), (20, 30, 40), (4, 2, 1)), (1, (0, 5, 0), (0, 45, 0), (1e3, 0.01, 1e-2)),
    (0, (400, 5, 0), (0, 0, 0), (3, 3, 3)), (1, (0, 0, 0), (0, 0, 0), (3, 0.3, 3)), (0, (0, 0, 0), (0, 0, 0), (1e-3, 1e-3, 1e-3)),
    (0, (3, 3, 3), (10, 20, 30), (0.5, 2, 1)), (0, (0, 0, 0), (0, 0, 0), (100, 100, 100)), (0, (2, 5, 2), (0, 0, 0), (4, 4, 4)),
    (0, (-30, 7, 11), (25, 130, -40), (0.7, 0.7, 0.7)),
]


@pytest.mark.parametrize("divisor", [1.0, 8.0])
def test_candidate_test_never_culls_an_exact_hit(orc, divisor):
    pad_abs, pad_rel, pad_quad, pad_quad2 = _constants()
    assert pad_rel >= 1e-6 and pad_abs >= 1e-6 and pad_quad >= 2e-6 and pad_quad2 >= 1e-5, "margins below the validated range"
    rng = np.random.default_rng(5)
    N = 120000
    total_hits = 0
    for ty, tr, rot, sc in GEOMS:
        g = _make_geom(orc, ty, tr, rot, sc)
        lo, hi, M = _world_box(g, pad_abs / divisor, pad_rel / divisor)
        o, d = _adversarial_rays(rng, M, N)
        t = np.zeros(N, f32)
        fn = orc.L.orc_box_test_array if ty == 1 else orc.L.orc_sphere_test_array
        fn(g.ctypes.data, o.ctypes.data, d.ctypes.data, N, t.ctypes.data)
        hit = t > 0
        qa, qb = _world_box(g, 0, 0, pad_quad / divisor) if ty == 0 else (0.0, 0.0)
        rs = _round_sphere(g, pad_abs / divisor, pad_rel / divisor) if ty == 0 else None
        if rs is not None:
            cand = _candidate_round(rs[0], rs[1], rs[2], o, d, pad_rel / divisor, pad_quad2 / divisor)
        else:
            cand = _candidate(lo, hi, o, d, pad_rel / divisor, qa, qb)
        bad = hit & ~cand
        assert not bad.any(), f"geom {(ty, tr, rot, sc)}: {int(bad.sum())} exact hits culled, e.g. o={o[bad][0]} d={d[bad][0]} t={t[bad][0]}"
        total_hits += int(hit.sum())
    assert total_hits > 100000          # the ray generator does produce hits to check
