"""Host-side structure of the 8-wide collapsed BVH (csrc/wide_bvh.{h,cpp}) — the claims its parity argument rests on:
  1. it is the reference's tree: every triangle is in exactly one leaf record, with the reference leaf's exact fp32 box and
     the triangle's edge form;
  2. every child's dequantised box (the fp32 value p + q * 2^e the kernel's test bounds) contains the exact box of everything
     below that child;
  3. for each of the six direction trees the depth-first order of the leaves — children ordered by their `before` masks —
     is the order in which the reference's threaded walk (hit links, BVH.cpp:180-239) reaches the leaves;
  4. the stack bound the builder reports covers every root-to-leaf path.
The GPU tier (test_gpu_parity.py) then checks the walk itself against the oracle bit for bit."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SC = os.path.join(ROOT, "scenes")
LEAF = 0x80000000


def _children(node):
    """-> list of (qlo[3], qhi[3], before[6], ref) of the n children of a 36-word (8-wide) or 68-word (16-wide) node.  The 16-wide node stores the
    child's POSITION in each direction's visiting order (four bits each); `before` is rebuilt from the positions of its siblings."""
    n = int(node[3] >> 24)
    out, pos = [], []
    for c in range(n):
        w0, w1, w2, ref = (int(x) for x in node[4 + 4 * c: 8 + 4 * c])
        qlo = (w0 & 255, (w0 >> 8) & 255, (w0 >> 16) & 255)
        qhi = (w0 >> 24, w1 & 255, (w1 >> 8) & 255)
        if len(node) == 36:
            before = ((w1 >> 16) & 255, w1 >> 24, w2 & 255, (w2 >> 8) & 255, (w2 >> 16) & 255, w2 >> 24)
        else:
            bits = (w1 >> 16) | ((w2 & 255) << 16)
            pos.append(tuple((bits >> (4 * d)) & 15 for d in range(6)))
            before = None
        out.append((qlo, qhi, before, ref))
    if pos:
        for d in range(6): assert sorted(p[d] for p in pos) == list(range(n)), "positions are a permutation"
        out = [(qlo, qhi, tuple(sum(1 << k for k in range(n) if pos[k][d] < pos[c][d]) for d in range(6)), ref) for c, (qlo, qhi, _, ref) in enumerate(out)]
    return out


@pytest.mark.parametrize("width", [8, 16])
@pytest.mark.parametrize("name", ["bunny_cornell.txt", "obj_test.txt"])
def test_wide_bvh_structure(ptk, name, width):
    s = ptk.Scene(os.path.join(SC, name))
    nodes, leaves, depth, stack = s.wide_bvh(width)
    tris = s.triangles(); bvh = s.bvh(); B = s.bvh_size
    assert len(nodes) > 0 and len(leaves) == s.n_triangles and depth >= 1
    # 1. leaves: a permutation of the triangles, exact boxes and edge forms
    assert sorted(leaves["tri"].tolist()) == list(range(s.n_triangles))
    leaf_node = {int(n["start"]): n for n in bvh[:B] if n["end"] - n["start"] == 1}
    for L in leaves:
        t = tris[L["tri"]]; v = t["v"].reshape(3, 3); n = leaf_node[int(L["tri"])]
        assert np.array_equal(L["lo"], n["pMin"]) and np.array_equal(L["hi"], n["pMax"])
        assert np.array_equal(L["v0"], v[0]) and np.array_equal(L["e1"], v[1] - v[0]) and np.array_equal(L["e2"], v[2] - v[0])
    # exact box of everything below a reference (bottom-up over the breadth-first node order)
    lo = np.zeros((len(nodes), 3), np.float32); hi = np.zeros((len(nodes), 3), np.float32)
    need = np.zeros(len(nodes), np.int64)
    for i in range(len(nodes) - 1, -1, -1):
        ch = _children(nodes[i])
        assert 1 <= len(ch) <= width
        p = nodes[i][:3].view(np.float32)
        scale = [np.array([((int(nodes[i][3]) >> (8 * a)) & 255) << 23], np.uint32).view(np.float32)[0] for a in range(3)]
        blo, bhi, below = [], [], 0
        for qlo, qhi, before, ref in ch:
            if ref & LEAF:
                clo, chi = leaves[ref & ~LEAF]["lo"], leaves[ref & ~LEAF]["hi"]
            else:
                assert i < ref < len(nodes)                       # breadth-first: children come later
                clo, chi = lo[ref], hi[ref]; below = max(below, int(need[ref]))
            for a in range(3):                                    # 2. fp32 dequantisation as the kernel's bound assumes
                dl = np.float32(p[a] + np.float32(np.float32(qlo[a]) * scale[a]))
                dh = np.float32(p[a] + np.float32(np.float32(qhi[a]) * scale[a]))
                assert dl <= clo[a] and dh >= chi[a], (i, a)
            blo.append(clo); bhi.append(chi)
        lo[i] = np.min(blo, axis=0); hi[i] = np.max(bhi, axis=0)
        need[i] = len(ch) - 1 + below
        # `before` masks describe a total order per direction
        for d in range(6):
            bm = sorted(c[2][d] for c in ch)
            assert [bin(x).count("1") for x in bm] == list(range(len(ch))) and all(bm[k] & ~bm[k + 1] == 0 for k in range(len(ch) - 1))
    assert np.array_equal(lo[0], bvh[0]["pMin"]) and np.array_equal(hi[0], bvh[0]["pMax"])
    assert need[0] == stack                                       # 4.
    # 3. leaf order per direction == the threaded walk's (always taking the hit link = visiting everything)
    for d in range(6):
        t = bvh[d * B:(d + 1) * B]
        ref_order, i = [], 0
        while i != -1:
            if t[i]["end"] - t[i]["start"] == 1: ref_order.append(int(t[i]["start"]))
            i = int(t[i]["hit"])
        order, stk = [], [0]
        while stk:
            r = stk.pop()
            if r & LEAF: order.append(int(leaves[r & ~LEAF]["tri"])); continue
            ch = sorted(_children(nodes[r]), key=lambda c: bin(c[2][d]).count("1"))
            stk.extend(c[3] for c in reversed(ch))
        assert order == ref_order, d
    s.close()
