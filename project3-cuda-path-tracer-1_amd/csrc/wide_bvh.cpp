// wide_bvh.cpp — collapses the reference's binary SAH tree (as threaded by BVH.cpp:180-239) into 8-wide quantised nodes.
// See wide_bvh.h for why the result is bit-identical to the reference's walk.
#include "wide_bvh.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>

namespace ptk {

namespace {

struct Topo {
    const std::vector<GpuBVHNode>& mt;
    int B;
    bool leaf(int i) const { return mt[i].end - mt[i].start <= 1; }
    int left(int i) const { return i + 1; }
    int right(int i) const { return i + 2 * (mt[i + 1].end - mt[i + 1].start); }     // a subtree over n leaves has 2n - 1 nodes
    int near_child(int d, int i) const { return mt[(size_t)d * B + i].hit; }
    double area(int i) const {
        const GpuBVHNode& n = mt[i];
        const double x = (double)n.pMax.x - n.pMin.x, y = (double)n.pMax.y - n.pMin.y, z = (double)n.pMax.z - n.pMin.z;
        return x * y + y * z + z * x;
    }
};

bool check_topology(const Topo& T) {
    if (T.B <= 0 || (int)T.mt.size() != 6 * T.B) return false;
    for (int i = 0; i < T.B; ++i) {
        const GpuBVHNode& n = T.mt[i];
        if (n.end - n.start < 1) return false;                       // empty leaves only exist in triangle-free scenes
        if (T.leaf(i)) continue;
        const int L = T.left(i), R = T.right(i);
        if (L >= T.B || R >= T.B || R <= L) return false;
        if (T.mt[L].start != n.start || T.mt[L].end != T.mt[R].start || T.mt[R].end != n.end) return false;
        for (int d = 0; d < 6; ++d) { const int h = T.near_child(d, i); if (h != L && h != R) return false; }
        for (int d = 1; d < 6; ++d) {                                // same boxes in every copy
            const GpuBVHNode& c = T.mt[(size_t)d * T.B + i];
            if (std::memcmp(&c.pMin, &n.pMin, sizeof(n.pMin)) || std::memcmp(&c.pMax, &n.pMax, sizeof(n.pMax))) return false;
        }
    }
    return true;
}

// depth-first order of the frontier slots below binary node `i`, near child first (direction d)
void order_slots(const Topo& T, int d, int i, const std::vector<int>& slot_of, std::vector<int>& out) {
    if (slot_of[i] >= 0) { out.push_back(slot_of[i]); return; }
    const int L = T.left(i), R = T.right(i), nr = T.near_child(d, i);
    order_slots(T, d, nr, slot_of, out);
    order_slots(T, d, nr == L ? R : L, slot_of, out);
}

float comp(const vec3& v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// Which frontier of the binary tree becomes the children of a wide node.  Any frontier is parity-preserving (wide_bvh.h); this picks the one
// that minimises the expected number of node visits of a random ray — the surface-area measure the reference's own builder uses
// (BVH.cpp:36-75) — by dynamic programming over the binary tree, bottom-up:
//   node(i)    = area(i) + slots(i, 8)                      a wide node rooted at inner node i: it is visited with probability ~ area(i)
//   slots(i,1) = leaf(i) ? 0 : node(i)                      subtree i behind ONE child slot (a leaf slot costs the same in every collapse)
//   slots(i,k) = min(slots(i,k-1), min_j slots(L,j) + slots(R,k-j))
// (the collapse of Ylitie, Karras, Laine, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", HPG 2017, section 4.1, with
// one triangle per leaf as in the reference).  Round 5: the greedy "open the largest child" expansion it replaces left the bunny's nodes 4.1 / 8
// full (3 202 nodes, 6 levels): every small subtree that reached a frontier became a node of its own.
// PTK_WIDE_COLLAPSE=greedy: the round-2 expansion (largest-area child first), kept for A/B measurements; anything else: the plan below
int collapse_mode() { const char* e = std::getenv("PTK_WIDE_COLLAPSE"); return (e && std::strcmp(e, "greedy") == 0) ? 0 : 1; }

struct CollapsePlan {
    bool on = false;
    int W = 8, K = 9;                // children per wide node (8 or 16); row length W + 1
    std::vector<double> open;        // [B][K]: cost of subtree i OPENED into <= k slots (k = 2..W; its two children share them), non-increasing in k
    std::vector<double> node;        // [B]: cost of subtree i as a wide node of its own (behind one slot); 0 for a leaf
    std::vector<uint8_t> split;      // [B][K]: slots given to the left child when i is opened into exactly k
    void solve(const Topo& T, double node_bias, int width) {
        W = width; K = W + 1;
        open.assign((size_t)T.B * K, 0.0); node.assign(T.B, 0.0); split.assign((size_t)T.B * K, 0);
        auto slots = [&](int i, int k) { return T.leaf(i) ? 0.0 : (k == 1 ? node[i] : std::min(node[i], open[(size_t)i * K + k])); };
        for (int i = T.B - 1; i >= 0; --i) {                 // preorder array: children come later
            if (T.leaf(i)) continue;
            const int L = T.left(i), R = T.right(i);
            double* o = &open[(size_t)i * K]; uint8_t* sp = &split[(size_t)i * K];
            for (int k = 2; k <= W; ++k) {
                double best = 1e300; int bj = 1;
                for (int j = 1; j < k; ++j) { const double c = slots(L, j) + slots(R, k - j); if (c < best) { best = c; bj = j; } }
                if (k > 2 && o[k - 1] <= best) { o[k] = o[k - 1]; sp[k] = 0; }       // 0: no better than with one slot less — use that plan
                else { o[k] = best; sp[k] = (uint8_t)bj; }
            }
            node[i] = T.area(i) + node_bias + o[W];
        }
        on = true;
    }
    void expand(const Topo& T, int i, int k, bool forced, std::vector<int>& fr) const {
        if (T.leaf(i) || k == 1 || (!forced && node[i] <= open[(size_t)i * K + k])) { fr.push_back(i); return; }
        while (split[(size_t)i * K + k] == 0) --k;
        const int j = split[(size_t)i * K + k];
        expand(T, T.left(i), j, false, fr); expand(T, T.right(i), k - j, false, fr);
    }
    void cut(const Topo& T, int r, std::vector<int>& fr) const { expand(T, r, W, true, fr); }
};

} // namespace

bool build_wide_bvh(const std::vector<GpuBVHNode>& mt, int B, const std::vector<Triangle>& tris, WideBvh& out, int width) {
    out = WideBvh();
    if (width != 8 && width != 16) return false;
    const int W = width;
    out.width = W;
    Topo T{mt, B};
    if (tris.empty() || !check_topology(T)) return false;
    struct Pending { int root; int level; };
    std::deque<Pending> queue;
    queue.push_back({0, 0});
    CollapsePlan plan;
    if (collapse_mode() != 0) { const char* e = std::getenv("PTK_WIDE_NODE_BIAS"); plan.solve(T, (e ? std::atof(e) : 0.0) * T.area(0), W); }
    std::vector<int> slot_of(B, -1);
    std::vector<int> node_root;                      // binary root of every wide node (for the stack bound)
    std::vector<std::vector<int>> kids;              // inner children (wide indices) per wide node
    std::vector<int> nkids;
    size_t next_index = 1;                           // wide index the next enqueued node will get
    while (!queue.empty()) {
        const Pending cur = queue.front(); queue.pop_front();
        const int r = cur.root;
        out.depth = std::max(out.depth, cur.level + 1);
        std::vector<int> fr;
        if (T.leaf(r)) fr.push_back(r);
        else if (plan.on) plan.cut(T, r, fr);
        else {
            fr.push_back(T.left(r)); fr.push_back(T.right(r));
            while ((int)fr.size() < W) {
                int best = -1; double ba = -1.0;
                for (size_t k = 0; k < fr.size(); ++k) if (!T.leaf(fr[k]) && T.area(fr[k]) > ba) { ba = T.area(fr[k]); best = (int)k; }
                if (best < 0) break;
                const int f = fr[best];
                fr[best] = T.left(f);
                fr.insert(fr.begin() + best + 1, T.right(f));
            }
        }
        const int n = (int)fr.size();
        uint32_t ndw[68]; std::memset(ndw, 0, sizeof(ndw));      // 9 (W = 8) or 17 (W = 16) uint4
        const GpuBVHNode& box = mt[r];
        uint32_t imask = 0;
        for (int c = 0; c < n; ++c) if (!T.leaf(fr[c])) imask |= 1u << c;
        // quantisation frame: origin = box.pMin, per-axis scale 2^e with p + 255 * 2^e >= box.pMax
        uint32_t eb[3];
        uint8_t qlo[3][16], qhi[3][16];
        for (int a = 0; a < 3; ++a) {
            const long double p = comp(box.pMin, a), hi = comp(box.pMax, a);
            if (!std::isfinite((double)p) || !std::isfinite((double)hi)) return false;
            int e = -126;
            const long double ext = hi - p;
            if (ext > 0) e = std::max(-126, (int)std::ceil(std::log2((double)(ext / 255.0L))) - 1);
            while (p + 255.0L * std::ldexp(1.0L, e) < hi) ++e;
            if (e > 20) return false;                                 // the kernel admits |1 / d| up to 1e30: q * (2^e / d) with q <= 255 stays finite (no +inf near
                                                                      // plane that would cull a child wrongly) only for e <= 20; larger extents (> 2.7e8 units) -> threaded walk
            eb[a] = (uint32_t)(e + 127);
            const long double s = std::ldexp(1.0L, e);
            for (int c = 0; c < W; ++c) { qlo[a][c] = 255; qhi[a][c] = 0; }
            for (int c = 0; c < n; ++c) {
                const long double lo_c = comp(mt[fr[c]].pMin, a), hi_c = comp(mt[fr[c]].pMax, a);
                long double ql = std::floor((lo_c - p) / s), qh = std::ceil((hi_c - p) / s);
                ql = std::min(255.0L, std::max(0.0L, ql)); qh = std::min(255.0L, std::max(0.0L, qh));
                while (ql > 0 && p + ql * s > lo_c) ql -= 1;
                while (qh < 255 && p + qh * s < hi_c) qh += 1;
                if (p + ql * s > lo_c || p + qh * s < hi_c) return false;
                qlo[a][c] = (uint8_t)ql; qhi[a][c] = (uint8_t)qh;
            }
        }
        float pf[3] = {box.pMin.x, box.pMin.y, box.pMin.z};
        std::memcpy(&ndw[0], pf, 12);
        ndw[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | ((uint32_t)n << 24);
        // children: inner ones become consecutive wide nodes (breadth-first), leaves consecutive leaf records (slot order)
        uint32_t ref[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<int> my_kids;
        for (int c = 0; c < n; ++c) {
            if (imask & (1u << c)) { queue.push_back({fr[c], cur.level + 1}); my_kids.push_back((int)next_index); ref[c] = (uint32_t)next_index; ++next_index; }
            else {
                const GpuBVHNode& lf = mt[fr[c]];
                if (lf.start < 0 || lf.start >= (int)tris.size()) return false;
                const Triangle& t = tris[lf.start];
                WideLeafHost L;
                L.lo[0] = lf.pMin.x; L.lo[1] = lf.pMin.y; L.lo[2] = lf.pMin.z;
                L.hi[0] = lf.pMax.x; L.hi[1] = lf.pMax.y; L.hi[2] = lf.pMax.z;
                const vec3* v = t.v;                                   // same fp32 subtractions as DevTriE (Bounds3.hpp:300-301)
                L.v0[0] = v[0].x; L.v0[1] = v[0].y; L.v0[2] = v[0].z;
                L.e1[0] = v[1].x - v[0].x; L.e1[1] = v[1].y - v[0].y; L.e1[2] = v[1].z - v[0].z;
                L.e2[0] = v[2].x - v[0].x; L.e2[1] = v[2].y - v[0].y; L.e2[2] = v[2].z - v[0].z;
                L.tri = lf.start;
                ref[c] = 0x80000000u | (uint32_t)out.leaves.size();
                out.leaves.push_back(L);
            }
        }
        // visiting order per direction -> before[d][c]: the children visited before child c by a ray of direction tree d
        // (W = 16: pos[d][c], the child's POSITION in direction d's visiting order, four bits each — a set of sixteen children per direction would not
        // fit the 16-byte child record; the kernel ranks a hit child by the hit children at smaller positions)
        uint32_t before[6][16], pos[6][16];
        for (int c = 0; c < n; ++c) slot_of[fr[c]] = c;
        for (int d = 0; d < 6; ++d) {
            std::vector<int> ord;
            order_slots(T, d, r, slot_of, ord);
            if ((int)ord.size() != n) return false;
            uint32_t seen = 0;
            for (int k = 0; k < n; ++k) { before[d][ord[k]] = seen; pos[d][ord[k]] = (uint32_t)k; seen |= 1u << ord[k]; }
            for (int c = n; c < W; ++c) { before[d][c] = 0; pos[d][c] = 0; }
        }
        for (int c = 0; c < n; ++c) slot_of[fr[c]] = -1;
        for (int c = 0; c < W; ++c) {
            uint32_t* w = &ndw[4 + 4 * c];
            w[0] = (uint32_t)qlo[0][c] | ((uint32_t)qlo[1][c] << 8) | ((uint32_t)qlo[2][c] << 16) | ((uint32_t)qhi[0][c] << 24);
            if (W == 8) {
                w[1] = (uint32_t)qhi[1][c] | ((uint32_t)qhi[2][c] << 8) | (before[0][c] << 16) | (before[1][c] << 24);
                w[2] = before[2][c] | (before[3][c] << 8) | (before[4][c] << 16) | (before[5][c] << 24);
            } else {
                w[1] = (uint32_t)qhi[1][c] | ((uint32_t)qhi[2][c] << 8) | (pos[0][c] << 16) | (pos[1][c] << 20) | (pos[2][c] << 24) | (pos[3][c] << 28);
                w[2] = pos[4][c] | (pos[5][c] << 4);
            }
            w[3] = ref[c];
        }
        if (W == 8) { WideNodeHost nd; std::memcpy(nd.w, ndw, sizeof(nd.w)); out.nodes.push_back(nd); }
        else { WideNode16Host nd; std::memcpy(nd.w, ndw, sizeof(nd.w)); out.nodes16.push_back(nd); }
        kids.push_back(my_kids); nkids.push_back(n);
    }
    if (out.leaves.size() != tris.size() || out.nodes.size() + out.nodes16.size() != next_index) return false;
    // per-ray stack bound: a visit pushes at most (children - 1) entries and descends into one
    std::vector<int> need(next_index, 0);
    for (int i = (int)next_index - 1; i >= 0; --i) {
        int below = 0;
        for (int k : kids[i]) below = std::max(below, need[k]);
        need[i] = (nkids[i] - 1) + below;
    }
    out.max_stack = need.empty() ? 0 : need[0];
    out.ok = true;
    return true;
}

} // namespace ptk
