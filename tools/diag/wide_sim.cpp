// wide_sim.cpp — CPU model of k_traverse_wide's walk (tools/diag/coherence_sim.py): per-ray node / leaf visit counts over the 8-wide
// collapsed tree, to evaluate work-list orderings without GPU time.  Diagnostic only (double-precision child tests: the counts are those
// of a conservative walk, not bit-exact with the kernel's margins).
//   g++ -O2 -shared -fPIC -o /tmp/libwidesim.so tools/diag/wide_sim.cpp
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>

static int mtbvh_dir(const float* d) {
    const float ax = std::fabs(d[0]), ay = std::fabs(d[1]), az = std::fabs(d[2]);
    int k = (ax > ay && ax > az) ? 0 : (ay > az ? 1 : 2);
    return k + (d[k] > 0 ? 0 : 3);
}

// optional per-node visit histogram (wide_tree_stats.py: which share of the node visits the first N nodes of the breadth-first array take)
static unsigned long long* g_hist = nullptr;
extern "C" void wide_sim_set_hist(unsigned long long* h) { g_hist = h; }

extern "C" void wide_sim(const uint32_t* nodes, int n_nodes, const float* leaves, int n_rays, const float* o, const float* d,
                         const float* tmin_init, int* node_visits, int* leaf_visits, float* t_out) {
    (void)n_nodes;
    for (int r = 0; r < n_rays; ++r) {
        const float* ro = o + 3 * r; const float* rd = d + 3 * r;
        const double inv[3] = {1.0 / rd[0], 1.0 / rd[1], 1.0 / rd[2]};
        float t_min = tmin_init[r];
        const int dir = mtbvh_dir(rd);
        uint32_t stk[64]; int sp = 0; uint32_t cur = 0; int nv = 0, nl = 0, spmax = 0;
        for (;;) {
            if (!(cur & 0x80000000u)) {
                ++nv;
                if (g_hist) ++g_hist[cur];
                const uint32_t* nd = nodes + 36 * (size_t)cur;
                float p[3]; std::memcpy(p, nd, 12);
                double sc[3];
                for (int a = 0; a < 3; ++a) { uint32_t e = ((nd[3] >> (8 * a)) & 255u) << 23; float f; std::memcpy(&f, &e, 4); sc[a] = f; }
                const int nc = (int)(nd[3] >> 24);
                uint32_t H = 0; uint32_t before[8]; uint32_t ref[8];
                for (int c = 0; c < nc; ++c) {
                    const uint32_t* w = nd + 4 + 4 * c;
                    const int qlo[3] = {(int)(w[0] & 255), (int)((w[0] >> 8) & 255), (int)((w[0] >> 16) & 255)};
                    const int qhi[3] = {(int)(w[0] >> 24), (int)(w[1] & 255), (int)((w[1] >> 8) & 255)};
                    const uint64_t ww = ((uint64_t)w[2] << 32) | w[1];
                    before[c] = (uint32_t)(ww >> (16 + 8 * dir)) & 255u; ref[c] = w[3];
                    double tn = -DBL_MAX, tf = DBL_MAX;
                    for (int a = 0; a < 3; ++a) {
                        const double lo = p[a] + qlo[a] * sc[a], hi = p[a] + qhi[a] * sc[a];
                        double t0 = (lo - ro[a]) * inv[a], t1 = (hi - ro[a]) * inv[a];
                        if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
                        tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
                    }
                    if (tn <= (tf < t_min ? tf : t_min) && tf >= 0) H |= 1u << c;
                }
                const int nh = __builtin_popcount(H);
                for (int c = 0; c < nc; ++c) if (H & (1u << c)) stk[sp + nh - 1 - __builtin_popcount(H & before[c])] = ref[c];
                sp += nh; if (sp > spmax) spmax = sp;
            } else {
                ++nl;
                const float* L = leaves + 16 * (size_t)(cur & 0x7fffffffu);
                double tn = -DBL_MAX, tf = DBL_MAX;
                for (int a = 0; a < 3; ++a) {
                    double t0 = ((double)L[a] - ro[a]) * inv[a], t1 = ((double)L[3 + a] - ro[a]) * inv[a];
                    if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
                    tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
                }
                if (tn <= tf && tf > 0 && !(tn > t_min)) {
                    const float* v0 = L + 6; const float* e1 = L + 9; const float* e2 = L + 12;
                    const double px = rd[1] * e2[2] - rd[2] * e2[1], py = rd[2] * e2[0] - rd[0] * e2[2], pz = rd[0] * e2[1] - rd[1] * e2[0];
                    const double det = e1[0] * px + e1[1] * py + e1[2] * pz;
                    if (det != 0) {
                        const double id = 1.0 / det, sx = ro[0] - v0[0], sy = ro[1] - v0[1], sz = ro[2] - v0[2];
                        const double u = (sx * px + sy * py + sz * pz) * id;
                        const double qx = sy * e1[2] - sz * e1[1], qy = sz * e1[0] - sx * e1[2], qz = sx * e1[1] - sy * e1[0];
                        const double v = (rd[0] * qx + rd[1] * qy + rd[2] * qz) * id, t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * id;
                        if (!(t < 0 || u < 0 || v < 0 || 1 - u - v < 0) && t_min > t) t_min = (float)t;
                    }
                }
            }
            if (sp == 0) break;
            cur = stk[--sp];
        }
        node_visits[r] = nv; leaf_visits[r] = nl; t_out[r] = t_min;
    }
}

// predictors: pred[4 r + 0] = children of the root the ray's slab test hits, [1] = node + leaf visits of a walk that stops descending below wide
// level 1 (root + its children: what two visits per ray can see), [2] = hit grandchildren counted there, [3] = same walk limited to level 2
extern "C" void wide_pred(const uint32_t* nodes, const int* level_of, const float* leaves, int n_rays, const float* o, const float* d, const float* tmin_init, int* pred) {
    (void)leaves;
    for (int r = 0; r < n_rays; ++r) {
        const float* ro = o + 3 * r; const float* rd = d + 3 * r;
        const double inv[3] = {1.0 / rd[0], 1.0 / rd[1], 1.0 / rd[2]};
        const float t_min = tmin_init[r];
        int* P = pred + 4 * r; P[0] = P[1] = P[2] = P[3] = 0;
        uint32_t stk[512]; int sp = 0; stk[sp++] = 0;
        while (sp) {
            const uint32_t cur = stk[--sp];
            const int lv = level_of[cur];
            const uint32_t* nd = nodes + 36 * (size_t)cur;
            float p[3]; std::memcpy(p, nd, 12);
            double sc[3];
            for (int a = 0; a < 3; ++a) { uint32_t e = ((nd[3] >> (8 * a)) & 255u) << 23; float f; std::memcpy(&f, &e, 4); sc[a] = f; }
            const int nc = (int)(nd[3] >> 24);
            for (int c = 0; c < nc; ++c) {
                const uint32_t* w = nd + 4 + 4 * c;
                const int qlo[3] = {(int)(w[0] & 255), (int)((w[0] >> 8) & 255), (int)((w[0] >> 16) & 255)};
                const int qhi[3] = {(int)(w[0] >> 24), (int)(w[1] & 255), (int)((w[1] >> 8) & 255)};
                double tn = -DBL_MAX, tf = DBL_MAX;
                for (int a = 0; a < 3; ++a) {
                    const double lo = p[a] + qlo[a] * sc[a], hi = p[a] + qhi[a] * sc[a];
                    double t0 = (lo - ro[a]) * inv[a], t1 = (hi - ro[a]) * inv[a];
                    if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
                    tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
                }
                if (!(tn <= (tf < t_min ? tf : t_min) && tf >= 0)) continue;
                if (lv == 0) { P[0]++; P[2] |= 1 << (8 + c); }
                if (lv == 1) { P[2]++; }
                if (lv <= 1) P[1]++;
                P[3]++;
                if (!(w[3] & 0x80000000u) && lv < 2 && sp < 500) stk[sp++] = w[3];
            }
        }
    }
}

// variant (round 4, idea M3): at a node visit the hit children that come FIRST in the ray's order and are leaves are tested at once (in parallel
// over the group's lanes on the GPU) instead of being pushed and popped one by one.  node_visits = loop iterations with a node, leaf_visits =
// iterations that pop a leaf (the ones that were not leading), inl = leaves tested inline.
extern "C" void wide_sim_inline(const uint32_t* nodes, const float* leaves, int n_rays, const float* o, const float* d,
                                const float* tmin_init, int* node_visits, int* leaf_visits, int* inl, int all_leaves) {
    auto leaf_test = [&](const float* L, const float* ro, const float* rd, const double* inv, float& t_min) {
        double tn = -DBL_MAX, tf = DBL_MAX;
        for (int a = 0; a < 3; ++a) {
            double t0 = ((double)L[a] - ro[a]) * inv[a], t1 = ((double)L[3 + a] - ro[a]) * inv[a];
            if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
            tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
        }
        if (tn <= tf && tf > 0 && !(tn > t_min)) {
            const float* v0 = L + 6; const float* e1 = L + 9; const float* e2 = L + 12;
            const double px = rd[1] * e2[2] - rd[2] * e2[1], py = rd[2] * e2[0] - rd[0] * e2[2], pz = rd[0] * e2[1] - rd[1] * e2[0];
            const double det = e1[0] * px + e1[1] * py + e1[2] * pz;
            if (det != 0) {
                const double id = 1.0 / det, sx = ro[0] - v0[0], sy = ro[1] - v0[1], sz = ro[2] - v0[2];
                const double u = (sx * px + sy * py + sz * pz) * id;
                const double qx = sy * e1[2] - sz * e1[1], qy = sz * e1[0] - sx * e1[2], qz = sx * e1[1] - sy * e1[0];
                const double v = (rd[0] * qx + rd[1] * qy + rd[2] * qz) * id, t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * id;
                if (!(t < 0 || u < 0 || v < 0 || 1 - u - v < 0) && t_min > t) t_min = (float)t;
            }
        }
    };
    for (int r = 0; r < n_rays; ++r) {
        const float* ro = o + 3 * r; const float* rd = d + 3 * r;
        const double inv[3] = {1.0 / rd[0], 1.0 / rd[1], 1.0 / rd[2]};
        float t_min = tmin_init[r];
        const int dir = mtbvh_dir(rd);
        uint32_t stk[64]; int sp = 0; uint32_t cur = 0; int nv = 0, nl = 0, ni = 0;
        for (;;) {
            if (!(cur & 0x80000000u)) {
                ++nv;
                const uint32_t* nd = nodes + 36 * (size_t)cur;
                float p[3]; std::memcpy(p, nd, 12);
                double sc[3];
                for (int a = 0; a < 3; ++a) { uint32_t e = ((nd[3] >> (8 * a)) & 255u) << 23; float f; std::memcpy(&f, &e, 4); sc[a] = f; }
                const int nc = (int)(nd[3] >> 24);
                uint32_t H = 0, LM = 0; uint32_t before[8]; uint32_t ref[8];
                for (int c = 0; c < nc; ++c) {
                    const uint32_t* w = nd + 4 + 4 * c;
                    const int qlo[3] = {(int)(w[0] & 255), (int)((w[0] >> 8) & 255), (int)((w[0] >> 16) & 255)};
                    const int qhi[3] = {(int)(w[0] >> 24), (int)(w[1] & 255), (int)((w[1] >> 8) & 255)};
                    const uint64_t ww = ((uint64_t)w[2] << 32) | w[1];
                    before[c] = (uint32_t)(ww >> (16 + 8 * dir)) & 255u; ref[c] = w[3];
                    if (ref[c] & 0x80000000u) LM |= 1u << c;
                    double tn = -DBL_MAX, tf = DBL_MAX;
                    for (int a = 0; a < 3; ++a) {
                        const double lo = p[a] + qlo[a] * sc[a], hi = p[a] + qhi[a] * sc[a];
                        double t0 = (lo - ro[a]) * inv[a], t1 = (hi - ro[a]) * inv[a];
                        if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
                        tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
                    }
                    if (tn <= (tf < t_min ? tf : t_min) && tf >= 0) H |= 1u << c;
                }
                // leading leaves (all_leaves: every hit leaf whose predecessors are leaves) are tested here, in visit order
                uint32_t inl_mask = 0;
                for (int c = 0; c < nc; ++c) if ((H & LM & (1u << c)) && (H & before[c] & ~LM) == 0) inl_mask |= 1u << c;
                if (!all_leaves) { /* only when ALL hit children are leading leaves or the node is a bottom node */ }
                int order[8], no = 0;
                for (int c = 0; c < nc; ++c) if (inl_mask & (1u << c)) order[no++] = c;
                for (int a = 0; a < no; ++a) for (int b2 = a + 1; b2 < no; ++b2)
                    if (__builtin_popcount(H & before[order[b2]]) < __builtin_popcount(H & before[order[a]])) { const int t = order[a]; order[a] = order[b2]; order[b2] = t; }
                for (int a = 0; a < no; ++a) { leaf_test(leaves + 16 * (size_t)(ref[order[a]] & 0x7fffffffu), ro, rd, inv, t_min); ++ni; }
                const uint32_t R = H & ~inl_mask;
                const int nh = __builtin_popcount(R);
                for (int c = 0; c < nc; ++c) if (R & (1u << c)) stk[sp + nh - 1 - __builtin_popcount(R & before[c])] = ref[c];
                sp += nh;
            } else {
                ++nl;
                leaf_test(leaves + 16 * (size_t)(cur & 0x7fffffffu), ro, rd, inv, t_min);
            }
            if (sp == 0) break;
            cur = stk[--sp];
        }
        node_visits[r] = nv; leaf_visits[r] = nl; inl[r] = ni;
    }
}

// variant (round 4, idea "leaf clusters"): a node whose children are ALL leaves is not fetched at all — the parent's reference names its leaves
// (consecutive records), which are fetched together and tested in parallel over the group's lanes.  trips = dependent round trips of the ray:
// inner-node visits + cluster visits (one each) + single leaves popped from mixed nodes.  cl = cluster visits, cl_leaves = leaves tested in them.
extern "C" void wide_sim_clusters(const uint32_t* nodes, const float* leaves, int n_rays, const float* o, const float* d,
                                  const float* tmin_init, int* trips_out, int* cl_out, int* cl_leaves_out) {
    auto leaf_test = [&](const float* L, const float* ro, const float* rd, const double* inv, float& t_min) {
        double tn = -DBL_MAX, tf = DBL_MAX;
        for (int a = 0; a < 3; ++a) {
            double t0 = ((double)L[a] - ro[a]) * inv[a], t1 = ((double)L[3 + a] - ro[a]) * inv[a];
            if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
            tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
        }
        if (tn <= tf && tf > 0 && !(tn > t_min)) {
            const float* v0 = L + 6; const float* e1 = L + 9; const float* e2 = L + 12;
            const double px = rd[1] * e2[2] - rd[2] * e2[1], py = rd[2] * e2[0] - rd[0] * e2[2], pz = rd[0] * e2[1] - rd[1] * e2[0];
            const double det = e1[0] * px + e1[1] * py + e1[2] * pz;
            if (det != 0) {
                const double id = 1.0 / det, sx = ro[0] - v0[0], sy = ro[1] - v0[1], sz = ro[2] - v0[2];
                const double u = (sx * px + sy * py + sz * pz) * id;
                const double qx = sy * e1[2] - sz * e1[1], qy = sz * e1[0] - sx * e1[2], qz = sx * e1[1] - sy * e1[0];
                const double v = (rd[0] * qx + rd[1] * qy + rd[2] * qz) * id, t = (e2[0] * qx + e2[1] * qy + e2[2] * qz) * id;
                if (!(t < 0 || u < 0 || v < 0 || 1 - u - v < 0) && t_min > t) t_min = (float)t;
            }
        }
    };
    auto all_leaves = [&](uint32_t ni) { const uint32_t* nd = nodes + 36 * (size_t)ni; const int nc = (int)(nd[3] >> 24); for (int c = 0; c < nc; ++c) if (!(nd[4 + 4 * c + 3] & 0x80000000u)) return false; return true; };
    for (int r = 0; r < n_rays; ++r) {
        const float* ro = o + 3 * r; const float* rd = d + 3 * r;
        const double inv[3] = {1.0 / rd[0], 1.0 / rd[1], 1.0 / rd[2]};
        float t_min = tmin_init[r];
        const int dir = mtbvh_dir(rd);
        uint32_t stk[64]; int sp = 0; uint32_t cur = 0; int trips = 0, ncl = 0, nll = 0;
        for (;;) {
            ++trips;
            if (!(cur & 0x80000000u)) {
                const uint32_t* nd = nodes + 36 * (size_t)cur;
                const int nc = (int)(nd[3] >> 24);
                if (cur != 0 && all_leaves(cur)) {                 // cluster: every leaf fetched and tested, no child-box tests
                    ++ncl; nll += nc;
                    int order[8];
                    for (int c = 0; c < nc; ++c) order[c] = c;
                    uint32_t bef[8];
                    for (int c = 0; c < nc; ++c) { const uint32_t* w = nd + 4 + 4 * c; const uint64_t ww = ((uint64_t)w[2] << 32) | w[1]; bef[c] = (uint32_t)(ww >> (16 + 8 * dir)) & 255u; }
                    for (int a = 0; a < nc; ++a) for (int b2 = a + 1; b2 < nc; ++b2)
                        if (__builtin_popcount(bef[order[b2]]) < __builtin_popcount(bef[order[a]])) { const int t = order[a]; order[a] = order[b2]; order[b2] = t; }
                    for (int a = 0; a < nc; ++a) leaf_test(leaves + 16 * (size_t)(nd[4 + 4 * order[a] + 3] & 0x7fffffffu), ro, rd, inv, t_min);
                } else {
                    float p[3]; std::memcpy(p, nd, 12);
                    double sc[3];
                    for (int a = 0; a < 3; ++a) { uint32_t e = ((nd[3] >> (8 * a)) & 255u) << 23; float f; std::memcpy(&f, &e, 4); sc[a] = f; }
                    uint32_t H = 0; uint32_t before[8]; uint32_t ref[8];
                    for (int c = 0; c < nc; ++c) {
                        const uint32_t* w = nd + 4 + 4 * c;
                        const int qlo[3] = {(int)(w[0] & 255), (int)((w[0] >> 8) & 255), (int)((w[0] >> 16) & 255)};
                        const int qhi[3] = {(int)(w[0] >> 24), (int)(w[1] & 255), (int)((w[1] >> 8) & 255)};
                        const uint64_t ww = ((uint64_t)w[2] << 32) | w[1];
                        before[c] = (uint32_t)(ww >> (16 + 8 * dir)) & 255u; ref[c] = w[3];
                        double tn = -DBL_MAX, tf = DBL_MAX;
                        for (int a = 0; a < 3; ++a) {
                            const double lo = p[a] + qlo[a] * sc[a], hi = p[a] + qhi[a] * sc[a];
                            double t0 = (lo - ro[a]) * inv[a], t1 = (hi - ro[a]) * inv[a];
                            if (t0 > t1) { const double t = t0; t0 = t1; t1 = t; }
                            tn = tn > t0 ? tn : t0; tf = tf < t1 ? tf : t1;
                        }
                        if (tn <= (tf < t_min ? tf : t_min) && tf >= 0) H |= 1u << c;
                    }
                    const int nh = __builtin_popcount(H);
                    for (int c = 0; c < nc; ++c) if (H & (1u << c)) stk[sp + nh - 1 - __builtin_popcount(H & before[c])] = ref[c];
                    sp += nh;
                }
            } else {
                leaf_test(leaves + 16 * (size_t)(cur & 0x7fffffffu), ro, rd, inv, t_min);
            }
            if (sp == 0) break;
            cur = stk[--sp];
        }
        trips_out[r] = trips; cl_out[r] = ncl; cl_leaves_out[r] = nll;
    }
}
