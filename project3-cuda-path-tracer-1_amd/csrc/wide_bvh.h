// wide_bvh.h — 8-wide, quantised, treelet-collapsed form of the reference's BVH (host builder; device layout in
// ptk_device.hpp: WideNode / WideLeaf).  Not a different tree: the SAME binary SAH tree the reference builds
// (BVH.cpp:13-92) and threads for six directions (BVH.cpp:180-239), cut into wide nodes: the children of a wide node are a frontier of the binary
// subtree below its root, chosen by dynamic programming to minimise the expected node visits (wide_bvh.cpp: CollapsePlan; round 5 — the bunny: 2 058
// nodes, 5.9 of 8 slots used, against 3 202 / 4.1 for the round-2 "open the largest child" expansion, PTK_WIDE_COLLAPSE=greedy).
//
// Why this is parity-preserving ("wide BVH"): in the reference's walk (pathtrace.cu:236-279) only the
// LEAF visits have an effect — the exact box test of the leaf, the `tEnter > t_min` cut against the current t_min, the
// triangle test and the strict `t_min > t` update.  A child box is a subset of its parent's (unions of the same fp32
// triangle bounds) and the slab test is monotone in the plane positions in fp32, so whenever a leaf passes its own test
// every ancestor passes too: inner-node tests only prune, they never change a result.  Hence ANY conservative pruning
// above the leaves gives bit-identical hits as long as the leaves that pass are tested in the reference's order — the
// depth-first order with the near child (the direction's `hit` link) first.  The wide node stores that order, per
// direction, as the set of children that precede each child.
#pragma once
#include <cstdint>
#include <vector>
#include "scene.h"

namespace ptk {

struct WideNodeHost { uint32_t w[36]; };                 // 144 B = 9 x uint4, see ptk_device.hpp
struct WideNode16Host { uint32_t w[68]; };               // 272 B = 17 x uint4: the 16-wide variant (round 5, PTK_WIDE_WIDTH=16), see ptk_device.hpp
struct WideLeafHost { float lo[3], hi[3], v0[3], e1[3], e2[3]; int tri; };   // 64 B

struct WideBvh {
    std::vector<WideNodeHost> nodes;                     // breadth-first: the top levels are the first entries
    std::vector<WideNode16Host> nodes16;                 // ... or these, when built with width 16 (exactly one of the two is filled)
    int width = 8;
    std::vector<WideLeafHost> leaves;                    // children of one node are consecutive
    int depth = 0;                                       // levels of wide nodes
    int max_stack = 0;                                   // upper bound of the per-ray stack the traversal kernel needs
    bool ok = false;
};

// mt: the 6 * B direction-ordered nodes of Scene::gpuBVHNodes (preorder, same boxes in every copy).  Returns false (and
// leaves `out.ok` false) if the table is not a preorder binary tree with one triangle per leaf; callers then keep the
// threaded walk.
bool build_wide_bvh(const std::vector<GpuBVHNode>& mt, int B, const std::vector<Triangle>& tris, WideBvh& out, int width = 8);

} // namespace ptk
