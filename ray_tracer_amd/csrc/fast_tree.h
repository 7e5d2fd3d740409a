// fast_tree.h -- traversal-only tree of a triangle mesh (fast_tree.cpp).
#ifndef RT_FAST_TREE_H
#define RT_FAST_TREE_H

#include <string>
#include <vector>
#include "flatten.h"

namespace rt {

struct FastTreeResult {
    uint32_t root = 0;        // index of the tree's root in FlatScene::qnodes / cqnodes (its packed form starts at pnodes[0])
    int32_t stack_need = 0;   // stack entries a nearest-first traversal below the root can need
    uint64_t wide_nodes = 0, references = 0;
    double leaf_area = 0.0;   // sum of the leaf boxes' half areas
    double ratio_sum = 0.0;   // over the child boxes of the top four levels: area of the 8-bit box / area of the f32 box (each capped at 16)
    uint64_t ratio_n = 0;
};

// Builds a 4-wide tree over the n triangles tris[tri_ids[0 .. n-1]] and writes it in every format the kernels read:
// appended to f.qnodes and f.cqnodes (same indices; `qdepth`: depth of every appended node below the tree's root), and
// as f.pnodes / f.ptris (left empty when the mesh is too large for their 32-bit byte offsets).  Reads nothing of `f`
// but the sizes of those arrays, so `f` may be a scene of its own that the caller merges later.  Leaves are
// RT_REF_TRI | tri_ids[i]; a triangle may be the child of several nodes, each time with the box of the part of it that
// lies on that node's side of the spatial splits above it.  Returns RT_OK or a negative RT_E_* code with `err` set;
// RT_E_UNSUPPORTED means that this mesh gets no such tree (a box that is not finite in f32) and nothing was appended.
int build_fast_tree(const DTri* tris, FlatScene& f, std::vector<uint8_t>& qdepth, const uint32_t* tri_ids, uint64_t n, FastTreeResult& out, std::string& err);

} // namespace rt
#endif
