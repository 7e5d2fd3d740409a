// fast_tree.h -- traversal-only tree of a triangle mesh (fast_tree.cpp).
#ifndef RT_FAST_TREE_H
#define RT_FAST_TREE_H

#include <string>
#include <vector>
#include "device_types.h"

namespace rt {

// Builds a 4-wide tree over the n triangles tris[tri_ids[0 .. n-1]] and appends its nodes to `qnodes` (`qdepth`: depth of
// every appended node below the tree's root, 255 for padding nodes).  Leaves are RT_REF_TRI | tri_ids[i]; a triangle
// may be the child of several nodes, each time with the box of the part of it that lies on that node's side of the
// spatial splits above it.  root_out: index of the root in qnodes; stack_need: stack entries a nearest-first
// traversal below the root can need.  Returns RT_OK or a negative RT_E_* code with `err` set.
int build_fast_tree(const DTri* tris, const uint32_t* tri_ids, uint64_t n, std::vector<QNode>& qnodes, std::vector<uint8_t>& qdepth,
                    uint32_t& root_out, int32_t& stack_need, std::string& err);

} // namespace rt
#endif
