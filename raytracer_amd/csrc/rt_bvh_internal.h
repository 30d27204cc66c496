// rt_bvh_internal.h - host-side helpers shared by rt_bvh.cpp and rt_hip.hip (not part of the C-ABI).
#pragma once
#include <cstdint>
#include <vector>

// 48-byte node image of the reference (scripts/scenes/chunker.py:247-257); same layout as Std140Node.
struct RtNode48 {
    float lo[3];
    float pad;
    float hi[3];
    int32_t id, c1, c2, off, cnt;
};
static_assert(sizeof(RtNode48) == 48, "std140 GraphNode");

// Binned-SAH BVH over `n` triangles (pos9 = 9 floats each) for the opt-in performance structure
// (RT_ACCEL_SAH): leaves of at most max_leaf triangles, depth <= max_depth.  Appends nodes (ids continue
// at nodes.size(), children after their parent) and leaf entries (tri_ids[i] of the triangles, in
// leaf order) to the output vectors; returns the id of the root.  The reference's split rule
// (rt_bvh_build) stays the parity structure.
int rt_sah_build(const float* pos9, const int32_t* tri_ids, int n, int max_depth, int max_leaf, std::vector<RtNode48>& nodes,
                 std::vector<int32_t>& leaf_ids);
