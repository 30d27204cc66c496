// rt_bvh.cpp - host-side BVH construction and refit with the reference's split rule.
//
// Native twin of scripts/scenes/chunker.py of the reference (:17-25 chunk_mesh_bvh, :113-245
// BVHParentNode, :197-214 refit, :79-86 flat leaf id list, :247-257 the 48-byte node image);
// the numpy restatement raytracer_amd/scripts/scenes/chunker.py is its oracle (same topology,
// ids, boxes and leaf lists byte for byte, tests/test_bvh_native.py).  The Python builder of
// the reference is unusable at 10^6 triangles; this one builds the
// subdivided-Suzanne configuration in about a second.
//
// Rule: a node stops at <= 5 triangles or depth 20; else the longest axis of
// its box is halved and a triangle goes LEFT iff its centroid ((A+B)+C)/3 in float32 lies in
// the lower half-box, inclusive on all six faces; if one side would be empty the node stays a
// leaf.  Both children get their ids (left, then right) before either subtree is descended.
//
// Compiled with -ffp-contract=off like everything else: centroid and half-box arithmetic must
// round exactly as numpy's float32 does.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt_hip.h"

namespace {

struct Node48 {
    float lo[3];
    float pad;
    float hi[3];
    int32_t id, c1, c2, off, cnt;
};
static_assert(sizeof(Node48) == 48, "std140 GraphNode");

constexpr int kMaxLeaf = 5;
constexpr int kMaxDepth = 20;

struct Builder {
    const float* pos;                 // n x 9
    std::vector<float> cen;           // n x 3, frozen centroids
    std::vector<Node48> nodes;        // creation order
    std::vector<std::vector<int>> leaf_tris;   // per node (kept for leaves only)
    int32_t first_node_id;

    void box_of(const std::vector<int>& tris, float lo[3], float hi[3]) const {
        if (tris.empty()) { lo[0] = lo[1] = lo[2] = hi[0] = hi[1] = hi[2] = 0.0f; return; }
        for (int c = 0; c < 3; c++) { lo[c] = INFINITY; hi[c] = -INFINITY; }
        for (int t : tris) {
            const float* p = pos + 9 * (size_t)t;
            for (int v = 0; v < 3; v++)
                for (int c = 0; c < 3; c++) {
                    const float x = p[3 * v + c];
                    if (x < lo[c]) lo[c] = x;
                    if (x > hi[c]) hi[c] = x;
                }
        }
    }

    int make_node(std::vector<int>&& tris) {
        Node48 n;
        std::memset(&n, 0, sizeof(n));
        box_of(tris, n.lo, n.hi);
        n.id = first_node_id + (int32_t)nodes.size();
        n.c1 = n.c2 = -1;
        n.off = -1;
        n.cnt = (int32_t)tris.size();
        nodes.push_back(n);
        leaf_tris.push_back(std::move(tris));
        return (int)nodes.size() - 1;
    }

    void split_recursively(int ni, int depth) {
        if ((int)leaf_tris[ni].size() <= kMaxLeaf || depth == kMaxDepth) return;
        const float* lo = nodes[ni].lo;
        const float* hi = nodes[ni].hi;
        int axis = 0;
        float best = std::fabs(lo[0] - hi[0]);
        for (int c = 1; c < 3; c++) {
            const float e = std::fabs(lo[c] - hi[c]);
            if (e > best) { best = e; axis = c; }       // first maximum wins, like argmax
        }
        float left_hi[3];
        for (int c = 0; c < 3; c++) {
            const float half = (c == axis) ? 0.5f : 0.0f;
            left_hi[c] = hi[c] - (hi[c] - lo[c]) * half;
        }
        std::vector<int> left, right;
        for (int t : leaf_tris[ni]) {
            const float* q = &cen[3 * (size_t)t];
            const bool inside = q[0] >= lo[0] && q[1] >= lo[1] && q[2] >= lo[2] && q[0] <= left_hi[0] && q[1] <= left_hi[1] &&
                                q[2] <= left_hi[2];
            (inside ? left : right).push_back(t);
        }
        if (left.empty() || right.empty()) return;       // stays an (oversized) leaf
        std::vector<int>().swap(leaf_tris[ni]);          // interior nodes do not keep their list
        const int l = make_node(std::move(left));
        const int r = make_node(std::move(right));
        nodes[ni].c1 = nodes[l].id;
        nodes[ni].c2 = nodes[r].id;
        split_recursively(l, depth + 1);
        split_recursively(r, depth + 1);
    }
};

}  // namespace

extern "C" {

int rt_bvh_build(const float* pos9, int n_tris, int32_t first_node_id, int32_t first_tri_id, int32_t first_leaf_offset,
                 void* nodes48_out, int nodes_cap, int32_t* leaf_tri_ids_out, int32_t* n_nodes_out) {
    if (!pos9 || n_tris <= 0 || !nodes48_out || !leaf_tri_ids_out || !n_nodes_out || nodes_cap <= 0) return RT_E_ARG;
    Builder b;
    b.pos = pos9;
    b.first_node_id = first_node_id;
    b.cen.resize(3 * (size_t)n_tris);
    for (int t = 0; t < n_tris; t++) {
        const float* p = pos9 + 9 * (size_t)t;
        for (int c = 0; c < 3; c++) b.cen[3 * (size_t)t + c] = ((p[c] + p[3 + c]) + p[6 + c]) / 3.0f;
    }
    std::vector<int> all((size_t)n_tris);
    for (int t = 0; t < n_tris; t++) all[t] = t;
    b.nodes.reserve(2 * (size_t)n_tris / 3 + 16);
    const int root = b.make_node(std::move(all));
    b.split_recursively(root, 0);
    if ((int)b.nodes.size() > nodes_cap) return RT_E_LIMIT;
    // flat leaf id list: leaves in creation order (chunker.py:79-86)
    int32_t off = first_leaf_offset;
    size_t w = 0;
    for (size_t i = 0; i < b.nodes.size(); i++) {
        Node48& n = b.nodes[i];
        if (n.c1 < 0 && n.c2 < 0) {
            n.off = off;
            for (int t : b.leaf_tris[i]) leaf_tri_ids_out[w++] = first_tri_id + t;
            off += n.cnt;
        }
    }
    if (w != (size_t)n_tris) return RT_E_STATE;          // every triangle is in exactly one leaf
    std::memcpy(nodes48_out, b.nodes.data(), b.nodes.size() * sizeof(Node48));
    *n_nodes_out = (int32_t)b.nodes.size();
    return RT_OK;
}

int rt_bvh_refit(const float* pos9, int n_tris, int32_t first_tri_id, int32_t first_leaf_offset, void* nodes48, int n_nodes,
                 const int32_t* leaf_tri_ids) {
    if (!pos9 || n_tris <= 0 || !nodes48 || n_nodes <= 0 || !leaf_tri_ids) return RT_E_ARG;
    Node48* nodes = (Node48*)nodes48;
    const int32_t id0 = nodes[0].id;
    // children have larger ids than their parent: one reverse sweep is bottom-up
    for (int i = n_nodes - 1; i >= 0; i--) {
        Node48& n = nodes[i];
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (n.c1 < 0 && n.c2 < 0) {
            const int64_t s = (int64_t)n.off - first_leaf_offset;
            if (s < 0 || s + n.cnt > n_tris) return RT_E_ARG;
            for (int k = 0; k < n.cnt; k++) {
                const int64_t t = (int64_t)leaf_tri_ids[s + k] - first_tri_id;
                if (t < 0 || t >= n_tris) return RT_E_ARG;
                const float* p = pos9 + 9 * (size_t)t;
                for (int v = 0; v < 3; v++)
                    for (int c = 0; c < 3; c++) {
                        const float x = p[3 * v + c];
                        if (x < lo[c]) lo[c] = x;
                        if (x > hi[c]) hi[c] = x;
                    }
            }
        } else {
            const int32_t ch[2] = {n.c1, n.c2};
            for (int q = 0; q < 2; q++) {
                if (ch[q] < 0) continue;
                const int64_t j = (int64_t)ch[q] - id0;
                if (j <= i || j >= n_nodes) return RT_E_ARG;
                for (int c = 0; c < 3; c++) {
                    if (nodes[j].lo[c] < lo[c]) lo[c] = nodes[j].lo[c];
                    if (nodes[j].hi[c] > hi[c]) hi[c] = nodes[j].hi[c];
                }
            }
        }
        for (int c = 0; c < 3; c++) { n.lo[c] = lo[c]; n.hi[c] = hi[c]; }
    }
    return RT_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// Binned SAH builder for the opt-in performance structure (rt_set_accel(RT_ACCEL_SAH)).  Not a
// restatement of anything in the reference: its BVH (above) costs 12 triangle tests per ray on
// the benchmark scene because a 12-triangle wall slab ends up as two leaves that both span the
// whole slab.  Topology only: boxes are refit on the device from the posed triangles.
// ---------------------------------------------------------------------------------------------
#include <algorithm>

#include "rt_bvh_internal.h"

namespace {

struct SahPrim { float lo[3], hi[3], c[3]; int32_t id; };
struct SahBox {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    void grow(const float* l, const float* h) {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    float half_area() const {
        const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return (dx < 0 || dy < 0 || dz < 0) ? 0.f : dx * dy + dy * dz + dz * dx;
    }
};

struct SahBuilder {
    std::vector<SahPrim> prims;
    std::vector<RtNode48>* nodes;
    std::vector<int32_t>* leaf_ids;
    int max_depth, max_leaf;

    int build(int b, int e, int depth) {
        const int ni = (int)nodes->size();
        RtNode48 nd;
        std::memset(&nd, 0, sizeof(nd));
        SahBox box, cbox;
        for (int i = b; i < e; i++) { box.grow(prims[i].lo, prims[i].hi); cbox.grow(prims[i].c, prims[i].c); }
        for (int a = 0; a < 3; a++) { nd.lo[a] = box.lo[a]; nd.hi[a] = box.hi[a]; }
        nd.id = ni; nd.c1 = nd.c2 = -1; nd.off = -1; nd.cnt = e - b;
        nodes->push_back(nd);
        const int n = e - b;
        if (n <= max_leaf || depth >= max_depth) return make_leaf(ni, b, e);
        // best of 3 axes x 15 bin boundaries
        constexpr int NB = 16;
        float best_cost = INFINITY;
        int best_axis = -1, best_split = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cbox.hi[a] - cbox.lo[a];
            if (!(ext > 0.f)) continue;
            SahBox bb[NB];
            int cnt[NB] = {0};
            const float scale = NB / ext;
            for (int i = b; i < e; i++) {
                int k = std::min(NB - 1, std::max(0, (int)((prims[i].c[a] - cbox.lo[a]) * scale)));
                cnt[k]++;
                bb[k].grow(prims[i].lo, prims[i].hi);
            }
            float right_area[NB];
            int right_cnt[NB];
            SahBox acc;
            int c = 0;
            for (int k = NB - 1; k > 0; k--) { acc.grow(bb[k].lo, bb[k].hi); c += cnt[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
            SahBox left;
            int lc = 0;
            for (int k = 0; k < NB - 1; k++) {
                left.grow(bb[k].lo, bb[k].hi);
                lc += cnt[k];
                if (lc == 0 || right_cnt[k + 1] == 0) continue;
                const float cost = left.half_area() * lc + right_area[k + 1] * right_cnt[k + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = k; }
            }
        }
        int mid;
        if (best_axis < 0) {                               // all centroids coincide: split by count
            mid = b + n / 2;
        } else {
            const float ext = cbox.hi[best_axis] - cbox.lo[best_axis], scale = NB / ext, lo0 = cbox.lo[best_axis];
            const int a = best_axis, s = best_split;
            mid = (int)(std::partition(prims.begin() + b, prims.begin() + e, [&](const SahPrim& p) {
                      return std::min(NB - 1, std::max(0, (int)((p.c[a] - lo0) * scale))) <= s;
                  }) - prims.begin());
            if (mid == b || mid == e) mid = b + n / 2;
        }
        const int l = build(b, mid, depth + 1);
        const int r = build(mid, e, depth + 1);
        (*nodes)[ni].c1 = l;
        (*nodes)[ni].c2 = r;
        return ni;
    }
    int make_leaf(int ni, int b, int e) {
        (*nodes)[ni].off = (int32_t)leaf_ids->size();
        for (int i = b; i < e; i++) leaf_ids->push_back(prims[i].id);
        return ni;
    }
};

}  // namespace

int rt_sah_build(const float* pos9, const int32_t* tri_ids, int n, int max_depth, int max_leaf, std::vector<RtNode48>& nodes,
                 std::vector<int32_t>& leaf_ids) {
    SahBuilder sb;
    sb.max_leaf = max_leaf;
    sb.nodes = &nodes;
    sb.leaf_ids = &leaf_ids;
    sb.max_depth = max_depth;
    sb.prims.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        SahPrim& p = sb.prims[i];
        const float* q = pos9 + 9 * (size_t)i;
        for (int a = 0; a < 3; a++) {
            p.lo[a] = std::min(q[a], std::min(q[3 + a], q[6 + a]));
            p.hi[a] = std::max(q[a], std::max(q[3 + a], q[6 + a]));
            p.c[a] = 0.5f * (p.lo[a] + p.hi[a]);
        }
        p.id = tri_ids[i];
    }
    return sb.build(0, n, 0);
}
