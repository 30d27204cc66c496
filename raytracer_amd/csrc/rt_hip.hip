// rt_hip.hip - kernels and C-ABI of librt_hip.so (gfx950 / MI355X only).
//
// Kernels
//   k_pose        device twin of numba update_triangles_to_csys + triangles_to_array
//                 (scripts/numba_utils/classes.py:195-224, 315-342): rest pose -> world pose,
//                 plus the per-triangle plane constants the intersection test needs
//   k_refit_level device twin of the chunker refit (scripts/scenes/chunker.py:197-214),
//                 bottom-up by tree level (exact: min/max are associative)
//   k_pack_nodes  copies refreshed child boxes into the 64-byte traversal records
//   k_tri_flags   refreshes the per-triangle flag word after rt_set_materials
//   k_trace       ray generation + traversal + bounce loop + accumulation (shaders/raytracer.fs.glsl
//                 main() :928-1042 and everything it calls): persistent waves, wave-voted node /
//                 triangle / path steps (rt_dev_trace.h); a wave folds the passes of the pixels
//                 it traced into the accumulator itself, in pass order; FIRST_HIT instantiation =
//                 primary-ray closest hit only
//   k_gather_*    read-back of posed triangles / refit boxes in the reference's std140 order
// Host: scene upload (traversal-order slots, 64-byte node records, top-level tree for scenes with
// more than 32 meshes), batched pose/refit, launch logic.  rt_bvh.cpp holds the BVH builder.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DRT_WAVES=6
// (see __graft_entry__.py).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "../../include/rt_hip.h"
#include "rt_bvh_internal.h"
#include "rt_dev_trace.h"

// ----------------------------------------------------------------------------------------
// device kernels
// ----------------------------------------------------------------------------------------
struct PoseArgs {
    const float4* rest;      // 6 x float4 per slot: A, B, C, nA, nB, nC (rest pose)
    const int* slot_mesh;    // mesh of each slot
    const int* slot_rem;     // triangles left in the leaf after this one
    const int* slot_rank;    // rank of each slot in the reference's traversal order
    const float* xf;         // 16 floats per mesh, row-major, translation in row 3
    const unsigned char* dirty;  // per mesh: 1 = re-pose (0 = leave), 2 = copy rest (no transform yet)
    const MatDev* mats;      // per mesh (transparentFromBehind goes into the triangle's flag word)
    float4* geo;
    float4* nrm;
    int n_slots;
};

__global__ __launch_bounds__(256) void k_pose(PoseArgs a) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.n_slots) return;
    int mesh = a.slot_mesh[k];
    unsigned char mode = a.dirty[mesh];
    if (mode == 0) return;
    const float4* r = a.rest + 6 * (size_t)k;
    float p[3][3], n[3][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        float4 pv = r[c], nv = r[3 + c];
        if (mode == 2) {
            p[c][0] = pv.x; p[c][1] = pv.y; p[c][2] = pv.z;
            n[c][0] = nv.x; n[c][1] = nv.y; n[c][2] = nv.z;
        } else {
            const float* T = a.xf + 16 * (size_t)mesh;
#pragma unroll
            for (int j = 0; j < 3; j++) {
                // [p,1] . T, k-ordered float32 sum, no fma (DESIGN.md arithmetic contract)
                p[c][j] = ((pv.x * T[0 + j] + pv.y * T[4 + j]) + pv.z * T[8 + j]) + T[12 + j];
                n[c][j] = (nv.x * T[0 + j] + nv.y * T[4 + j]) + nv.z * T[8 + j];
            }
        }
    }
    v3 A = V3(p[0][0], p[0][1], p[0][2]), B = V3(p[1][0], p[1][1], p[1][2]), C = V3(p[2][0], p[2][1], p[2][2]);
    v3 N = cross3(B - A, C - A);                 // raytracer.fs.glsl:316-319
    float d = -dot3(N, A);                       // :345
    float4* g = a.geo + 4 * (size_t)k;
    g[0] = make_float4(A.x, A.y, A.z, N.x);
    g[1] = make_float4(B.x, B.y, B.z, N.y);
    g[2] = make_float4(C.x, C.y, C.z, N.z);
    const int flags = mesh | (a.mats[mesh].transparentFromBehind ? RT_TFB_BIT : 0);
    const int next = a.slot_rem[k] > 0 ? (int)((uint32_t)(k + 1) << 6) : -1;   // byte offset of the leaf's next slot (rt_dev_trace.h)
    g[3] = make_float4(d, __int_as_float(next), __int_as_float(a.slot_rank[k]), __int_as_float(flags));
    // hit record (rt_dev_trace.h): normals + mesh; the box fields (.w of [1] and [2], all of [3]) belong to k_ref_leaf_boxes,
    // which runs after this kernel in every prep of an RT_ACCEL_SAH scene
    float4* o = a.nrm + 4 * (size_t)k;
    o[0] = make_float4(n[0][0], n[0][1], n[0][2], __int_as_float(mesh));
    o[1] = make_float4(n[1][0], n[1][1], n[1][2], 0.0f);
    o[2] = make_float4(n[2][0], n[2][1], n[2][2], 0.0f);
}

// materials changed: refresh the flag word of every slot (mesh index | transparentFromBehind)
__global__ __launch_bounds__(256) void k_tri_flags(const int* slot_mesh, const MatDev* mats, float4* geo, int n_slots) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    const int mesh = slot_mesh[k];
    geo[4 * (size_t)k + 3].w = __int_as_float(mesh | (mats[mesh].transparentFromBehind ? RT_TFB_BIT : 0));
}

// visited-node table: c1/c2 = child visit indices (-1 none); leaves: [start, start+count) slots
struct VNode { int c1, c2, start, count; };

// SAH structure only: how far outside a triangle rayTriangle can accept a point.  An edge function N.((B-A)x(P-A)) carries a
// rounding error of <= 14 x 2^-24 |N||B-A||P-A|, i.e. it accepts points up to 14 x 2^-24 D outside its edge
// (D = the longest edge) - and at a vertex of angle theta up to 14 x 2^-24 D / sin(theta) beyond the tip, along the adjacent
// edge, which for a sliver exceeds any fixed slack.  So every box carries, on top of its slack, the largest
// 2^-18 D + 2^-19 D / sin(theta_min) of the triangles below it (sin(theta_min) = |N| / product of the two longest edges; a few
// 10^-6 D for ordinary triangles): the refit hands it up in the free .w lane of the box minima, k_pack_nodes adds it to the
// slack.  It must widen the TEST, not the stored box: a stored box that is larger by even 10^-6 is no longer left at
// t < 5e-7 by the bounce ray that starts on it, and culling rule (b) stops discarding it (measured: 40.7 -> 51.4 ms).  A
// triangle with |N| = 0 is never hit (|N.d| < 1e-6).  0 for the reference's own structure: its tests are the reference's.
__device__ __forceinline__ float sliver_tip_pad(float4 g0, float4 g1, float4 g2) {
    const float ab = (g1.x - g0.x) * (g1.x - g0.x) + (g1.y - g0.y) * (g1.y - g0.y) + (g1.z - g0.z) * (g1.z - g0.z);
    const float bc = (g2.x - g1.x) * (g2.x - g1.x) + (g2.y - g1.y) * (g2.y - g1.y) + (g2.z - g1.z) * (g2.z - g1.z);
    const float ca = (g0.x - g2.x) * (g0.x - g2.x) + (g0.y - g2.y) * (g0.y - g2.y) + (g0.z - g2.z) * (g0.z - g2.z);
    const float l1 = fmaxf(ab, fmaxf(bc, ca)), l3 = fminf(ab, fminf(bc, ca)), l2 = ab + bc + ca - l1 - l3;   // squared edge lengths, sorted
    const float n2 = g0.w * g0.w + g1.w * g1.w + g2.w * g2.w;           // |N|^2, N = (B-A)x(C-A) (k_pose)
    const float d = sqrtf(l1);
    if (!(n2 > 0.0f)) return 0.0f;
    const float pad = 0x1p-19f * d * sqrtf(l1 * fmaxf(l2, 0.0f) / n2) + 0x1p-18f * d;    // beyond a tip + outside an edge (<= 14 x 2^-24 D)
    return pad == pad ? pad : INFINITY;                                  // (not capped: an infinite slack only means "always enter")
}

__global__ __launch_bounds__(256) void k_refit_level(const VNode* vn, const int* level_nodes, int n, const float4* geo, float4* nbox, int sah) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int v = level_nodes[i];
    VNode nd = vn[v];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float pad = 0.0f;
    if (nd.c1 < 0 && nd.c2 < 0) {
        for (int k = nd.start; k < nd.start + nd.count; k++) {
            if (sah) pad = fmaxf(pad, sliver_tip_pad(geo[4 * (size_t)k], geo[4 * (size_t)k + 1], geo[4 * (size_t)k + 2]));
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float4 p = geo[4 * (size_t)k + c];
                lo[0] = p.x < lo[0] ? p.x : lo[0]; hi[0] = p.x > hi[0] ? p.x : hi[0];
                lo[1] = p.y < lo[1] ? p.y : lo[1]; hi[1] = p.y > hi[1] ? p.y : hi[1];
                lo[2] = p.z < lo[2] ? p.z : lo[2]; hi[2] = p.z > hi[2] ? p.z : hi[2];
            }
        }
    } else {
        int ch[2] = {nd.c1, nd.c2};
#pragma unroll
        for (int s = 0; s < 2; s++) {
            if (ch[s] < 0) continue;
            float4 a = nbox[2 * (size_t)ch[s]], b = nbox[2 * (size_t)ch[s] + 1];
            pad = fmaxf(pad, a.w);
            lo[0] = a.x < lo[0] ? a.x : lo[0]; hi[0] = b.x > hi[0] ? b.x : hi[0];
            lo[1] = a.y < lo[1] ? a.y : lo[1]; hi[1] = b.y > hi[1] ? b.y : hi[1];
            lo[2] = a.z < lo[2] ? a.z : lo[2]; hi[2] = b.z > hi[2] ? b.z : hi[2];
        }
    }
    nbox[2 * (size_t)v] = make_float4(lo[0], lo[1], lo[2], pad);
    nbox[2 * (size_t)v + 1] = make_float4(hi[0], hi[1], hi[2], 0.0f);
}

// interior record r <- boxes of its children; PackInfo is static topology (references as INDICES: >= 0 record, < 0 leaf with
// first slot ~ref, RT_IDX_NONE absent; the device form is made by dev_ref when the records are packed)
struct PackInfo { int c1, c2, ref1, ref2; };
__host__ __device__ __forceinline__ int dev_ref(int ref) {
    return ref == RT_IDX_NONE ? RT_NONE : ref >= 0 ? RT_REF_OF_RECORD(ref) : RT_REF_OF_LEAF(~ref);
}

// SAH structure: each box carries the slack of its widened slab test (rt_dev_trace.h,
// sah_widened_hit): e2 = 2 * (2^-18 of its largest coordinate + 2^-22 + the triangles' pad).  0 for the reference's
// structure, whose tests are the reference's own.
__device__ __forceinline__ float box_slack(float4 lo, float4 hi, int sah) {
    if (!sah) return 0.0f;
    const float m = fmaxf(fmaxf(fmaxf(fabsf(lo.x), fabsf(hi.x)), fmaxf(fabsf(lo.y), fabsf(hi.y))), fmaxf(fabsf(lo.z), fabsf(hi.z)));
    return 2.0f * (m * 0x1p-18f + 0x1p-22f + lo.w);       // lo.w: the subtree's triangle pad (k_refit_level)
}

// one thread per leaf of the REFERENCE's tree: box of its (posed) triangles, written to the hit record of every member slot
__global__ __launch_bounds__(256) void k_ref_leaf_boxes(const int* leaf_off, const int* leaf_slots, int n_leaves, const float4* geo,
                                                        float4* tri_hit) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_leaves) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int q = leaf_off[i]; q < leaf_off[i + 1]; q++) {
        const int k = leaf_slots[q];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float4 p = geo[4 * (size_t)k + c];
            lo[0] = p.x < lo[0] ? p.x : lo[0]; hi[0] = p.x > hi[0] ? p.x : hi[0];
            lo[1] = p.y < lo[1] ? p.y : lo[1]; hi[1] = p.y > hi[1] ? p.y : hi[1];
            lo[2] = p.z < lo[2] ? p.z : lo[2]; hi[2] = p.z > hi[2] ? p.z : hi[2];
        }
    }
    for (int q = leaf_off[i]; q < leaf_off[i + 1]; q++) {
        const int k = leaf_slots[q];
        float4* h = tri_hit + 4 * (size_t)k;
        h[1].w = lo[0];
        h[2].w = lo[1];
        h[3] = make_float4(lo[2], hi[0], hi[1], hi[2]);
    }
}

__global__ __launch_bounds__(256) void k_pack_nodes(const PackInfo* pi, int n_rec, const float4* nbox, float4* nodes,
                                                    const int* root_v, const int* root_ref, int n_meshes, float4* roots, int pad, int root_variants) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_rec) {
        PackInfo p = pi[i];
        // An absent child (ref RT_IDX_NONE) gets an all-NaN box and a NaN slack: in the min/max form of the slab test (rays with finite
        // 1/d) every slab value and so the interval is NaN, and "NaN <= x" is false - it can never be entered, whatever the ray,
        // which is what lets k_trace drop the compare of the reference there.  (No finite or infinite box has that property: the
        // min/max form is symmetric in lo and hi, so an inverted box is the whole line.)  An EMPTY leaf that the walk kept as a
        // child (c >= 0, no triangles) has ref RT_IDX_NONE too: same treatment, whatever the refit wrote for it.
        const float qn = __int_as_float(0x7fc00000);
        const float4 zn = make_float4(qn, qn, qn, qn);
        const bool a1 = p.c1 >= 0 && p.ref1 != RT_IDX_NONE, a2 = p.c2 >= 0 && p.ref2 != RT_IDX_NONE;
        float4 lo1 = a1 ? nbox[2 * (size_t)p.c1] : zn, hi1 = a1 ? nbox[2 * (size_t)p.c1 + 1] : zn;
        float4 lo2 = a2 ? nbox[2 * (size_t)p.c2] : zn, hi2 = a2 ? nbox[2 * (size_t)p.c2 + 1] : zn;
        const float s1 = a1 ? box_slack(lo1, hi1, pad) : qn, s2 = a2 ? box_slack(lo2, hi2, pad) : qn;   // (both read their box's .w)
        lo2.w = s1;
        hi2.w = s2;
        lo1.w = __int_as_float(dev_ref(p.ref1));
        hi1.w = __int_as_float(dev_ref(p.ref2));
        float4* o = nodes + 4 * (size_t)i;
        o[0] = lo1; o[1] = hi1; o[2] = lo2; o[3] = hi2;
    }
    if (i < n_meshes) {
        float4 lo = nbox[2 * (size_t)root_v[i]], hi = nbox[2 * (size_t)root_v[i] + 1];
        hi.w = box_slack(lo, hi, pad);
        if (root_ref[i] == RT_IDX_NONE) {         // a mesh without triangles: a box no ray with finite 1/d can enter (see above)
            const float qn = __int_as_float(0x7fc00000);
            lo = hi = make_float4(qn, qn, qn, qn);
        }
        lo.w = __int_as_float(dev_ref(root_ref[i]));
        roots[2 * i] = lo;
        roots[2 * i + 1] = hi;
        // root_variants = 8: the table once per direction octant, variant v at entry v * n_meshes: axis a in (hi, lo) = (near, far)
        // order where bit a of v is set (rt_dev_trace.h::box_test_oct); reference and slack stay where they are
        for (int v = 1; v < root_variants; v++) {
            float4 a = lo, b = hi;
            if (v & 1) { a.x = hi.x; b.x = lo.x; }
            if (v & 2) { a.y = hi.y; b.y = lo.y; }
            if (v & 4) { a.z = hi.z; b.z = lo.z; }
            roots[2 * ((size_t)v * n_meshes + i)] = a;
            roots[2 * ((size_t)v * n_meshes + i) + 1] = b;
        }
    }
}

// std140 read-back helpers
__global__ __launch_bounds__(256) void k_gather_tris28(const float4* geo, const float4* nrm, const int* slot_id, int n_slots, float* out28, int n_tris) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_slots) return;
    float4 g0 = geo[4 * (size_t)k], g1 = geo[4 * (size_t)k + 1], g2 = geo[4 * (size_t)k + 2], g3 = geo[4 * (size_t)k + 3];
    float4 a = nrm[4 * (size_t)k], b = nrm[4 * (size_t)k + 1], c = nrm[4 * (size_t)k + 2];
    int id = slot_id[k];
    (void)g3;
    if (id < 0 || id >= n_tris) return;
    float* o = out28 + 28 * (size_t)id;
    o[0] = g0.x; o[1] = g0.y; o[2] = g0.z; o[4] = g1.x; o[5] = g1.y; o[6] = g1.z; o[8] = g2.x; o[9] = g2.y; o[10] = g2.z;
    o[12] = a.x; o[13] = a.y; o[14] = a.z; o[16] = b.x; o[17] = b.y; o[18] = b.z; o[20] = c.x; o[21] = c.y; o[22] = c.z;
}

// Persistent waves: the grid just fills the chip, and every wave takes jobs (a small tile of pixels x
// all passes of the launch) from a launch-wide counter until none are left (rt_dev_trace.h).  Dynamic
// LDS: mesh root table + argument copy + the per-lane node stack / leaf queue columns.
#ifndef RT_WAVES
#define RT_WAVES 5
#endif
template <bool STATS, int CULL, bool FIRST_HIT, bool SAH, bool DEEP, bool MLDS, bool ROCT>
__global__ __launch_bounds__(RT_BLOCK, RT_WAVES) void k_trace(TraceArgs a) {
    extern __shared__ float4 s_mem[];
    trace_pixels<STATS, CULL, FIRST_HIT, SAH, DEEP, MLDS, ROCT>(a, s_mem);
}

// Scratch of rt_render_static, the entry point that allocates nothing (include/rt_hip.h): the work
// counter, the node stack entries beyond the LDS ones and the waves' staging slots (radiance of the
// samples of the jobs in flight) live in this file-scope array instead of hipMalloc'ed buffers.  Every GPU the library
// runs on holds its own copy until the process ends; its DEVICE address is taken once per context
// with hipGetSymbolAddress (rt_create) - in host code the name `scratch` is only a stand-in.
//   [0, 256)                                 work counter
//   [256, 256 + RT_STATIC_OVF_BYTES)         stack_ovf (int per deep entry and launch lane, a region per workgroup)
//   [256 + RT_STATIC_OVF_BYTES, end)         stage (per wave: 2 slots x job_cap float3)
#define RT_STATIC_OVF_BYTES ((size_t)16 << 20)
#define RT_STATIC_STAGE_OFF ((size_t)256 + RT_STATIC_OVF_BYTES)
static_assert(RT_STATIC_SCRATCH_BYTES > RT_STATIC_STAGE_OFF + (1u << 20), "scratch layout");
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

// ----------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------
namespace {

struct Std140Tri { float posA[3], _a, posB[3], _b, posC[3], _c, nA[3], _d, nB[3], _e, nC[3]; int32_t meshIndex, triId, _f[3]; };
struct Std140Mesh { int32_t index, node0Id, _p[2]; float bboxMin[4], bboxMax[4]; MatDev material; };
struct Std140Node { float aabbMin[3], _p, aabbMax[3]; int32_t id, c1, c2, off, cnt; };
static_assert(sizeof(Std140Tri) == 112 && sizeof(Std140Mesh) == 128 && sizeof(Std140Node) == 48, "std140 sizes");

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        release();
        if (!count) return hipSuccess;
        const hipError_t e = hipMalloc((void**)&p, count * sizeof(T));
        if (e != hipSuccess) { p = nullptr; return e; }   // n stays 0: "capacity < need" checks must not pass with a null buffer
        n = count;
        return hipSuccess;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

}  // namespace

// ---- top-level tree over the mesh roots (scenes with more meshes than the LDS root mask holds) ----
// Its interior nodes are ordinary 64-byte records whose leaf children ARE the mesh roots (same box,
// same reference), so the traversal needs nothing new; their boxes are refit with the rest (levels
// above the mesh levels).  Exactness: a top-level box is the exact union of root boxes, so skipping
// it is the same conservative shortcut as skipping a missed root box - whatever the topology.
// Topology: a binned-SAH tree over the mesh boxes handed in (rest-pose boxes at rt_set_scene, posed
// boxes at every later pose change), one mesh per leaf; a leaf the builder could not split
// (coinciding centres) is halved by a median rule.  Always n_meshes - 1 interior nodes, so a
// rebuild overwrites the same device ranges.
struct TlasHost {
    std::vector<float> box_lo, box_hi;      // rest-pose box of each mesh
    std::vector<VNode> vnodes;              // visit nodes: mesh part [0, n_mesh_visits), then the top-level ones
    std::vector<PackInfo> pack;             // records: mesh part [0, n_mesh_recs), then the top-level ones
    std::vector<int> vdepth_mesh, root_v, root_ref;
    size_t n_mesh_visits = 0, n_mesh_recs = 0;
    int mesh_depth = 0;
};

void build_tlas(int n_meshes, const float* box_lo, const float* box_hi, const TlasHost& th, std::vector<VNode>& vnodes,
                std::vector<int>& vdepth, std::vector<PackInfo>& pack, int& tlas_root, int& max_depth) {
    vnodes.resize(th.n_mesh_visits);
    pack.resize(th.n_mesh_recs);
    vdepth = th.vdepth_mesh;
    std::vector<float> cen(3 * (size_t)n_meshes, 0.f);
    for (size_t k = 0; k < cen.size(); k++) cen[k] = std::isfinite(box_lo[k]) && std::isfinite(box_hi[k]) ? 0.5f * (box_lo[k] + box_hi[k]) : 0.f;
    std::vector<RtNode48> tn;
    std::vector<int32_t> tids;
    int troot;
    {
        std::vector<float> fake(9 * (size_t)n_meshes, 0.f);   // "triangle" (lo, hi, lo): same box, same centre
        std::vector<int32_t> ids(n_meshes);
        for (int j = 0; j < n_meshes; j++) {
            ids[j] = j;
            for (int a = 0; a < 3; a++) {
                const bool ok = std::isfinite(box_lo[3 * j + a]) && std::isfinite(box_hi[3 * j + a]);
                fake[9 * (size_t)j + a] = fake[9 * (size_t)j + 6 + a] = ok ? box_lo[3 * j + a] : 0.f;
                fake[9 * (size_t)j + 3 + a] = ok ? box_hi[3 * j + a] : 0.f;
            }
        }
        // depth 12 is ample for the 256 meshes the interface allows; whatever the builder leaves
        // together at that depth the median rule splits (<= 8 more levels)
        troot = rt_sah_build(fake.data(), ids.data(), n_meshes, 12, 1, tn, tids);
    }
    struct Job { std::vector<int> meshes; int sah_node, depth, parent_rec, which; };
    std::vector<Job> jobs;
    std::vector<std::pair<int, int>> vnode_of_rec;
    int tlas_height = 0;
    tlas_root = -1;
    jobs.push_back({{}, troot, 0, -1, 0});
    while (!jobs.empty()) {
        Job job = std::move(jobs.back());
        jobs.pop_back();
        int v, ref;
        if (job.sah_node >= 0) {
            const RtNode48& t = tn[job.sah_node];
            if (t.c1 < 0 && t.c2 < 0) {                   // a leaf of the SAH tree: its meshes (normally one)
                job.meshes.assign(tids.begin() + t.off, tids.begin() + t.off + t.cnt);
                job.sah_node = -1;
            }
        }
        if (job.sah_node < 0 && job.meshes.size() == 1) { // a mesh root: nothing new, just link it
            v = th.root_v[job.meshes[0]];
            ref = th.root_ref[job.meshes[0]];
        } else {
            v = (int)vnodes.size();
            vnodes.push_back({-1, -1, 0, 0});
            vdepth.push_back(job.depth);
            ref = (int)pack.size();
            pack.push_back({-1, -1, RT_IDX_NONE, RT_IDX_NONE});
            tlas_height = std::max(tlas_height, job.depth + 1);
            vnode_of_rec.push_back({ref, v});
            if (job.sah_node >= 0) {
                const RtNode48& t = tn[job.sah_node];
                jobs.push_back({{}, t.c1, job.depth + 1, ref, 1});
                jobs.push_back({{}, t.c2, job.depth + 1, ref, 2});
            } else {
                float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
                for (int j : job.meshes)
                    for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], cen[3 * j + a]); hi[a] = std::max(hi[a], cen[3 * j + a]); }
                int axis = 0;
                for (int a = 1; a < 3; a++) if (hi[a] - lo[a] > hi[axis] - lo[axis]) axis = a;
                std::sort(job.meshes.begin(), job.meshes.end(), [&](int x, int y) {
                    return cen[3 * x + axis] < cen[3 * y + axis] || (cen[3 * x + axis] == cen[3 * y + axis] && x < y);
                });
                const size_t half = job.meshes.size() / 2;
                jobs.push_back({std::vector<int>(job.meshes.begin(), job.meshes.begin() + half), -1, job.depth + 1, ref, 1});
                jobs.push_back({std::vector<int>(job.meshes.begin() + half, job.meshes.end()), -1, job.depth + 1, ref, 2});
            }
        }
        if (job.parent_rec < 0) tlas_root = ref;
        else if (job.which == 1) { pack[job.parent_rec].c1 = v; pack[job.parent_rec].ref1 = ref; }
        else { pack[job.parent_rec].c2 = v; pack[job.parent_rec].ref2 = ref; }
    }
    for (auto& rv : vnode_of_rec) { vnodes[rv.second].c1 = pack[rv.first].c1; vnodes[rv.second].c2 = pack[rv.first].c2; }
    // mesh levels move below the top-level ones: a parent is always refit after its children
    for (size_t v = 0; v < th.n_mesh_visits; v++) vdepth[v] += tlas_height;
    max_depth = th.mesh_depth + tlas_height;
}

// level lists of the refit (level l covers level_nodes[level_off[l] .. level_off[l+1]))
void level_lists(const std::vector<int>& vdepth, int max_depth, std::vector<int>& level_off, std::vector<int>& level_nodes) {
    level_off.assign(max_depth + 2, 0);
    level_nodes.resize(vdepth.size());
    for (int d : vdepth) level_off[d + 1]++;
    for (size_t l = 1; l < level_off.size(); l++) level_off[l] += level_off[l - 1];
    std::vector<int> fill(level_off.begin(), level_off.end() - 1);
    for (size_t v = 0; v < vdepth.size(); v++) level_nodes[fill[vdepth[v]]++] = (int)v;
}

#define RT_JOB_SAMPLES 512u          // samples of a job: tail of a launch ~ half a job per wave, fold at (tile pixels x 3) lanes
#define RT_JOB_SAMPLES_STATIC 256u   // rt_render_static: 6144 waves x 2 slots x 256 x 12 B fit its 48 MiB
#define RT_MAX_LAUNCH_PASSES 1024u   // a job holds all passes of its launch

struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;       // stream in use
    hipStream_t own_stream = nullptr;   // created by rt_create
    char err[512] = {0};
    // scene (device)
    DevBuf<float4> rest, geo, nrm, nodes, roots, nbox;   // nrm = the 64-byte hit records (normals, mesh, reference leaf box)
    DevBuf<int> ref_leaf_off, ref_leaf_slots;   // RT_ACCEL_SAH: the reference's leaves as lists of our slots
    int n_ref_leaves = 0;
    DevBuf<int> slot_mesh, slot_rem, slot_id, slot_rank, rank_slot, level_nodes, root_v, root_ref;
    DevBuf<float> xf;
    DevBuf<unsigned char> dirty;
    DevBuf<VNode> vnodes;
    DevBuf<PackInfo> pack;
    DevBuf<MatDev> mats;
    DevBuf<SphereDev> spheres;
    DevBuf<unsigned long long> counters;
    DevBuf<unsigned int> work_counter;
    DevBuf<float> stage;                  // staging slots of the launch's waves: 2 x job_cap float3 each (grow-only)
    int n_cus = 0;
    size_t lds_per_cu = 160 * 1024;
    int stack_lds[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};       // per k_trace variant: node stack entries in LDS for the scene in place
    DevBuf<int> stack_ovf;                // deeper entries (rare): per workgroup (stack_depth - cap) x RT_BLOCK
    size_t stack_ovf_n = 0;
    unsigned char* static_scratch = nullptr;   // device address of `scratch` on this context's GPU (hipGetSymbolAddress)
    bool launch_static = false;           // the launch in preparation belongs to rt_render_static: nothing may be allocated
    DevBuf<int32_t> fh_ids;               // rt_first_hit: device-side results (grow-only)
    DevBuf<float> fh_ts;
    DevBuf<float> tris_tmp;               // rt_read_tris28: gathered std140 image (grow-only)
    MatDev* h_mats_pinned = nullptr;      // rt_set_materials: pinned staging of the 256 materials, guarded by mats_copied
    hipEvent_t mats_copied = nullptr;
    uint32_t max_cycles = 0xffffffffu;    // passes with a larger index leave the image as it is (raytracer.fs.glsl:931-935)
    int last_root_variants = 0;
    int last_stack_lds = 0, last_lds_bytes = 0, last_blocks_per_cu = 0;   // layout of the most recent k_trace launch (rt_get_info)
    int tlas_rebuilds = 0;
    int stack_lds_force = 0;              // RT_STACK_LDS: LDS node stack entries, whatever the occupancy (tests of the deep-entry path)
    int blocks_per_cu[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // per k_trace variant, for the scene in place (0 = not yet asked)
    // scene (host mirrors)
    std::vector<float> h_xf;
    std::vector<unsigned char> h_dirty;
    TlasHost th;                          // host copy of what a rebuild of the top-level tree needs
    bool tlas_pose_dirty = false;         // a pose changed since the top-level tree was built
    bool tlas_static = false;             // RT_TLAS_STATIC: keep the rest-pose topology (measurements)
    std::vector<int> h_level_nodes;       // staging of a rebuild (kept alive for the async copy)
    std::vector<char> h_has_xf;           // per mesh: a transform has been set (else the mesh is at its rest pose)
    std::vector<int> level_off;           // level l covers level_nodes[level_off[l] .. level_off[l+1])
    std::vector<int> node_first_visit;    // original node id -> visit index (-1 if unreachable)
    std::vector<Std140Node> h_nodes;      // topology as uploaded (for rt_read_nodes48)
    MatDev atmo{};
    int n_tris = 0, n_slots = 0, n_meshes = 0, n_nodes = 0, n_visit = 0, n_rec = 0, n_spheres = 0;
    int n_roots = 0;                      // entries of the root table: one per mesh, or per group of meshes (SAH structure, sah_group_below)
    bool root_oct = false;                // the root table of the scene in place exists in its eight octant variants (ROCT instantiations of k_trace)
    bool root_oct_allowed = true;         // RT_ROOT_OCT=0: never
    int treelet = 8;                      // RT_TREELET: interior records laid out in breadth-first blocks of this many (0 = depth-first order)
    bool sah_tip_pad = true;              // RT_SAH_TIP_PAD=0: the SAH boxes' slack without the sliver-tip pad (tests: what the pad is for)
    int sah_group_below = 0;              // RT_SAH_GROUP_BELOW: meshes with fewer triangles share one SAH tree (0 = one tree per mesh)
    int stack_depth = 1;
    int tlas_above = 32;                  // scenes with more meshes than this get the top-level tree (the LDS root mask holds 32)
    int accel = RT_ACCEL_REFERENCE;       // structure built by the next rt_set_scene (rt_set_accel)
    bool scene_is_sah = false;
    int sah_leaf = 3;                     // triangles per SAH leaf (RT_SAH_LEAF); measured best with the 3-slot leaf queue
    int tlas_root = -1;                   // record of the top-level tree's root (scenes with > 32 meshes), else -1
    bool have_scene = false, have_cam = false, prep_pending = false;
    bool flags_pending = false;           // materials changed since the triangle flag words were written
    int cull = RT_CULL_OFF;               // distance culling (rt_set_cull): RT_CULL_OFF / RT_CULL_ON / RT_CULL_BAND
    int node_bias = 0;
    int burst8 = -1;                      // RT_BURST8; -1 = measured default (4 of 8 lanes, SAH scenes: 3 of 8)
    uint32_t job_samples = RT_JOB_SAMPLES;   // RT_JOB_SAMPLES: samples of a job (measurements)
    int recheck_all = 0;                  // RT_SAH_RECHECK_ALL: run the SAH structure's repeat path on every query (tests)
    int refill_below = 0;                 // lanes still traversing below which a wave shades/refills; 0 = measured default
    // camera / image
    DevCamera cam{};
    DevBuf<float> accum_own;
    float* accum = nullptr;
    uint64_t n_accum = 0;
    bool sum_mode = false;
    // stats
    bool stats_on = false;
    rt_stats_t st{};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending_events;
};

namespace {

int fail(rt_ctx* c, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(c->err, sizeof(c->err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCK(c, call)                                                                          \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) return fail((c), RT_E_DEVICE, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
int upload(rt_ctx* c, DevBuf<T>& b, const std::vector<T>& v) {
    HIPCK(c, b.alloc(v.size()));
    if (!v.empty()) HIPCK(c, hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    return RT_OK;
}

inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// numba dtype rules of mat33_create_from_quaternion (scripts/numba_utils/functions.py:35-82)
void pose_matrix(const float pos[3], const float q_in[4], float T[16]) {
    float q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
    float nn = std::sqrt(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    if (!(std::fabs((double)nn - 1.0) <= 1e-8 + 1e-5))
        for (int i = 0; i < 4; i++) q[i] = q[i] / nn;
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z, ww = w * w;
    float xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    double inv = 1.0 / (double)(((xx + yy) + zz) + ww);
    std::memset(T, 0, 64);
    T[0] = (float)((double)(((xx - yy) - zz) + ww) * inv);
    T[5] = (float)((double)(((-xx + yy) - zz) + ww) * inv);
    T[10] = (float)((double)(((-xx - yy) + zz) + ww) * inv);
    T[4] = (float)(2.0 * (double)(xy + zw) * inv);
    T[1] = (float)(2.0 * (double)(xy - zw) * inv);
    T[8] = (float)(2.0 * (double)(xz - yw) * inv);
    T[2] = (float)(2.0 * (double)(xz + yw) * inv);
    T[9] = (float)(2.0 * (double)(yz + xw) * inv);
    T[6] = (float)(2.0 * (double)(yz - xw) * inv);
    T[12] = pos[0]; T[13] = pos[1]; T[14] = pos[2]; T[15] = 1.0f;
}

// pose dirty meshes, refit every level bottom-up, re-pack the traversal records
// Poses changed in a scene with a top-level tree: its topology is rebuilt from the posed mesh
// boxes (the 8 corners of each rest-pose box through the mesh's transform: conservative, and only
// the topology depends on them - the boxes themselves come from the device refit).  ~0.1 ms of
// host time for 192 meshes; same record and visit-node ranges on the device, new level lists.
int rebuild_tlas(rt_ctx* c) {
    TlasHost& th = c->th;
    const int n = c->n_meshes;
    std::vector<float> lo(th.box_lo), hi(th.box_hi);
    for (int j = 0; j < n; j++) {
        if (!c->h_has_xf[j]) continue;
        const float* T = &c->h_xf[16 * (size_t)j];
        const float* rl = &th.box_lo[3 * (size_t)j];
        const float* rh = &th.box_hi[3 * (size_t)j];
        if (!(std::isfinite(rl[0]) && std::isfinite(rh[0]))) continue;       // a mesh without triangles
        float plo[3] = {INFINITY, INFINITY, INFINITY}, phi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 8; k++) {
            const float p[3] = {(k & 1) ? rh[0] : rl[0], (k & 2) ? rh[1] : rl[1], (k & 4) ? rh[2] : rl[2]};
            for (int a = 0; a < 3; a++) {
                const float q = p[0] * T[0 + a] + p[1] * T[4 + a] + p[2] * T[8 + a] + T[12 + a];
                plo[a] = std::min(plo[a], q);
                phi[a] = std::max(phi[a], q);
            }
        }
        for (int a = 0; a < 3; a++) { lo[3 * (size_t)j + a] = plo[a]; hi[3 * (size_t)j + a] = phi[a]; }
    }
    std::vector<int> vdepth;
    int tlas_root = -1, max_depth = 0;
    build_tlas(n, lo.data(), hi.data(), th, th.vnodes, vdepth, th.pack, tlas_root, max_depth);
    if (th.vnodes.size() != (size_t)c->n_visit || th.pack.size() != (size_t)c->n_rec)
        return fail(c, RT_E_STATE, "top-level tree rebuild changed its size");
    level_lists(vdepth, max_depth, c->level_off, c->h_level_nodes);
    const size_t nv = th.vnodes.size() - th.n_mesh_visits, nr = th.pack.size() - th.n_mesh_recs;
    HIPCK(c, hipMemcpyAsync(c->vnodes.p + th.n_mesh_visits, th.vnodes.data() + th.n_mesh_visits, nv * sizeof(VNode), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->pack.p + th.n_mesh_recs, th.pack.data() + th.n_mesh_recs, nr * sizeof(PackInfo), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->level_nodes.p, c->h_level_nodes.data(), c->h_level_nodes.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
    c->tlas_root = tlas_root;
    c->tlas_rebuilds++;
    if (c->stack_depth != max_depth + 1) {
        c->stack_depth = max_depth + 1;
        for (int k = 0; k < 16; k++) c->blocks_per_cu[k] = 0;               // LDS layout and occupancy follow the depth
    }
    return RT_OK;
}

int run_prep(rt_ctx* c) {
    if (!c->prep_pending) {
        // materials alone changed (rt_set_materials): only the triangles' flag words follow them -
        // no pose upload, no host synchronisation, no refit
        if (c->flags_pending && c->n_slots) {
            hipLaunchKernelGGL(k_tri_flags, dim3(blocks_for(c->n_slots)), dim3(256), 0, c->stream, c->slot_mesh.p, c->mats.p, c->geo.p, c->n_slots);
            HIPCK(c, hipGetLastError());
        }
        c->flags_pending = false;
        return RT_OK;
    }
    if (c->tlas_root >= 0 && c->tlas_pose_dirty && !c->tlas_static) {
        int rc = rebuild_tlas(c);                  // its copies complete at the synchronisation below
        if (rc) return rc;
    }
    c->tlas_pose_dirty = false;
    HIPCK(c, hipMemcpyAsync(c->xf.p, c->h_xf.data(), c->h_xf.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipMemcpyAsync(c->dirty.p, c->h_dirty.data(), c->h_dirty.size(), hipMemcpyHostToDevice, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));   // pageable sources: h_dirty is cleared below
    if (c->n_slots) {
        PoseArgs pa{c->rest.p, c->slot_mesh.p, c->slot_rem.p, c->slot_rank.p, c->xf.p, c->dirty.p, c->mats.p, c->geo.p, c->nrm.p, c->n_slots};
        hipLaunchKernelGGL(k_pose, dim3(blocks_for(c->n_slots)), dim3(256), 0, c->stream, pa);
    }
    if (c->flags_pending && c->n_slots)
        hipLaunchKernelGGL(k_tri_flags, dim3(blocks_for(c->n_slots)), dim3(256), 0, c->stream, c->slot_mesh.p, c->mats.p, c->geo.p, c->n_slots);
    c->flags_pending = false;
    if (c->scene_is_sah && c->n_ref_leaves)
        hipLaunchKernelGGL(k_ref_leaf_boxes, dim3(blocks_for(c->n_ref_leaves)), dim3(256), 0, c->stream, c->ref_leaf_off.p,
                           c->ref_leaf_slots.p, c->n_ref_leaves, c->geo.p, c->nrm.p);
    for (int l = (int)c->level_off.size() - 2; l >= 0; l--) {
        int n = c->level_off[l + 1] - c->level_off[l];
        if (n <= 0) continue;
        hipLaunchKernelGGL(k_refit_level, dim3(blocks_for(n)), dim3(256), 0, c->stream, c->vnodes.p,
                           c->level_nodes.p + c->level_off[l], n, c->geo.p, c->nbox.p, c->scene_is_sah && c->sah_tip_pad ? 1 : 0);
    }
    size_t m = (size_t)std::max(c->n_rec, c->n_roots);
    if (m)
        hipLaunchKernelGGL(k_pack_nodes, dim3(blocks_for(m)), dim3(256), 0, c->stream, c->pack.p, c->n_rec, c->nbox.p,
                           c->nodes.p, c->root_v.p, c->root_ref.p, c->n_roots, c->roots.p, c->scene_is_sah ? 1 : 0, c->root_oct ? 8 : 1);
    HIPCK(c, hipGetLastError());
    for (auto& d : c->h_dirty) d = 0;
    c->prep_pending = false;
    return RT_OK;
}

int rank_lds_entries(const rt_ctx* c);
DevScene scene_view(const rt_ctx* c) {
    DevScene s;
    s.tri_geo = c->geo.p; s.tri_hit = c->nrm.p; s.nodes = c->nodes.p; s.roots = c->roots.p;
    s.mats = c->mats.p; s.spheres = c->spheres.p; s.rank_slot = c->rank_slot.p; s.slot_id = c->slot_id.p; s.n_meshes = c->n_meshes; s.n_roots = c->n_roots; s.n_spheres = c->n_spheres;
    s.stack_depth = c->stack_depth;
    s.tlas_root = c->tlas_root >= 0 ? dev_ref(c->tlas_root) : -1;
    s.rank_lds = rank_lds_entries(c);
    return s;
}

// A job = a tile of (1 << lx) x (1 << ly) pixels with all `n_passes` passes of the launch: the largest power-of-two
// tile that keeps the job at `target` samples or fewer (one pixel if the passes alone exceed it), as square as possible.
void set_job_shape(const rt_ctx* c, TraceArgs& a, uint32_t n_passes, uint32_t target) {
    uint32_t l = 0;
    while (l < 10 && ((uint64_t)2 << l) * n_passes <= target) l++;
    a.tile_lx = (l + 1) / 2;
    a.tile_ly = l / 2;
    a.n_tx = ((uint32_t)c->cam.width + (1u << a.tile_lx) - 1) >> a.tile_lx;
    a.n_ty = ((uint32_t)c->cam.height + (1u << a.tile_ly) - 1) >> a.tile_ly;
    a.n_jobs = a.n_tx * a.n_ty;
    a.job_cap = n_passes << l;
}
void fill_args(const rt_ctx* c, TraceArgs& a) {
    a.scene = scene_view(c);
    a.cam = c->cam;
    a.atmo = c->atmo;
    a.accum = c->accum;
    a.ids = nullptr;
    a.ts = nullptr;
    a.counters = c->counters.p;
    a.stage = nullptr;
    a.work_counter = c->work_counter.p;
    a.n_tx = a.n_ty = a.n_jobs = 0;           // set_job_shape
    a.tile_lx = a.tile_ly = 0;
    a.job_cap = 0;
    a.node_bias = c->node_bias;
    // (SAH scenes: 3 of 8 with a top-level tree or beyond the L2, 4 of 8 for the root-table scenes since the vote got cheaper -
    // re-swept on the final kernel of round 3: 39.40 -> 39.30 ms on the bench frame, the other two flat)
    a.burst8 = c->burst8 >= 0 ? c->burst8 : (c->scene_is_sah && (c->tlas_root >= 0 || c->n_slots > 100000) ? 3 : 4);
    a.recheck_all = c->recheck_all;
    // best measured (tools/sweep_sched.sh): a scene that lives in the L2 gains from shading fuller path blocks,
    // one that does not (991 k triangles) from refilling earlier (28 since round 3: tools/sweep_inproc.py); a scene with a top-level tree (long traversals) shades fuller still;
    // the root-table scenes went from 20 to 22 when round 3 made the path block cheaper (profiles/r03_sched_sweep.txt)
    a.refill_below = c->refill_below > 0 ? c->refill_below : (!c->cull ? 32 : !c->scene_is_sah ? 28 : c->n_slots > 100000 ? 28 : c->tlas_root >= 0 ? 16 : 22);
}

// dynamic LDS of a k_trace launch: root table + kernel-argument copy + per-lane node stack and leaf
// queue (+ a 16-bit entry-distance plane for the leaf queue when culling).  The node stack of a deep tree (dragon scene:
// 29 levels with the top-level tree = 48 KB per workgroup) would leave 3 workgroups per CU, so
// only as many entries as keep RT_WAVES workgroups resident live in LDS; the rare deeper ones go
// to `stack_ovf` in global memory (DEEP instantiations; a scene whose whole stack fits runs the
// instantiation without that test).
// rank -> slot entries k_trace keeps in LDS: the whole table of a small SAH scene with a root table (the reference
// structure needs no table: its slots are in rank order)
int rank_lds_entries(const rt_ctx* c) {
    return (c->scene_is_sah && c->tlas_root < 0 && c->n_slots <= 2048) ? c->n_slots : 0;
}

size_t trace_lds_bytes(const rt_ctx* c, int cap, bool cull, bool sah, bool deep, bool roct = false) {
    // (roct: the root table eight times, the rank table as 16-bit entries - the ROCT instantiations' layout, rt_dev_trace.h)
    const size_t fixed = ((size_t)(c->tlas_root >= 0 ? 0 : 2 * c->n_roots * (roct ? 8 : 1) + 5 * c->n_meshes) +
                          (size_t)(roct ? (rank_lds_entries(c) + 7) / 8 : (rank_lds_entries(c) + 3) / 4) + RT_PC_F4) * sizeof(float4);   // roots, materials, rank table, arguments
    const size_t lq = RT_LQ_OF(sah);
    if (RT_LQ64_OF(sah, cull, deep)) return fixed + ((size_t)cap * 4 + lq * 8) * RT_BLOCK;      // 8-byte leaf queue entries (rt_dev_trace.h)
    return fixed + ((size_t)cap + lq) * RT_BLOCK * 4 + (cull ? lq * RT_BLOCK * 2 : 0);
}

template <bool STATS, int CULL, bool FIRST, bool SAH, bool DEEP, bool MLDS, bool ROCT = false>
int launch_one(rt_ctx* c, const TraceArgs& a, int key, int cap) {
    if (c->blocks_per_cu[key] <= 0) {
        int nb = 0, nb_best = 0;
        for (;;) {
            HIPCK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_trace<STATS, CULL, FIRST, SAH, DEEP, MLDS, ROCT>, RT_BLOCK, trace_lds_bytes(c, cap, CULL, SAH, DEEP, ROCT)));
            if (!DEEP || nb >= RT_WAVES || cap <= RT_MIN_STACK_LDS || c->stack_lds_force > 0) break;
            // lds_stack_entries guessed the allocation granule wrong, or registers are the limit
            if (nb_best == 0)
                HIPCK(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb_best, k_trace<STATS, CULL, FIRST, SAH, DEEP, MLDS, ROCT>, RT_BLOCK, trace_lds_bytes(c, RT_MIN_STACK_LDS, CULL, SAH, DEEP, ROCT)));
            if (nb >= nb_best) break;
            cap--;
        }
        if (nb < 1) return fail(c, RT_E_LIMIT, "scene needs %zu bytes of LDS per workgroup (%d roots, %d meshes)", trace_lds_bytes(c, cap, CULL, SAH, DEEP, ROCT), c->n_roots, c->n_meshes);
        c->blocks_per_cu[key] = nb;
        c->stack_lds[key] = cap;
        if (getenv("RT_DEBUG")) fprintf(stderr, "[rt] k_trace variant %d: %d of %d node stack entries in LDS, %zu bytes of LDS, %d workgroups per CU\n", key, cap, c->stack_depth, trace_lds_bytes(c, cap, CULL, SAH, DEEP, ROCT), nb);
    }
    cap = c->stack_lds[key];
    const size_t lds = trace_lds_bytes(c, cap, CULL, SAH, DEEP, ROCT);
    // every wave takes whole chunks: no point in more waves than chunks
    unsigned grid = std::max(1u, (unsigned)std::min<uint64_t>((uint64_t)c->blocks_per_cu[key] * c->n_cus, ((uint64_t)a.n_jobs + 3) / 4));
    TraceArgs b = a;
    b.scene.stack_depth = cap;
    const size_t deep = (size_t)std::max(0, c->stack_depth - cap);
    if (c->launch_static) {
        // rt_render_static: the deep entries live in the static scratch; a scene whose deep entries
        // do not fit for a full grid runs on as many workgroups as they fit for
        const size_t cap_ent = RT_STATIC_OVF_BYTES / sizeof(int);
        if (deep * RT_BLOCK > cap_ent) return fail(c, RT_E_LIMIT, "rt_render_static: BVH too deep for the static scratch (%zu entries per lane beyond LDS)", deep);
        if (deep) grid = (unsigned)std::min<size_t>(grid, cap_ent / (deep * RT_BLOCK));
        // ... and the staging slots of its waves: 2 x job_cap float3 each
        const size_t per_block = (size_t)(RT_BLOCK / 64) * 2 * a.job_cap * 3 * sizeof(float);
        if (!FIRST) {
            if (per_block > RT_STATIC_SCRATCH_BYTES - RT_STATIC_STAGE_OFF) return fail(c, RT_E_LIMIT, "rt_render_static: a job of %u samples does not fit the static scratch", a.job_cap);
            grid = (unsigned)std::min<size_t>(grid, (RT_STATIC_SCRATCH_BYTES - RT_STATIC_STAGE_OFF) / per_block);
        }
        b.stack_ovf = (int*)(c->static_scratch + 256);
        b.work_counter = (unsigned int*)c->static_scratch;
        b.stage = (float*)(c->static_scratch + RT_STATIC_STAGE_OFF);
    } else {
        const size_t need_stage = FIRST ? 0 : (size_t)grid * (RT_BLOCK / 64) * 2 * a.job_cap * 3;
        if (c->stage.n < need_stage) {
            HIPCK(c, hipStreamSynchronize(c->stream));
            if (c->stage.alloc(need_stage) != hipSuccess) return fail(c, RT_E_NOMEM, "no device memory for the staging slots of %u workgroups", grid);
        }
        b.stage = c->stage.p;
        const size_t need_ovf = deep * (size_t)grid * RT_BLOCK;
        if (c->stack_ovf_n < need_ovf) {
            HIPCK(c, hipStreamSynchronize(c->stream));
            c->stack_ovf.release();
            c->stack_ovf_n = 0;
            HIPCK(c, c->stack_ovf.alloc(need_ovf));
            c->stack_ovf_n = need_ovf;
        }
        b.stack_ovf = c->stack_ovf.p;
    }
    b.ovf_block_bytes = (uint32_t)(deep * RT_BLOCK * sizeof(int));
    c->last_stack_lds = cap;
    c->last_lds_bytes = (int)lds;
    c->last_blocks_per_cu = c->blocks_per_cu[key];
    c->last_root_variants = c->tlas_root >= 0 ? 0 : ROCT ? 8 : 1;
    HIPCK(c, hipMemsetAsync(b.work_counter, 0, sizeof(unsigned int), c->stream));
    hipLaunchKernelGGL((k_trace<STATS, CULL, FIRST, SAH, DEEP, MLDS, ROCT>), dim3(grid), dim3(RT_BLOCK), lds, c->stream, b);
    return RT_OK;
}

// node stack entries kept in LDS: all of them if RT_WAVES workgroups per CU still fit, else as many as do
// (*deep_layout: the DEEP instantiation is to be used even if every entry fits - its LDS layout is the smaller one, see RT_LQ64_OF)
int lds_stack_entries(const rt_ctx* c, bool* deep_layout) {
    *deep_layout = false;
    if (c->stack_lds_force > 0) return std::min(c->stack_depth, c->stack_lds_force);
    // LDS is handed out in 1280-byte granules on gfx950 (160 KB / 128): budget per workgroup rounded down to that
    const size_t budget = (c->lds_per_cu / RT_WAVES) / 1280 * 1280;
    if (trace_lds_bytes(c, c->stack_depth, c->cull, c->scene_is_sah, false) <= budget) return c->stack_depth;   // whole stack in LDS
    const size_t base = trace_lds_bytes(c, 0, c->cull, c->scene_is_sah, true);
    const int fit = (int)((budget > base ? budget - base : 0) / ((size_t)RT_BLOCK * 4));
    *deep_layout = true;
    return std::max(RT_MIN_STACK_LDS, std::min(c->stack_depth, fit));
}

template <bool STATS, int CULL, bool FIRST, bool SAH, bool DEEP>
int launch_mats(rt_ctx* c, const TraceArgs& a, int key, int cap) {
    // MLDS: the materials of every possible hit are in k_trace's LDS copy (a scene with a root table, i.e. without a top-level
    // tree, and without spheres) - proved here on the host, per launch.  Only the render kernels have the instantiation.
    if constexpr (!STATS && !FIRST) {
        if (c->tlas_root < 0 && c->n_spheres == 0 && c->n_meshes > 0) {
            // ROCT: the root table in its eight octant variants (rt_set_scene wrote them: root_oct) where that layout fits the LDS
            // of a full-occupancy launch with the whole node stack, like the plain one does (!DEEP)
            if constexpr (SAH && !DEEP) {
                if (c->root_oct && cap == c->stack_depth && trace_lds_bytes(c, cap, CULL, SAH, DEEP, true) <= (c->lds_per_cu / RT_WAVES) / 1280 * 1280)
                    return launch_one<STATS, CULL, FIRST, SAH, DEEP, true, true>(c, a, key, cap);
            }
            return launch_one<STATS, CULL, FIRST, SAH, DEEP, true>(c, a, key, cap);
        }
    }
    return launch_one<STATS, CULL, FIRST, SAH, DEEP, false>(c, a, key, cap);
}

template <bool STATS, int CULL, bool FIRST>
int launch_accel(rt_ctx* c, const TraceArgs& a, int key, int cap, bool deep_layout) {
    const bool deep = cap < c->stack_depth || deep_layout;
    if (c->scene_is_sah) return deep ? launch_mats<STATS, CULL, FIRST, true, true>(c, a, key, cap) : launch_mats<STATS, CULL, FIRST, true, false>(c, a, key, cap);
    return deep ? launch_mats<STATS, CULL, FIRST, false, true>(c, a, key, cap) : launch_mats<STATS, CULL, FIRST, false, false>(c, a, key, cap);
}

int launch_trace(rt_ctx* c, const TraceArgs& a, bool first_hit) {
    // the instantiation is fixed by (stats, cull mode, first hit) and, per scene, by (structure, deep):
    // 12 cache slots (of 16) for the occupancy of the scene in place
    const int key = (c->stats_on ? 8 : 0) | (c->cull << 1) | (first_hit ? 1 : 0);
    bool dl = false;
    const int cap = lds_stack_entries(c, &dl);
    switch (key) {
    case 0: return launch_accel<false, 0, false>(c, a, key, cap, dl);
    case 1: return launch_accel<false, 0, true>(c, a, key, cap, dl);
    case 2: return launch_accel<false, 1, false>(c, a, key, cap, dl);
    case 3: return launch_accel<false, 1, true>(c, a, key, cap, dl);
    case 4: return launch_accel<false, 2, false>(c, a, key, cap, dl);
    case 5: return launch_accel<false, 2, true>(c, a, key, cap, dl);
    case 8: return launch_accel<true, 0, false>(c, a, key, cap, dl);
    case 9: return launch_accel<true, 0, true>(c, a, key, cap, dl);
    case 10: return launch_accel<true, 1, false>(c, a, key, cap, dl);
    case 11: return launch_accel<true, 1, true>(c, a, key, cap, dl);
    case 12: return launch_accel<true, 2, false>(c, a, key, cap, dl);
    case 13: return launch_accel<true, 2, true>(c, a, key, cap, dl);
    default: return fail(c, RT_E_STATE, "no k_trace instantiation for key %d", key);
    }
}

int check_ready(rt_ctx* c) {
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    if (!c->have_cam) return fail(c, RT_E_STATE, "no camera");
    if (!c->accum || c->cam.width <= 0 || c->cam.height <= 0) return fail(c, RT_E_STATE, "no image size");
    return RT_OK;
}

// The pair of events around one trace launch.  Destroyed on every path that does not hand them to
// note_launch (an error between creation and the hand-over must not leak them).
struct LaunchTimer {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool handed = false;
    hipError_t create() {
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        return e;
    }
    ~LaunchTimer() {
        if (handed) return;
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
};

// Fold the event pairs whose launches have already finished into kernel_ms without blocking, so
// that a long render loop which never asks for statistics does not pile up events.
void note_launch(rt_ctx* c, hipEvent_t e0, hipEvent_t e1) {
    c->pending_events.push_back({e0, e1});
    if (c->pending_events.size() < 64) return;
    size_t done = 0;
    while (done < c->pending_events.size() && hipEventQuery(c->pending_events[done].second) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->pending_events[done].first, c->pending_events[done].second) == hipSuccess) c->st.kernel_ms += ms;
        (void)hipEventDestroy(c->pending_events[done].first);
        (void)hipEventDestroy(c->pending_events[done].second);
        done++;
    }
    c->pending_events.erase(c->pending_events.begin(), c->pending_events.begin() + done);
}

int resolve_events(rt_ctx* c) {
    if (c->pending_events.empty()) return RT_OK;
    HIPCK(c, hipStreamSynchronize(c->stream));
    for (auto& ev : c->pending_events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) c->st.kernel_ms += ms;
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    c->pending_events.clear();
    return RT_OK;
}

}  // namespace

extern "C" {

const char* rt_backend_name(void) { return "hip"; }
const char* rt_last_error(const rt_ctx* c) { return c ? c->err : "null context"; }

int rt_create(int device, rt_ctx** out) {
    if (!out) return RT_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return RT_E_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return RT_E_DEVICE;
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return RT_E_NOMEM;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) { delete c; return RT_E_DEVICE; }
    c->stream = c->own_stream;
    if (const char* e = getenv("RT_SAH_LEAF")) c->sah_leaf = std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("RT_SAH_GROUP_BELOW")) c->sah_group_below = std::max(0, atoi(e));
    if (const char* e = getenv("RT_SAH_TIP_PAD")) c->sah_tip_pad = atoi(e) != 0;
    if (const char* e = getenv("RT_TREELET")) c->treelet = std::max(0, atoi(e));
    if (const char* e = getenv("RT_ROOT_OCT")) c->root_oct_allowed = atoi(e) != 0;
    if (const char* e = getenv("RT_TLAS_STATIC")) c->tlas_static = atoi(e) != 0;
    if (const char* e = getenv("RT_TLAS_ABOVE")) c->tlas_above = std::max(1, std::min(32, atoi(e)));
    if (const char* e = getenv("RT_NODE_BIAS")) c->node_bias = atoi(e);
    if (const char* e = getenv("RT_BURST8")) c->burst8 = std::max(0, std::min(64, atoi(e)));
    if (const char* e = getenv("RT_SAH_RECHECK_ALL")) c->recheck_all = atoi(e) != 0;
    if (const char* e = getenv("RT_STACK_LDS")) c->stack_lds_force = std::max(0, std::min(64, atoi(e)));
    if (const char* e = getenv("RT_JOB_SAMPLES")) c->job_samples = (uint32_t)std::max(16, std::min(1024, atoi(e)));
    if (const char* e = getenv("RT_REFILL_BELOW")) c->refill_below = std::max(1, std::min(64, atoi(e)));
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
        c->n_cus = prop.multiProcessorCount;
        if (prop.maxSharedMemoryPerMultiProcessor > 0) c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    }
    if (c->n_cus <= 0) c->n_cus = 256;
    // this GPU's copy of the static scratch (each device holds its own, at its own address)
    void* sp = nullptr;
    if (c->work_counter.alloc(1) != hipSuccess || c->counters.alloc(16) != hipSuccess ||
        hipMemsetAsync(c->counters.p, 0, 128, c->stream) != hipSuccess ||
        hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp ||
        hipHostMalloc((void**)&c->h_mats_pinned, 256 * sizeof(MatDev), hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&c->mats_copied, hipEventDisableTiming) != hipSuccess) {
        if (c->h_mats_pinned) (void)hipHostFree(c->h_mats_pinned);
        c->work_counter.release();
        c->counters.release();
        (void)hipStreamDestroy(c->own_stream);
        delete c;
        return RT_E_DEVICE;
    }
    c->static_scratch = (unsigned char*)sp;
    *out = c;
    return RT_OK;
}

static void free_scene(rt_ctx* c) {
    c->rest.release(); c->geo.release(); c->nrm.release(); c->nodes.release(); c->roots.release(); c->nbox.release();
    c->slot_mesh.release(); c->slot_rem.release(); c->slot_id.release(); c->slot_rank.release(); c->rank_slot.release(); c->ref_leaf_off.release(); c->ref_leaf_slots.release(); c->level_nodes.release();
    c->root_v.release(); c->root_ref.release(); c->xf.release(); c->dirty.release(); c->vnodes.release();
    c->pack.release(); c->mats.release(); c->spheres.release();
    c->have_scene = false;
}

int rt_destroy(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& ev : c->pending_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    free_scene(c);
    c->accum_own.release();
    c->stage.release();
    c->work_counter.release();
    c->stack_ovf.release();
    c->counters.release();
    c->fh_ids.release();
    c->fh_ts.release();
    c->tris_tmp.release();
    if (c->h_mats_pinned) (void)hipHostFree(c->h_mats_pinned);
    if (c->mats_copied) (void)hipEventDestroy(c->mats_copied);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return RT_OK;
}

int rt_set_scene(rt_ctx* c, const float* tris28, int n_tris, const void* meshes128, int n_meshes, const void* nodes48,
                 int n_nodes, const int32_t* leaf_ids, int n_ids, const void* spheres96, int n_spheres,
                 const void* atmosphere80) {
    if (!c) return RT_E_ARG;
    if (n_tris < 0 || n_meshes < 0 || n_nodes < 0 || n_ids < 0 || n_spheres < 0) return fail(c, RT_E_ARG, "negative count");
    if ((n_tris && !tris28) || (n_meshes && !meshes128) || (n_nodes && !nodes48) || (n_ids && !leaf_ids) ||
        (n_spheres && !spheres96) || !atmosphere80)
        return fail(c, RT_E_ARG, "null buffer with non-zero count");
    if (n_meshes > 256 || n_spheres > 256) return fail(c, RT_E_LIMIT, "more than 256 meshes/spheres");
    if (n_tris > (1 << 25) || n_nodes > (1 << 25)) return fail(c, RT_E_LIMIT, "more than 2^25 triangles or nodes (32-bit record offsets)");
    HIPCK(c, hipSetDevice(c->device));
    const Std140Tri* tris = (const Std140Tri*)tris28;
    const Std140Mesh* meshes = (const Std140Mesh*)meshes128;
    const Std140Node* nodes = (const Std140Node*)nodes48;

    // ---- validation (same rules as the oracle) ----
    for (int i = 0; i < n_ids; i++)
        if (leaf_ids[i] < 0 || leaf_ids[i] >= n_tris) return fail(c, RT_E_ARG, "leaf_tri_ids[%d] out of range", i);
    for (int i = 0; i < n_tris; i++) {
        if (tris[i].meshIndex < 0 || tris[i].meshIndex >= n_meshes) return fail(c, RT_E_ARG, "triangle %d: bad mesh index", i);
        if (tris[i].triId != i) return fail(c, RT_E_ARG, "triangle %d: triId != index (scenes/utils.py:18-25)", i);
    }
    for (int i = 0; i < n_nodes; i++) {
        const Std140Node& n = nodes[i];
        if (n.c1 >= n_nodes || n.c2 >= n_nodes) return fail(c, RT_E_ARG, "node %d: child out of range", i);
        if ((n.c1 >= 0 && n.c1 <= i) || (n.c2 >= 0 && n.c2 <= i)) return fail(c, RT_E_ARG, "node %d: child id not greater than parent id", i);
        if (n.c1 < 0 && n.c2 < 0)
            if (n.cnt < 0 || n.off < 0 || (int64_t)n.off + n.cnt > n_ids) return fail(c, RT_E_ARG, "leaf node %d: triangle range out of bounds", i);
    }
    for (int j = 0; j < n_meshes; j++)
        if (meshes[j].node0Id < 0 || meshes[j].node0Id >= n_nodes) return fail(c, RT_E_ARG, "mesh %d: bad root node", j);

    // ---- traversal-order walk: mesh ascending, depth first, second child first ----
    struct Visit { int node, depth, parent_rec, which; };
    std::vector<VNode> vnodes;
    std::vector<int> vdepth, vnode_id;
    std::vector<PackInfo> pack;
    std::vector<int> root_v, root_ref;     // per root of the walk (= per mesh, or per group of meshes of the SAH structure)
    std::vector<float4> rest;
    std::vector<int> slot_mesh, slot_rem, slot_id;
    std::vector<int> first_visit;
    int max_depth = 0;
    // walks the trees `wn` (roots `wroot[j]`, leaf lists in `wids`) and fills the vectors above
    std::vector<char> listed;              // triangle already met in a leaf of this walk
    // (per_mesh: root j holds exactly the triangles of mesh j, as the reference's trees do; else a root may hold several meshes)
    auto walk = [&](const Std140Node* wn, int wn_count, const int32_t* wids, const std::vector<int>& wroot, bool per_mesh) -> int {
        vnodes.clear(); vdepth.clear(); vnode_id.clear(); pack.clear(); rest.clear();
        slot_mesh.clear(); slot_rem.clear(); slot_id.clear();
        listed.assign((size_t)n_tris, 0);
        first_visit.assign(wn_count, -1);
        max_depth = 0;
        const size_t visit_cap = 4 * (size_t)wn_count + 16;
        const int n_walk_roots = (int)wroot.size();
        const bool wrap_leaf_roots = n_meshes <= c->tlas_above;
        root_v.assign(n_walk_roots, 0);
        root_ref.assign(n_walk_roots, RT_IDX_NONE);
        for (int j = 0; j < n_walk_roots; j++) {
            std::vector<Visit> st;
            st.push_back({wroot[j], 0, -1, 0});
            while (!st.empty()) {
                Visit cur = st.back();
                st.pop_back();
                if (vnodes.size() >= visit_cap) return fail(c, RT_E_LIMIT, "BVH is not a tree (shared sub-graphs blow up)");
                const Std140Node& n = wn[cur.node];
                int v = (int)vnodes.size();
                vnodes.push_back({-1, -1, 0, 0});
                vdepth.push_back(cur.depth);
                vnode_id.push_back(cur.node);
                if (first_visit[cur.node] < 0) first_visit[cur.node] = v;
                if (cur.depth > max_depth) max_depth = cur.depth;
                if (cur.depth + 1 > RT_MAX_STACK) return fail(c, RT_E_LIMIT, "root %d: BVH deeper than %d", j, RT_MAX_STACK - 1);
                int ref;
                const bool leaf = n.c1 < 0 && n.c2 < 0;
                if (leaf) {
                    vnodes[v].start = (int)slot_id.size();
                    vnodes[v].count = n.cnt;
                    for (int i = 0; i < n.cnt; i++) {
                        const Std140Tri& t = tris[wids[n.off + i]];
                        if (per_mesh && t.meshIndex != j)
                            return fail(c, RT_E_ARG, "triangle %d in the BVH of mesh %d belongs to mesh %d", t.triId, j, t.meshIndex);
                        // the reference's builder puts every triangle into exactly one leaf (chunker.py:216-244);
                        // a second listing would make rank and triId disagree for the previous-hit skip (:600)
                        if (listed[t.triId]) return fail(c, RT_E_ARG, "triangle %d is listed in more than one leaf", t.triId);
                        listed[t.triId] = 1;
                        const float* src[6] = {t.posA, t.posB, t.posC, t.nA, t.nB, t.nC};
                        for (int q = 0; q < 6; q++) rest.push_back(make_float4(src[q][0], src[q][1], src[q][2], 0.f));
                        slot_mesh.push_back(t.meshIndex);
                        slot_rem.push_back(n.cnt - 1 - i);
                        slot_id.push_back(t.triId);
                    }
                    ref = n.cnt > 0 ? ~vnodes[v].start : RT_IDX_NONE;     // an empty leaf contributes nothing
                } else {
                    ref = (int)pack.size();
                    pack.push_back({-1, -1, RT_IDX_NONE, RT_IDX_NONE});
                    // push first child first so that the second child is visited first
                    if (n.c1 >= 0) st.push_back({n.c1, cur.depth + 1, ref, 1});
                    if (n.c2 >= 0) st.push_back({n.c2, cur.depth + 1, ref, 2});
                }
                if (cur.parent_rec < 0) { root_v[j] = v; root_ref[j] = ref; }
                else if (cur.which == 1) { pack[cur.parent_rec].c1 = v; pack[cur.parent_rec].ref1 = ref; }
                else { pack[cur.parent_rec].c2 = v; pack[cur.parent_rec].ref2 = ref; }
            }
        }
        // A root of the root table that is a LEAF (a mesh of a few triangles) gets a record of its own: one child, the leaf, and
        // one absent child.  k_trace can then rely on every root it takes from the table being an interior record (it takes the
        // next record up when a node step ends, and "a record in hand" is all it keeps track of); the price is one node step for
        // such a mesh.  Appended after the records of the walk, whose order the loop below relies on.  Not under a top-level
        // tree: there a mesh root is a child of a tree node like any other.
        if (wrap_leaf_roots)
            for (int j = 0; j < n_walk_roots; j++)
                if (root_ref[j] < 0 && root_ref[j] != RT_IDX_NONE) {
                    pack.push_back({root_v[j], -1, root_ref[j], RT_IDX_NONE});
                    root_ref[j] = (int)pack.size() - 1;
                }
        // child visit indices for the refit (vnode.c1/c2), from the pack table
        int rec = 0;
        for (size_t v = 0; v < vnodes.size(); v++) {
            const Std140Node& n = wn[vnode_id[v]];
            if (!(n.c1 < 0 && n.c2 < 0)) {
                vnodes[v].c1 = pack[rec].c1;
                vnodes[v].c2 = pack[rec].c2;
                rec++;
            }
        }
        return RT_OK;
    };
    std::vector<int> ref_roots(n_meshes);
    for (int j = 0; j < n_meshes; j++) ref_roots[j] = meshes[j].node0Id;
    int wrc = walk(nodes, n_nodes, leaf_ids, ref_roots, true);
    if (wrc) return wrc;
    // rank = position in the reference's traversal order: what decides ties (rt_dev_trace.h)
    std::vector<int> slot_rank(slot_id.size()), rank_slot(slot_id.size());
    for (size_t k = 0; k < slot_id.size(); k++) slot_rank[k] = rank_slot[k] = (int)k;
    const std::vector<int> ref_first_visit = first_visit;
    // the reference's leaves as (start, count) ranges of reference ranks, for the SAH structure's candidate gate
    std::vector<std::pair<int, int>> ref_leaves;
    for (const VNode& vn : vnodes)
        if (vn.c1 < 0 && vn.c2 < 0 && vn.count > 0) ref_leaves.push_back({vn.start, vn.count});
    const bool sah = c->accel == RT_ACCEL_SAH && !slot_id.empty();
    if (sah) {
        // ---- opt-in performance structure: a binned-SAH tree per mesh over the same triangles ----
        // Same slots machinery, other topology; every triangle keeps the rank the reference's
        // traversal gave it, so equal-distance hits still resolve as the shader's.  What changes is
        // WHICH boxes are tested: a hit can differ from the exact modes where the reference's own
        // leaf box misses a triangle by rounding.
        std::vector<int> rank_of_tri(n_tris, -1);
        for (size_t k = 0; k < slot_id.size(); k++)
            if (rank_of_tri[slot_id[k]] < 0) rank_of_tri[slot_id[k]] = (int)k;
        std::vector<RtNode48> sn;
        std::vector<int32_t> sids;
        std::vector<int> sroots;
        {
            std::vector<std::vector<int32_t>> per_mesh(n_meshes);
            for (size_t k = 0; k < slot_id.size(); k++)
                if (rank_of_tri[slot_id[k]] == (int)k) per_mesh[slot_mesh[k]].push_back(slot_id[k]);
            // Groups: meshes with fewer than `sah_group_below` triangles share ONE tree (the eight wall / light / ground meshes of
            // the Suzanne room: 2 roots to test per ray instead of 9).  Any structure over the same triangles gives the
            // reference's result - ranks decide ties, the flag word of each triangle carries its mesh, the winner is checked
            // against its reference leaf box - so grouping is a matter of speed only.  The topology is built from the positions
            // handed in (rest poses): it is a good one as long as the members of a group keep their relative poses; the boxes
            // are refit from the posed triangles whatever the topology.  Root-table scenes only (a top-level tree is built over
            // per-mesh roots).
            std::vector<std::vector<int32_t>> per_group;
            int small_group = -1;
            for (int j = 0; j < n_meshes; j++) {
                const bool merge = c->sah_group_below > 0 && n_meshes <= c->tlas_above && !per_mesh[j].empty() && (int)per_mesh[j].size() < c->sah_group_below;
                if (!merge) { per_group.push_back(std::move(per_mesh[j])); continue; }
                if (small_group < 0) { small_group = (int)per_group.size(); per_group.emplace_back(); }
                per_group[small_group].insert(per_group[small_group].end(), per_mesh[j].begin(), per_mesh[j].end());
            }
            sroots.resize(per_group.size());
            for (int j = 0; j < (int)per_group.size(); j++) {
                const std::vector<int32_t>& ids = per_group[j];
                if (ids.empty()) {                                   // keep the (empty) reference leaf
                    RtNode48 e;
                    std::memset(&e, 0, sizeof(e));
                    e.id = (int32_t)sn.size(); e.c1 = e.c2 = -1; e.off = 0; e.cnt = 0;
                    sroots[j] = e.id;
                    sn.push_back(e);
                    continue;
                }
                std::vector<float> pos9(9 * ids.size());
                for (size_t i = 0; i < ids.size(); i++) {
                    const Std140Tri& t = tris[ids[i]];
                    const float* src[3] = {t.posA, t.posB, t.posC};
                    for (int q = 0; q < 3; q++)
                        for (int a = 0; a < 3; a++) pos9[9 * i + 3 * q + a] = src[q][a];
                }
                sroots[j] = rt_sah_build(pos9.data(), ids.data(), (int)ids.size(), RT_MAX_STACK - 2, c->sah_leaf, sn, sids);
            }
        }
        static_assert(sizeof(RtNode48) == sizeof(Std140Node), "node images");
        if ((wrc = walk((const Std140Node*)sn.data(), (int)sn.size(), sids.data(), sroots, false))) return wrc;
        slot_rank.resize(slot_id.size());
        rank_slot.assign(n_tris > 0 ? (size_t)*std::max_element(rank_of_tri.begin(), rank_of_tri.end()) + 1 : 0, 0);
        for (size_t k = 0; k < slot_id.size(); k++) {
            slot_rank[k] = rank_of_tri[slot_id[k]];
            rank_slot[slot_rank[k]] = (int)k;
        }
    }
    // ---- record order: treelets ----
    // The walk numbers interior records depth-first (parent, then the subtree visited first).  For a scene whose records do
    // not fit the L2 (991 k triangles: 25 MB of them) that strews the hot top levels over the whole array and makes a descent
    // touch one cache line per level.  Records are addressed by index only, so they can be laid out freely: breadth-first
    // blocks of `treelet` records (a subtree of ~log2(treelet) levels in consecutive lines), the blocks themselves in
    // depth-first order - the top of every tree first, each subtree's blocks together.  Host-only; traversal order, ranks and
    // slots are untouched, so the image cannot change.  RT_TREELET=0 keeps the depth-first order.
    if (c->treelet > 1 && pack.size() > 2) {
        std::vector<int> order;                       // new index -> old index
        order.reserve(pack.size());
        std::vector<int> roots_todo;
        for (int j = (int)root_ref.size() - 1; j >= 0; j--)
            if (root_ref[j] >= 0) roots_todo.push_back(root_ref[j]);
        std::vector<int> q, frontier;
        while (!roots_todo.empty()) {
            const int r = roots_todo.back();
            roots_todo.pop_back();
            q.assign(1, r);
            frontier.clear();
            int taken = 0;
            for (size_t h = 0; h < q.size(); h++) {
                const int x = q[h];
                if (taken == c->treelet) { frontier.push_back(x); continue; }
                order.push_back(x);
                taken++;
                if (pack[x].ref1 >= 0) q.push_back(pack[x].ref1);
                if (pack[x].ref2 >= 0) q.push_back(pack[x].ref2);
            }
            for (size_t h = frontier.size(); h-- > 0;) roots_todo.push_back(frontier[h]);
        }
        if (order.size() == pack.size()) {            // (always: every record hangs off a root)
            std::vector<int> new_of(pack.size());
            for (size_t n = 0; n < order.size(); n++) new_of[order[n]] = (int)n;
            std::vector<PackInfo> np(pack.size());
            for (size_t n = 0; n < order.size(); n++) {
                PackInfo pi = pack[order[n]];
                if (pi.ref1 >= 0) pi.ref1 = new_of[pi.ref1];
                if (pi.ref2 >= 0) pi.ref2 = new_of[pi.ref2];
                np[n] = pi;
            }
            pack.swap(np);
            for (int& rr : root_ref)
                if (rr >= 0) rr = new_of[rr];
        }
    }
    std::vector<int> ref_leaf_off, ref_leaf_slots;
    if (sah) {
        ref_leaf_off.push_back(0);
        for (auto& lf : ref_leaves) {
            for (int r = lf.first; r < lf.first + lf.second; r++) ref_leaf_slots.push_back(rank_slot[r]);
            ref_leaf_off.push_back((int)ref_leaf_slots.size());
        }
    }
    first_visit = ref_first_visit;          // rt_read_nodes48 maps the REFERENCE's nodes (unavailable with RT_ACCEL_SAH)
    // ---- more than 32 meshes: a top-level tree over the mesh roots replaces the flat mesh loop ----
    // (build_tlas; built here from the rest-pose boxes, rebuilt by run_prep whenever poses change)
    TlasHost th;
    int tlas_root = -1;
    if (n_meshes > c->tlas_above) {
        th.box_lo.assign(3 * (size_t)n_meshes, INFINITY);
        th.box_hi.assign(3 * (size_t)n_meshes, -INFINITY);
        for (int i = 0; i < n_tris; i++) {
            const float* p[3] = {tris[i].posA, tris[i].posB, tris[i].posC};
            const int j = tris[i].meshIndex;
            for (int q = 0; q < 3; q++)
                for (int a = 0; a < 3; a++) {
                    th.box_lo[3 * j + a] = std::min(th.box_lo[3 * j + a], p[q][a]);
                    th.box_hi[3 * j + a] = std::max(th.box_hi[3 * j + a], p[q][a]);
                }
        }
        th.n_mesh_visits = vnodes.size();
        th.n_mesh_recs = pack.size();
        th.mesh_depth = max_depth;
        th.vdepth_mesh = vdepth;
        th.root_v = root_v;
        th.root_ref = root_ref;
        build_tlas(n_meshes, th.box_lo.data(), th.box_hi.data(), th, vnodes, vdepth, pack, tlas_root, max_depth);
        th.vnodes = vnodes;
        th.pack = pack;
    }
    // level lists (deepest first at refit time)
    std::vector<int> level_off, level_nodes;
    level_lists(vdepth, max_depth, level_off, level_nodes);

    // ---- upload ----
    free_scene(c);
    c->n_tris = n_tris; c->n_meshes = n_meshes; c->n_nodes = n_nodes; c->n_spheres = n_spheres;
    c->n_roots = (int)root_v.size();
    c->n_slots = (int)slot_id.size(); c->n_visit = (int)vnodes.size(); c->n_rec = (int)pack.size();
    int rc;
    if ((rc = upload(c, c->rest, rest))) return rc;
    if ((rc = upload(c, c->slot_mesh, slot_mesh))) return rc;
    if ((rc = upload(c, c->slot_rem, slot_rem))) return rc;
    if ((rc = upload(c, c->slot_id, slot_id))) return rc;
    if ((rc = upload(c, c->slot_rank, slot_rank))) return rc;
    if ((rc = upload(c, c->rank_slot, rank_slot))) return rc;
    if ((rc = upload(c, c->ref_leaf_off, ref_leaf_off))) return rc;
    if ((rc = upload(c, c->ref_leaf_slots, ref_leaf_slots))) return rc;
    c->n_ref_leaves = sah ? (int)ref_leaves.size() : 0;
    if ((rc = upload(c, c->vnodes, vnodes))) return rc;
    if ((rc = upload(c, c->pack, pack))) return rc;
    if ((rc = upload(c, c->level_nodes, level_nodes))) return rc;
    if ((rc = upload(c, c->root_v, root_v))) return rc;
    if ((rc = upload(c, c->root_ref, root_ref))) return rc;
    std::vector<MatDev> mats(n_meshes);
    for (int j = 0; j < n_meshes; j++) mats[j] = meshes[j].material;
    if ((rc = upload(c, c->mats, mats))) return rc;
    std::vector<SphereDev> sph(n_spheres);
    if (n_spheres) std::memcpy(sph.data(), spheres96, (size_t)n_spheres * 96);
    if ((rc = upload(c, c->spheres, sph))) return rc;
    std::memcpy(&c->atmo, atmosphere80, 80);
    HIPCK(c, c->geo.alloc(4 * (size_t)c->n_slots));
    HIPCK(c, c->nrm.alloc(4 * (size_t)c->n_slots));
    HIPCK(c, c->nodes.alloc(4 * (size_t)c->n_rec));
    // a small SAH scene with a root table gets the table in its eight octant variants (box_test_oct): the condition under which
    // k_trace keeps the rank -> slot table in LDS, where 16-bit entries make the room (rank_lds_entries, trace_lds_bytes)
    c->root_oct = c->root_oct_allowed && sah && tlas_root < 0 && c->n_roots > 0 && c->n_slots <= 2048;
    HIPCK(c, c->roots.alloc(2 * (size_t)c->n_roots * (c->root_oct ? 8 : 1)));
    HIPCK(c, c->nbox.alloc(2 * (size_t)c->n_visit));
    HIPCK(c, c->xf.alloc(16 * (size_t)std::max(n_meshes, 1)));
    HIPCK(c, c->dirty.alloc((size_t)std::max(n_meshes, 1)));
    c->h_xf.assign(16 * (size_t)std::max(n_meshes, 1), 0.f);
    c->h_dirty.assign((size_t)std::max(n_meshes, 1), 2);        // 2 = copy rest pose as given
    c->h_has_xf.assign((size_t)std::max(n_meshes, 1), 0);
    c->level_off = level_off;
    c->stack_depth = max_depth + 1;
    for (int k = 0; k < 16; k++) c->blocks_per_cu[k] = 0;   // LDS layout and occupancy are per scene
    if (getenv("RT_DEBUG")) fprintf(stderr, "[rt] scene: %d slots, %d records, max depth %d, %s\n", c->n_slots, c->n_rec, max_depth, sah ? "SAH" : "reference structure");
    c->tlas_root = tlas_root;
    c->th = std::move(th);
    c->tlas_pose_dirty = false;
    c->scene_is_sah = sah;
    c->node_first_visit = first_visit;
    c->h_nodes.assign(nodes, nodes + n_nodes);
    c->have_scene = true;
    c->prep_pending = true;
    // host staging vectors die at scope exit: make sure the async copies have been consumed
    HIPCK(c, hipStreamSynchronize(c->stream));
    return run_prep(c);
}

int rt_set_materials(rt_ctx* c, const void* meshes128, int n_meshes, const void* atmosphere80) {
    if (!c) return RT_E_ARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    if (n_meshes != c->n_meshes || !meshes128) return fail(c, RT_E_ARG, "mesh count mismatch");
    HIPCK(c, hipSetDevice(c->device));
    const Std140Mesh* meshes = (const Std140Mesh*)meshes128;
    // pinned staging, so that the copy is stream-ordered and the host does not wait for the device
    // (the previous copy out of the staging buffer has normally long finished)
    HIPCK(c, hipEventSynchronize(c->mats_copied));
    for (int j = 0; j < n_meshes; j++) c->h_mats_pinned[j] = meshes[j].material;
    if (n_meshes) {
        HIPCK(c, hipMemcpyAsync(c->mats.p, c->h_mats_pinned, (size_t)n_meshes * 80, hipMemcpyHostToDevice, c->stream));
        HIPCK(c, hipEventRecord(c->mats_copied, c->stream));
    }
    if (atmosphere80) std::memcpy(&c->atmo, atmosphere80, 80);
    c->flags_pending = true;     // transparentFromBehind lives in the triangles' flag words: k_tri_flags before the next launch
    return RT_OK;
}

int rt_update_mesh_transform(rt_ctx* c, int mesh, const float T16[16]) {
    if (!c || !T16) return RT_E_ARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    if (mesh < 0 || mesh >= c->n_meshes) return fail(c, RT_E_ARG, "mesh index out of range");
    std::memcpy(&c->h_xf[16 * (size_t)mesh], T16, 64);
    c->h_dirty[mesh] = 1;
    c->h_has_xf[mesh] = 1;
    c->tlas_pose_dirty = true;
    c->prep_pending = true;      // batched: all dirty meshes are posed by one launch before the next use
    return RT_OK;
}

int rt_update_mesh_pose(rt_ctx* c, int mesh, const float pos[3], const float quat[4]) {
    if (!c || !pos || !quat) return RT_E_ARG;
    float T[16];
    pose_matrix(pos, quat, T);
    return rt_update_mesh_transform(c, mesh, T);
}

int rt_read_tris28(rt_ctx* c, float* out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    HIPCK(c, hipSetDevice(c->device));
    int rc = run_prep(c);
    if (rc) return rc;
    // static part (padding, mesh index, id) on the host, posed part gathered on the device
    DevBuf<float>& tmp = c->tris_tmp;
    if (tmp.n < 28 * (size_t)std::max(c->n_tris, 1)) {
        HIPCK(c, hipStreamSynchronize(c->stream));
        HIPCK(c, tmp.alloc(28 * (size_t)std::max(c->n_tris, 1)));
    }
    HIPCK(c, hipMemsetAsync(tmp.p, 0, 28 * (size_t)std::max(c->n_tris, 1) * 4, c->stream));
    if (c->n_slots)
        hipLaunchKernelGGL(k_gather_tris28, dim3(blocks_for(c->n_slots)), dim3(256), 0, c->stream, c->geo.p, c->nrm.p,
                           c->slot_id.p, c->n_slots, tmp.p, c->n_tris);
    hipError_t e = hipMemcpyAsync(out, tmp.p, 28 * (size_t)c->n_tris * 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, RT_E_DEVICE, "read_tris28: %s", hipGetErrorString(e));
    // ids and mesh indices: slot tables are authoritative
    std::vector<int> sm(c->n_slots), si(c->n_slots);
    if (c->n_slots) {
        HIPCK(c, hipMemcpy(sm.data(), c->slot_mesh.p, (size_t)c->n_slots * 4, hipMemcpyDeviceToHost));
        HIPCK(c, hipMemcpy(si.data(), c->slot_id.p, (size_t)c->n_slots * 4, hipMemcpyDeviceToHost));
    }
    int32_t* oi = (int32_t*)out;
    for (int i = 0; i < c->n_tris; i++) oi[28 * (size_t)i + 24] = i;
    for (int k = 0; k < c->n_slots; k++) oi[28 * (size_t)si[k] + 23] = sm[k];
    return RT_OK;
}

int rt_read_nodes48(rt_ctx* c, void* out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    if (c->scene_is_sah) return fail(c, RT_E_STATE, "the reference's nodes are not on the device in a scene built with RT_ACCEL_SAH");
    HIPCK(c, hipSetDevice(c->device));
    int rc = run_prep(c);
    if (rc) return rc;
    std::vector<float4> nb(2 * (size_t)c->n_visit);
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (c->n_visit) HIPCK(c, hipMemcpy(nb.data(), c->nbox.p, nb.size() * sizeof(float4), hipMemcpyDeviceToHost));
    Std140Node* o = (Std140Node*)out;
    for (int i = 0; i < c->n_nodes; i++) {
        o[i] = c->h_nodes[i];
        int v = c->node_first_visit[i];
        if (v >= 0) {
            o[i].aabbMin[0] = nb[2 * (size_t)v].x; o[i].aabbMin[1] = nb[2 * (size_t)v].y; o[i].aabbMin[2] = nb[2 * (size_t)v].z;
            o[i].aabbMax[0] = nb[2 * (size_t)v + 1].x; o[i].aabbMax[1] = nb[2 * (size_t)v + 1].y; o[i].aabbMax[2] = nb[2 * (size_t)v + 1].z;
        }
    }
    return RT_OK;
}

int rt_set_camera(rt_ctx* c, const float vp[3], const float m[16], const float pos[3], float dof, float aa) {
    if (!c || !vp || !m || !pos) return RT_E_ARG;
    std::memcpy(c->cam.view_params, vp, 12);
    std::memcpy(c->cam.l2w, m, 64);
    std::memcpy(c->cam.cam_pos, pos, 12);
    c->cam.dof = dof;
    c->cam.aa = aa;
    c->have_cam = true;
    return RT_OK;
}

int rt_resize(rt_ctx* c, int w, int h) {
    if (!c) return RT_E_ARG;
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || (int64_t)w * h > (1 << 28)) return fail(c, RT_E_ARG, "bad image size %dx%d", w, h);
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipStreamSynchronize(c->stream));
    const bool own = (c->accum == c->accum_own.p);
    HIPCK(c, c->accum_own.alloc((size_t)w * h * 3));
    HIPCK(c, hipMemsetAsync(c->accum_own.p, 0, (size_t)w * h * 12, c->stream));
    if (own || !c->accum) c->accum = c->accum_own.p;
    c->cam.width = w;
    c->cam.height = h;
    c->n_accum = 0;
    return RT_OK;
}

int rt_reset_accum(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    if (!c->accum) return fail(c, RT_E_STATE, "no image size");
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipMemsetAsync(c->accum, 0, (size_t)c->cam.width * c->cam.height * 12, c->stream));
    c->n_accum = 0;
    return RT_OK;
}

int rt_bind_accum(rt_ctx* c, void* p) {
    if (!c) return RT_E_ARG;
    if (c->cam.width <= 0 || c->cam.height <= 0 || (!p && !c->accum_own.p)) return fail(c, RT_E_STATE, "no image size");
    c->accum = p ? (float*)p : c->accum_own.p;
    c->n_accum = 0;
    return RT_OK;
}

int rt_set_accum_passes(rt_ctx* c, uint64_t n) {
    if (!c) return RT_E_ARG;
    c->n_accum = n;
    c->sum_mode = true;
    return RT_OK;
}

// rt_render and rt_render_static: same launches, other homes for the launch's scratch
static int render_impl(rt_ctx* c, uint32_t first, uint32_t n_passes, uint32_t stride, int rpp, int max_bounces, int flags, bool use_static) {
    if (!c) return RT_E_ARG;
    int rc = check_ready(c);
    if (rc) return rc;
    if (rpp < 1 || max_bounces < 0 || stride < 1) return fail(c, RT_E_ARG, "bad sampling arguments");
    if (rpp > 65535 || max_bounces > 65535) return fail(c, RT_E_LIMIT, "rays_per_pixel and max_bounces are limited to 65535");
    if (flags != RT_ACCUM_MEAN && flags != RT_ACCUM_SUM) return fail(c, RT_E_ARG, "bad flags");
    if (flags == RT_ACCUM_MEAN && stride != 1) return fail(c, RT_E_ARG, "running mean needs pass_stride 1");
    // raytracer.fs.glsl:931-935: a pass whose index exceeds MAX_CYCLES returns the previous frame
    if (first > c->max_cycles) n_passes = 0;
    else n_passes = (uint32_t)std::min<uint64_t>(n_passes, ((uint64_t)c->max_cycles - first) / stride + 1);
    HIPCK(c, hipSetDevice(c->device));
    // rt_render_static allocates nothing and must be legal under stream capture: a pending pose update would make run_prep
    // copy from pageable memory and synchronise the stream (and may rebuild the top-level tree), so it is refused here -
    // the caller flushes with rt_prepare (or any rt_render / rt_first_hit) first.  A pending materials-only update is one
    // kernel launch (k_tri_flags; its copy was queued from pinned memory by rt_set_materials) and goes through.
    if (use_static && c->prep_pending)
        return fail(c, RT_E_STATE, "rt_render_static: pose updates are pending; call rt_prepare first (it may synchronise, rt_render_static must not)");
    if ((rc = run_prep(c))) return rc;
    TraceArgs a;
    fill_args(c, a);
    a.pass_stride = stride;
    a.rays_per_pixel = rpp;
    a.max_bounces = max_bounces;
    a.sum_mode = (flags == RT_ACCUM_SUM);
    // a job holds all passes of its launch, so a launch takes at most RT_MAX_LAUNCH_PASSES passes; nothing else
    // bounds it (no per-(pass, pixel) buffer any more): configs[1] and configs[3] are one launch each
    const uint64_t n_px = (uint64_t)c->cam.width * c->cam.height;
    const uint64_t per_pass = n_px * (uint64_t)rpp;
    const uint32_t chunk = RT_MAX_LAUNCH_PASSES;
    c->launch_static = use_static;
    for (uint32_t done = 0; done < n_passes; done += chunk) {
        a.first_pass = first + done * stride;
        a.n_passes = std::min(chunk, n_passes - done);
        set_job_shape(c, a, a.n_passes, use_static ? RT_JOB_SAMPLES_STATIC : c->job_samples);
        if (use_static) {                              // no events either: creating one may allocate
            rc = launch_trace(c, a, false);
            if (rc) { c->launch_static = false; return rc; }
        } else {
            LaunchTimer t;
            HIPCK(c, t.create());
            HIPCK(c, hipEventRecord(t.e0, c->stream));
            if ((rc = launch_trace(c, a, false))) return rc;
            HIPCK(c, hipEventRecord(t.e1, c->stream));
            note_launch(c, t.e0, t.e1);
            t.handed = true;
        }
        c->st.launches++;
    }
    c->launch_static = false;
    HIPCK(c, hipGetLastError());
    c->n_accum += n_passes;
    c->sum_mode = (flags == RT_ACCUM_SUM);
    c->st.samples += per_pass * n_passes;
    return RT_OK;
}

int rt_render(rt_ctx* c, uint32_t first, uint32_t n_passes, uint32_t stride, int rpp, int max_bounces, int flags) {
    return render_impl(c, first, n_passes, stride, rpp, max_bounces, flags, false);
}

int rt_render_static(rt_ctx* c, uint32_t first, uint32_t n_passes, uint32_t stride, int rpp, int max_bounces, int flags) {
    return render_impl(c, first, n_passes, stride, rpp, max_bounces, flags, true);
}

int rt_prepare(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    HIPCK(c, hipSetDevice(c->device));
    return run_prep(c);
}

int rt_set_max_cycles(rt_ctx* c, uint32_t max_cycles) {
    if (!c) return RT_E_ARG;
    c->max_cycles = max_cycles;
    return RT_OK;
}

int rt_first_hit(rt_ctx* c, uint32_t pass, int32_t* ids, float* ts) {
    if (!c || !ids || !ts) return RT_E_ARG;
    int rc = check_ready(c);
    if (rc) return rc;
    HIPCK(c, hipSetDevice(c->device));
    if ((rc = run_prep(c))) return rc;
    const size_t n = (size_t)c->cam.width * c->cam.height;
    if (c->fh_ids.n < n || c->fh_ts.n < n) {               // grow-only device-side results
        HIPCK(c, hipStreamSynchronize(c->stream));
        if (c->fh_ids.alloc(n) != hipSuccess || c->fh_ts.alloc(n) != hipSuccess) {
            c->fh_ids.release();
            c->fh_ts.release();
            return fail(c, RT_E_NOMEM, "first_hit: out of device memory");
        }
    }
    TraceArgs a;
    fill_args(c, a);
    a.ids = c->fh_ids.p;
    a.ts = c->fh_ts.p;
    a.first_pass = pass;
    a.n_passes = 1;
    a.pass_stride = 1;
    a.rays_per_pixel = 1;
    a.max_bounces = 0;
    a.sum_mode = 0;
    set_job_shape(c, a, 1, 256);                            // 16 x 16 pixels
    {
        LaunchTimer t;
        HIPCK(c, t.create());
        HIPCK(c, hipEventRecord(t.e0, c->stream));
        if ((rc = launch_trace(c, a, true))) return rc;
        HIPCK(c, hipEventRecord(t.e1, c->stream));
        note_launch(c, t.e0, t.e1);
        t.handed = true;
    }
    c->st.launches++;
    c->st.samples += n;
    HIPCK(c, hipGetLastError());
    HIPCK(c, hipMemcpyAsync(ids, c->fh_ids.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipMemcpyAsync(ts, c->fh_ts.p, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int rt_read_accum(rt_ctx* c, float* out, uint64_t* n_out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->accum) return fail(c, RT_E_STATE, "no image size");
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipMemcpyAsync(out, c->accum, (size_t)c->cam.width * c->cam.height * 12, hipMemcpyDeviceToHost, c->stream));
    // k_trace's end-of-kernel check (a job slot still in use = radiance that never reached the accumulator): an image with
    // missing pixels must not be handed out as RT_OK.  Reported once, then cleared.
    unsigned long long unfolded = 0;
    HIPCK(c, hipMemcpyAsync(&unfolded, c->counters.p + 15, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (unfolded) {
        (void)hipMemsetAsync(c->counters.p + 15, 0, 8, c->stream);
        return fail(c, RT_E_DEVICE, "internal error: %llu jobs were not folded into the accumulator", unfolded);
    }
    if (n_out) *n_out = c->n_accum;
    return RT_OK;
}

int rt_read_mean(rt_ctx* c, float* out) {
    int rc = rt_read_accum(c, out, nullptr);
    if (rc) return rc;
    if (c->sum_mode && c->n_accum > 0) {
        const float d = (float)c->n_accum;
        const size_t n = (size_t)c->cam.width * c->cam.height * 3;
        for (size_t i = 0; i < n; i++) out[i] = out[i] / d;
    }
    return RT_OK;
}

int rt_set_accel(rt_ctx* c, int accel) {
    if (!c) return RT_E_ARG;
    if (accel != RT_ACCEL_REFERENCE && accel != RT_ACCEL_SAH) return fail(c, RT_E_ARG, "unknown acceleration structure %d", accel);
    c->accel = accel;            // takes effect at the next rt_set_scene
    return RT_OK;
}

int rt_set_cull(rt_ctx* c, int mode) {
    if (!c) return RT_E_ARG;
    if (mode != RT_CULL_OFF && mode != RT_CULL_ON && mode != RT_CULL_BAND) return fail(c, RT_E_ARG, "unknown culling mode %d", mode);
    c->cull = mode;
    return RT_OK;
}

int rt_enable_stats(rt_ctx* c, int e) {
    if (!c) return RT_E_ARG;
    c->stats_on = e != 0;
    return RT_OK;
}

int rt_reset_stats(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int rc = resolve_events(c);
    if (rc) return rc;
    HIPCK(c, hipMemsetAsync(c->counters.p, 0, 128, c->stream));
    c->st = rt_stats_t{};
    return RT_OK;
}

int rt_get_info(rt_ctx* c, rt_info_t* o) {
    if (!c || !o) return RT_E_ARG;
    std::memset(o, 0, sizeof(*o));
    if (!c->have_scene) return fail(c, RT_E_STATE, "no scene");
    o->accel = c->scene_is_sah ? RT_ACCEL_SAH : RT_ACCEL_REFERENCE;
    o->n_slots = c->n_slots;
    o->n_records = c->n_rec;
    o->top_level_tree = c->tlas_root >= 0;
    o->stack_entries = c->stack_depth;
    o->stack_entries_lds = c->last_stack_lds;
    o->lds_bytes = c->last_lds_bytes;
    o->workgroups_per_cu = c->last_blocks_per_cu;
    o->top_level_rebuilds = c->tlas_rebuilds;
    o->n_roots = c->n_roots;
    o->root_table_variants = c->last_root_variants;
    return RT_OK;
}

int rt_get_stats(rt_ctx* c, rt_stats_t* o) {
    if (!c || !o) return RT_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int rc = resolve_events(c);
    if (rc) return rc;
    unsigned long long h[16];
    HIPCK(c, hipMemcpyAsync(h, c->counters.p, 128, hipMemcpyDeviceToHost, c->stream));
    HIPCK(c, hipStreamSynchronize(c->stream));
    if (h[15]) {                                   // reported once (here or by rt_read_accum), then cleared
        (void)hipMemsetAsync(c->counters.p + 15, 0, 8, c->stream);
        return fail(c, RT_E_DEVICE, "internal error: %llu jobs were not folded into the accumulator", h[15]);
    }
    c->st.rays = h[0]; c->st.box_tests = h[1]; c->st.tri_tests = h[2]; c->st.shaded_hits = h[3];
    for (int q = 0; q < 6; q++) c->st.sched[q] = h[4 + q];
    for (int q = 0; q < 3; q++) c->st.cull_audit[q] = h[10 + q];
    *o = c->st;
    return RT_OK;
}

// ---- rt_allreduce_sum: the one exchange step of the spp-sharded path, for hosts without torch ----
// RCCL is resolved at the first call, not linked: the library has no RCCL dependency for the callers that never combine
// shards.  The communicator belongs to ONE RCCL instance (a process can hold several: torch ships its own), so the
// instance is looked for in this order: RT_RCCL_LIB (a path), an instance the process has already loaded under the
// usual names (RTLD_NOLOAD: the one the caller created its communicator with), then a fresh load of librccl.so.1.
namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*nccl_errstr_fn)(int);
struct RcclApi { nccl_allreduce_fn allreduce = nullptr; nccl_errstr_fn errstr = nullptr; char from[256] = {0}; };
RcclApi g_rccl;
std::once_flag g_rccl_once;           // contexts of several host threads may make their first call at the same time
void load_rccl_once() {
    void* h = nullptr;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    if (const char* e = getenv("RT_RCCL_LIB")) {
        h = dlopen(e, RTLD_NOW | RTLD_LOCAL);
        snprintf(g_rccl.from, sizeof(g_rccl.from), "%s (RT_RCCL_LIB)", e);
    }
    for (int k = 0; !h && k < 2; k++) {
        h = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD);
        if (h) snprintf(g_rccl.from, sizeof(g_rccl.from), "%s (already loaded)", names[k]);
    }
    for (int k = 0; !h && k < 1; k++) {
        h = dlopen(names[k], RTLD_NOW | RTLD_LOCAL);
        if (h) snprintf(g_rccl.from, sizeof(g_rccl.from), "%s", names[k]);
    }
    if (!h) return;
    g_rccl.allreduce = (nccl_allreduce_fn)dlsym(h, "ncclAllReduce");
    g_rccl.errstr = (nccl_errstr_fn)dlsym(h, "ncclGetErrorString");
}
void load_rccl() { std::call_once(g_rccl_once, load_rccl_once); }
}  // namespace

int rt_allreduce_sum(rt_ctx* c, void* nccl_comm, uint64_t total_passes) {
    if (!c || !nccl_comm) return RT_E_ARG;
    if (!c->accum) return fail(c, RT_E_STATE, "no image size");
    if (!c->sum_mode && c->n_accum > 0) return fail(c, RT_E_STATE, "the accumulator holds a running mean: render the shards with RT_ACCUM_SUM");
    load_rccl();
    if (!g_rccl.allreduce) return fail(c, RT_E_DEVICE, "RCCL not found (librccl.so.1; RT_RCCL_LIB names another)");
    HIPCK(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->cam.width * c->cam.height * 3;
    const int rc = g_rccl.allreduce(c->accum, c->accum, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, nccl_comm, c->stream);
    if (rc != 0) return fail(c, RT_E_DEVICE, "ncclAllReduce (%s): %s", g_rccl.from, g_rccl.errstr ? g_rccl.errstr(rc) : "error");
    c->n_accum = total_passes;
    c->sum_mode = true;
    return RT_OK;
}

int rt_sync(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    HIPCK(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

void* rt_stream(rt_ctx* c) { return c ? (void*)c->stream : nullptr; }

int rt_set_stream(rt_ctx* c, void* s) {
    if (!c) return RT_E_ARG;
    HIPCK(c, hipSetDevice(c->device));
    int rc = resolve_events(c);          // events belong to the old stream
    if (rc) return rc;
    HIPCK(c, hipStreamSynchronize(c->stream));
    c->stream = (s == RT_STREAM_OWN) ? c->own_stream : (hipStream_t)s;   // NULL = the null stream
    return RT_OK;
}

}  // extern "C"
