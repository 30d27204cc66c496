// rt_dev_trace.h - device data layout and the wave-scheduled path tracer (gfx950).
//
// Restates, for one ray per lane of a 64-wide wavefront:
//   boundingBoxIntersect   shaders/raytracer.fs.glsl:421-453
//   rayTriangle            :305-409  (plane + three edge tests; the per-triangle constants
//                                     N = cross(B-A, C-A) and d = -dot(N,A) are precomputed
//                                     by the scene-prep kernel with the same expression tree)
//   raySphere              :251-276
//   calculateRayCollision  :518-641
//   traceRay               :691-915
//   main()                 :928-1044
//
// Data layout in HBM (built by rt_set_scene, refreshed by the prep kernels):
//   tri_geo  4 x float4 per triangle SLOT (64 B, one half cache line):
//              {A.xyz, N.x} {B.xyz, N.y} {C.xyz, N.z} {d, bits(next), bits(rank), bits(flags)}
//            slots are in traversal order of the reference (mesh ascending, depth-first,
//            second child first, leaf order), so (a) every leaf and every subtree is one
//            contiguous range, (b) the slot index is the rank that decides ties exactly as
//            the reference's "first found wins" (strict <, :625) whatever order the kernel
//            visits nodes in (rank = slot, stored explicitly so that a scene built with
//            RT_ACCEL_SAH, whose slots follow another tree, resolves ties the same way).
//            next = byte offset of the leaf's next slot in tri_geo, -1 after its last triangle (a lane walks a leaf by
//            loading this word: no count, no add, no compare);
//            flags = mesh index | RT_TFB_BIT (the mesh's transparentFromBehind).
//   tri_hit  4 x float4 per slot (64 B), read only for the WINNER of a query - one scattered cache line per hit:
//              {nA.xyz, bits(mesh)} {nB.xyz, lo.x} {nC.xyz, lo.y} {lo.z, hi.x, hi.y, hi.z}
//            corner normals, mesh index and (RT_ACCEL_SAH scenes) the box [lo, hi] of the triangle's leaf in the
//            REFERENCE's tree (what decides whether a hit found through the SAH tree counts).  Until round 3 normals
//            (48 B) and boxes (32 B) were two arrays: two lines per hit, which is what a scene that does not fit the L2
//            (991 k triangles: 80 MB of them) pays for.
//   nodes    4 x float4 per INTERIOR node: both children's boxes and references
//              {lo1.xyz, bits(ref1)} {hi1.xyz, bits(ref2)} {lo2.xyz, e1} {hi2.xyz, e2}
//            ref >= 0: BYTE OFFSET of an interior record (index << 6); ref < 0 and != RT_NONE: leaf, byte offset of its
//            first slot in tri_geo = ~ref (RT_REF_* below: no shift per step; records and slots are both 64 B);
//            e = slack of the child's widened slab test (RT_ACCEL_SAH scenes; 0 otherwise).
//   roots    2 x float4 per mesh: {lo.xyz, bits(ref)} {hi.xyz, e} of the mesh's root node.
//   mats     the 80-byte std140 Material of every mesh, as uploaded.
//   stage    per wave: two slots of one float3 per sample of a JOB (see below); the wave that traced a
//            job folds it into the accumulator itself, passes in order.
//
// Execution model: persistent waves take JOBS from a launch-wide counter.  A job is a
// small tile of pixels (tile_w x tile_h, powers of two) with ALL passes of the launch, ~512 samples;
// every lane runs one sample at a time and takes the next one of its wave's job when it is done
// (pass-major inside the job, so the 64 lanes of a wave trace a handful of pixels over several
// passes at once).  The radiance of each sample is parked in the wave's staging slot of the job;
// when the last sample of a job has finished, the wave applies the running mean (or the sum) to the
// job's pixels pass by pass, in order, with the reference's expressions (:1008-1021) - the only
// place the accumulator is touched: 12 bytes read and 12 written per pixel per launch.  A wave
// holds two jobs at most: the one it is handing out and the previous one whose last samples are
// still in flight.  A lane holds node work (record in hand,
// node stack, mesh roots to visit) and leaf work (triangle in progress, 2-entry leaf queue) at
// once; the wave votes (ballot + popcount) between a node step and a triangle step, runs
// triangle steps in bursts, and when fewer than `refill_below` lanes still traverse, the lanes
// that wait shade their hit and start their next ray in one path block.  The order in which a
// lane meets its boxes and triangles changes with the votes; the result does not: it is the
// minimum of (t, slot rank) over the same set of tests.  Node stack entries beyond the ones that
// fit in LDS at full occupancy live in global memory (DEEP instantiations).
#pragma once
#include "rt_dev_math.h"

// Wave priority (s_setprio, round 3): the traversal steps are bound by vector-instruction issue, the path block by the latency of
// its dependent loads (winner's rank -> records -> material).  With the traversal at a higher priority than the path block
// the issue slots go to the waves that can use every one of them, and a wave in its path block fills the gaps it would
// have left anyway: -1.5 % on the bench frame for any traversal priority 1..3 over path priority 0; the reverse
// (path block first) +-0, path 1 over traversal 0 +1 % (profiles/r03_k_trace_isa.txt).
#ifndef RT_PRIO_TRAV
#define RT_PRIO_TRAV 1
#endif
#ifndef RT_PRIO_PATH
#define RT_PRIO_PATH 0
#endif
// child / root references as the device stores them (k_pack_nodes): byte offsets, see the layout above.  ("No record" as -1, an
// inline constant of the instruction set instead of a value that has to sit in a register: measured +-0, round 3.)
#define RT_NONE ((int)0x80000000)
#define RT_REF_OF_RECORD(i) ((int)((uint32_t)(i) << 6))
#define RT_REF_OF_LEAF(slot) (~(int)((uint32_t)(slot) << 6))
#define RT_LEAF_OFFSET(ref) (~(ref))
#define RT_IDX_NONE ((int)0x80000000)      // host-side index form of a reference (PackInfo): absent
#define RT_BLOCK 256
#define RT_MISS_RANK 0x7fffffff
#define RT_SPHERE_RANK (-(1 << 30))

// Distance culling (CULL builds only).  A box is skipped when
//  * its entry distance lies beyond best_t*(1+RT_CULL_REL)+RT_CULL_ABS (nothing in it can beat the
//    hit in hand), or
//  * its exit distance is below RT_CULL_BEHIND = 5e-7: the ray has left the box before t = 1e-6,
//    the smallest distance rayTriangle accepts (:349), so a hit in it would have to lie outside
//    the box of its own leaf.  This is what removes the 12 triangles of the slab a bounce ray
//    starts on (its origin sits on the slab's face: exit distance ~ +-1e-7).
// Both are exact in real arithmetic; in float32 a hit within rounding (~1e-7) of BOTH a triangle
// edge and its box face could be dropped - never observed: 0 differing pixels against STRICT in
// 32 frames x 1920x1080 x 64 passes (2.1e10 rays) and in every test (tools/cmp_cull.py).
#define RT_CULL_ABS 0x1p-10f
#define RT_CULL_REL 0x1p-10f
#ifndef RT_CULL_BEHIND
#define RT_CULL_BEHIND 5e-7f
#endif

struct MatDev {          // std140 Material, 80 bytes (raytracer.fs.glsl:53-68)
    float color[4];
    float specularColor[3]; float pad0;
    float emissionColor[3]; float emissionStrength;
    float specularStrength, smoothness, transmission, ior;
    float metallic; int transparentFromBehind; int pad1[2];
};
static_assert(sizeof(MatDev) == 80, "std140 Material");

struct SphereDev { float position[3]; float radius; MatDev material; };
static_assert(sizeof(SphereDev) == 96, "std140 Sphere");

struct DevScene {
    const float4* tri_geo;
    const float4* tri_hit;
    const float4* nodes;
    const float4* roots;
    const MatDev* mats;
    const SphereDev* spheres;
    const int* rank_slot;  // reference rank -> slot (identity unless the scene was built with RT_ACCEL_SAH)
    const int* slot_id;    // triId of each slot
    int n_meshes;
    int n_roots;          // entries of `roots`: one per mesh, or fewer when the SAH structure groups small meshes into one tree
    int n_spheres;
    int stack_depth;      // node stack entries a lane has in LDS (rt_hip.hip sizes it for full occupancy); DEEP builds: deeper entries go to stack_ovf
    int tlas_root;        // > 32 meshes: record of the top-level tree over the mesh roots; else -1 (roots in LDS)
    int rank_lds;         // entries of rank_slot that k_trace copies into LDS (all of them or 0: rt_hip.hip::trace_lds_bytes)
};

struct DevCamera {
    float view_params[3];
    float l2w[16];
    float cam_pos[3];
    float dof, aa;
    int width, height;
};

struct Counters { unsigned long long rays, box, tri, hits; };
// wave-level scheduler statistics (STATS builds): executions of each block and lanes active in them
struct SchedStats { unsigned long long node_steps, node_lanes, tri_steps, tri_lanes, path_steps, path_lanes; };

// The reference's slab test, comparison for comparison (NaN compares false, no t >= 0 test).
// On a hit, [tmin, tmax] is the entry/exit interval the reference ends up with.
__device__ __forceinline__ bool box_test(v3 o, v3 inv, float4 lo, float4 hi, float& tmin_out, float& tmax_out) {
    float tmin = (lo.x - o.x) * inv.x, tmax = (hi.x - o.x) * inv.x, t;
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo.y - o.y) * inv.y, tymax = (hi.y - o.y) * inv.y;
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    bool hit = !((tmin > tymax) || (tymin > tmax));
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo.z - o.z) * inv.z, tzmax = (hi.z - o.z) * inv.z;
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    hit = hit && !((tmin > tzmax) || (tzmin > tmax));
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    tmin_out = tmin;
    tmax_out = tmax;
    return hit;
}

// Same test for rays whose 1/d components are all finite (every ray except those with an
// exactly zero direction component): no slab value can then be NaN (finite*finite, or
// +-inf from overflow), and for non-NaN values the reference's compare-and-swap ladder is
// exactly  max(entry_x, entry_y, entry_z) <= min(exit_x, exit_y, exit_z)  with per-axis
// entry/exit = min/max of the two slab values (proof: DESIGN.md 5).  A v_min/v_max costs 4.7
// cycles where the v_cmp + v_cndmask pair of a ladder step costs 8.2 (tools/ubench/valu_rate.hip).
__device__ __forceinline__ bool box_test_finite(v3 o, v3 inv, float4 lo, float4 hi, float& tmin_out, float& tmax_out) {
    const float ax = (lo.x - o.x) * inv.x, bx = (hi.x - o.x) * inv.x;
    const float ay = (lo.y - o.y) * inv.y, by = (hi.y - o.y) * inv.y;
    const float az = (lo.z - o.z) * inv.z, bz = (hi.z - o.z) * inv.z;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    tmin_out = tmin;
    tmax_out = tmax;
    return tmin <= tmax;
}

// Octant form of the same test (round 3; the ROCT instantiations' root loop).  For a ray with finite 1/d the per-axis entry /
// exit values of the min/max form follow from the SIGN of 1/d alone: a box has lo <= hi, float subtraction and multiplication
// are monotonic, so with 1/d > 0  (lo - o) * (1/d) <= (hi - o) * (1/d)  and min(a, b) IS a, max(a, b) IS b (where the two
// coincide the sign of a zero may differ: every use of the interval is a comparison); with 1/d < 0 the other way round.  A table
// that holds each box once per direction octant, every axis already in (near, far) order, is therefore tested with six
// subtract / multiply pairs, one max3 and one min3: six min / max fewer per box, at 4.7 issue cycles each where the f32
// arithmetic costs 2.4 (tools/ubench/valu_rate.hip).  An all-NaN box (a mesh without triangles) is NaN in either order.
__device__ __forceinline__ bool box_test_oct(v3 o, v3 inv, float4 near, float4 far, float& tmin_out, float& tmax_out) {
    const float ax = (near.x - o.x) * inv.x, bx = (far.x - o.x) * inv.x;
    const float ay = (near.y - o.y) * inv.y, by = (far.y - o.y) * inv.y;
    const float az = (near.z - o.z) * inv.z, bz = (far.z - o.z) * inv.z;
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(ax, ay), az);
    const float tmax = __builtin_fminf(__builtin_fminf(bx, by), bz);
    tmin_out = tmin;
    tmax_out = tmax;
    return tmin <= tmax;
}
// the octant bit of an axis: 1/d negative (v_cmp_class: negative normal | negative denormal; the table is only read by rays
// whose 1/d are all finite, the others take the compare ladder on the plain (lo, hi) table)
__device__ __forceinline__ bool oct_axis_bit(float inv_a) { return __builtin_amdgcn_classf(inv_a, (1 << 3) | (1 << 4)); }

// true when the culling rules allow skipping a box with interval [tmin,tmax]; NaN never culls.
// CULL: RT_CULL_OFF (0) no rule, RT_CULL_ON (1) both rules, RT_CULL_BAND (2) the distance band alone (rule (a): the
// only way it can drop a hit is a triangle distance that is off by more than the band)
template <int CULL>
__device__ __forceinline__ bool culled(float tmin, float tmax, float best_t) {
    if (!CULL) return false;
    return (CULL == 1 && tmax < RT_CULL_BEHIND) || (tmin > fma_(best_t, RT_CULL_REL, best_t) + RT_CULL_ABS);
}

// SAH structure only.  Its boxes are not the reference's, so their slab tests are made
// conservative: a box counts as hit when  tmin <= tmax + (e2 + 2^-18 max|o|) * max|1/d|, where
// e2 = 2 * (2^-18 max|box coordinate| + 2^-22 + pad) is stored with the box (pad: the triangles' edge and sliver-tip
// overshoot, rt_hip.hip::sliver_tip_pad).  That is the slab test of the box widened by at least e2/2 + 2^-19 max|o| on every
// side (each axis widens its interval by e|1/d_a| <= e max|1/d|): 16 x / 10 x the rounding of the slab arithmetic and of the
// hit point (<= 4 x 2^-24 of the coordinates + 3 x 2^-24 of the origin; derivation in DESIGN.md 5), so a ray whose
// triangle test can pass always reaches the triangle.  The unwidened [tmin, tmax] still drives
// the culling rules and the near-first order.  Which triangles COUNT is decided by the
// reference's own leaf box (in the hit record, see the path block).
__device__ __forceinline__ float sah_slack(v3 o, v3 inv) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(inv.x), __builtin_fabsf(inv.y)), __builtin_fabsf(inv.z));
}
__device__ __forceinline__ float sah_org(v3 o) {
    return __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fabsf(o.z)) * 0x1p-18f;
}
__device__ __forceinline__ bool sah_widened_hit(float tmin, float tmax, float e2, float mo, float mi) {
    return tmin <= fma_(e2 + mo, mi, tmax);
}
// a ray with an exactly zero direction component (slab values may be NaN): the reference's
// ladder on the box widened in space.  The interval handed to the culling rules is the widened
// box's too: on an axis with 1/d = inf the unwidened box yields (-inf, -inf) or (+inf, +inf) as
// soon as the ray passes a rounding error outside it, which the rules would read as "behind the
// origin" / "beyond the best hit" and drop a box the widened test had just accepted
// (tests/test_gpu_fuzz.py::test_axis_aligned_rays_and_nan_slabs).  The widened interval contains
// the unwidened one whenever both are numbers, and NaN never culls.
__device__ __forceinline__ bool sah_box_test_any(v3 o, v3 inv, float4 lo, float4 hi, float e2, float mo, float& tmin_out, float& tmax_out) {
    const float e = e2 * 0.5f + mo;
    return box_test(o, inv, make_float4(lo.x - e, lo.y - e, lo.z - e, 0.f), make_float4(hi.x + e, hi.y + e, hi.z + e, 0.f), tmin_out, tmax_out);
}

// Calibration (timing experiments only, profiles/r03_k_trace_isa.txt): N dummy instructions at the END of a block - what one more
// vector / scalar instruction costs THERE.  -DRT_CAL_NODE_VALU=8, -DRT_CAL_NODE_SALU=16, -DRT_CAL_TRI_VALU=8, -DRT_CAL_PATH_VALU=64
template <int N>
__device__ __forceinline__ void cal_valu() {
    if (N > 0) {
        float d = 0.0f;
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("v_max_f32 %0, %0, %0" : "+v"(d));
    }
}
template <int N>
__device__ __forceinline__ void cal_salu() {
    if (N > 0) {
        int d = 0;
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d) : : "scc");
    }
}
#ifndef RT_CAL_NODE_VALU
#define RT_CAL_NODE_VALU 0
#endif
#ifndef RT_CAL_NODE_SALU
#define RT_CAL_NODE_SALU 0
#endif
#ifndef RT_CAL_TRI_VALU
#define RT_CAL_TRI_VALU 0
#endif
#ifndef RT_CAL_PATH_VALU
#define RT_CAL_PATH_VALU 0
#endif

// A per-lane boolean as the wave's 64-bit lane mask and back: logic on masks runs on the scalar unit
__device__ __forceinline__ uint64_t lane_mask(bool b) { return __builtin_amdgcn_ballot_w64(b); }
__device__ __forceinline__ bool in_mask(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// lanes in a mask, as a 32-bit scalar the compiler knows nothing about: it widens the compare of two __builtin_popcountll results
// to 64 bits, for which the scalar unit has no ordered compare - and issues a VECTOR compare in the middle of the vote
__device__ __forceinline__ int lanes_in(uint64_t m) {
    int c;
    asm("s_bcnt1_i32_b64 %0, %1" : "=s"(c) : "s"(m) : "scc");
    return c;
}

// 2 m + b as ONE vector instruction (add with carry-in from the lane mask of b); the compiler's own form is select + shift-or
__device__ __forceinline__ uint32_t shl1_add(uint32_t m, uint64_t lanes_of_b) {
    uint32_t r;
    uint64_t carry_out;
    asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(r), "=s"(carry_out) : "v"(m), "s"(lanes_of_b));
    return r;
}

// p += bytes after a store through p + bytes: as C the compiler computes the sum once, before the store, into a second register
// and copies it back where the branches join (add + move); the tied operand keeps it one instruction, in place, after the store
template <uint32_t BYTES>
__device__ __forceinline__ void bump_in_place(int& p) {
    asm volatile("v_add_u32 %0, %1, %0" : "+v"(p) : "n"(BYTES) : "memory");
}

// Entry distance of a queued leaf, kept in 16 bits (LDS is what limits occupancy here): the
// top half of the binary32 pattern.  Truncation rounds a positive distance down, so the re-check
// when the leaf is taken up can only cull less than the full value would; a negative distance
// (the origin is inside or past the box) moves towards zero but keeps its sign, and no
// non-positive distance is ever beyond the best hit; a NaN produced by arithmetic has its quiet
// bit (bit 22) set and stays NaN (never culls).
__device__ __forceinline__ unsigned short pack_dist(float t) { return (unsigned short)(__float_as_uint(t) >> 16); }
__device__ __forceinline__ float unpack_dist(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }

// fragPosition -> world-space view point (raytracer.fs.glsl:961-962, raytracer.vs.glsl:6-9)
__device__ __forceinline__ v3 view_point_of(const DevCamera& cam, int px, int py) {
    float xn = (2.0f * ((float)px + 0.5f)) / (float)cam.width - 1.0f;
    float yn = (2.0f * ((float)py + 0.5f)) / (float)cam.height - 1.0f;
    float lx = (xn * 0.5f) * cam.view_params[0], ly = (yn * 0.5f) * cam.view_params[1], lz = cam.view_params[2];
    float r[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float t = cam.l2w[4 * i + 0] * lx;
        t = fma_(cam.l2w[4 * i + 1], ly, t);
        t = fma_(cam.l2w[4 * i + 2], lz, t);
        r[i] = t + cam.l2w[4 * i + 3];
    }
    return V3(r[0], r[1], r[2]);
}

// M_WAIT: the lane needs a sample but both job slots of its wave are still in use (the previous job's last samples
// are in flight): it asks again at the next path block
enum { M_TRAV = 0, M_WAIT = 1, M_PATH = 2, M_DONE = 3 };

// deferred leaves a lane can hold (the leaf queue); a lane joins a node step only while two
// slots are free, because one node step can find two leaves
#ifndef RT_LQ
#define RT_LQ 2
#endif
// RT_ACCEL_SAH scenes: a third slot lets a lane with one leaf waiting still join node steps (leaves
// of <= 4 triangles are soon done, so the later culling costs little): lanes per node step
// 38.7 -> 41.7, per triangle step 34.1 -> 38.7, 2-3 % on every scene.  With the reference's
// larger leaves the same slot costs 8 % (triangle tests start later and cull less).
#ifndef RT_LQ_SAH
#define RT_LQ_SAH 3
#endif
#define RT_LQ_OF(sah) ((sah) ? RT_LQ_SAH : RT_LQ)
#define RT_LQ64_OF(sah, cull, deep) ((sah) && (cull) != 0 && !(deep))   // 8-byte leaf queue entries (see trace_pixels)
#define RT_MIN_STACK_LDS 4         // node stack entries a lane always has in LDS
#define RT_TFB_BIT 0x40000000      // tri_geo[3].w = mesh index | RT_TFB_BIT if the mesh is transparentFromBehind

struct TraceArgs {
    DevScene scene;
    DevCamera cam;
    MatDev atmo;
    float* accum;                 // W*H*3 (render)
    int32_t* ids;                 // W*H   (first hit)
    float* ts;                    // W*H   (first hit)
    unsigned long long* counters; // 4
    float* stage;                 // launch waves x 2 slots x job_cap x 3: radiance of the samples of the jobs in flight
    unsigned int* work_counter;   // next job to hand out (zeroed before the launch)
    int* stack_ovf;               // node stack entries beyond the LDS ones, a region per workgroup: entry e of thread t at [(e - stack_depth) * RT_BLOCK + t]
    uint32_t ovf_block_bytes;     // bytes of a workgroup's region ((scene depth - stack_depth) x RT_BLOCK x 4)
    uint32_t n_jobs, n_tx, n_ty;  // jobs = n_ty x n_tx tiles of (1 << tile_lx) x (1 << tile_ly) pixels, each with all n_passes passes
    uint32_t tile_lx, tile_ly;
    uint32_t job_cap;             // samples a staging slot holds (>= pixels of a tile x n_passes)
    uint32_t first_pass, n_passes, pass_stride;
    int rays_per_pixel, max_bounces, sum_mode;
    int recheck_all;              // SAH, tests only: repeat every query that hit a triangle with the per-hit check on
    int refill_below;             // traversal yields to the path block when fewer lanes than this traverse
    int node_bias;                // builds with -DRT_NODE_BIAS_KNOB only: a node step is taken when cN + node_bias > cT (default vote: cN > cT)
    int burst8;                   // a triangle burst lasts while 8 * (lanes with leaf work) >= burst8 * (lanes traversing at the vote)
};

// Everything only the path block reads (camera, atmosphere, sampling parameters, output and
// material pointers) is read from an LDS copy of the kernel arguments there: held in scalar
// registers across the traversal loop it costs ~70 SGPRs and spills.
// Per-wave bookkeeping of the two jobs a wave can hold (a lane's `elem` says which slot its sample is in): tile origin, samples that
// must finish before the job can be folded (0 = slot free), samples finished so far.  Wave-private LDS, 16 bytes:
// the Suzanne room on the reference's BVH fills the LDS of six workgroups per CU to within 80 bytes.
struct WaveJobs { uint16_t ox[2], oy[2], expected[2], done[2]; };
struct PathConsts { TraceArgs a; WaveJobs wj[RT_BLOCK / 64]; };
#define RT_PC_F4 ((int)((sizeof(PathConsts) + 15) / 16))   // float4 slots of LDS it takes

// material index encoding: -1 atmosphere (kernel argument), [0,256) mesh, 256+i sphere i
__device__ __forceinline__ const MatDev* material_of(const DevScene& S, int idx) {
    return idx < 256 ? &S.mats[idx] : &S.spheres[idx - 256].material;
}

// Traversal registers of one lane.  A lane holds node work (cur / node stack / mesh roots not
// yet visited) and leaf work (the triangle in progress / the leaf queue) at the same time, so
// it can take part in whichever kind of step the wave votes for.
struct Trav {
    int cur;          // interior record to test next (its byte offset in nodes, >= 0), or RT_NONE
    int sp;           // node stack: LDS address of the lane's top entry (trace_pixels)
    int lq;           // leaf queue: LDS address of the lane's top entry (trace_pixels)
    int k;            // triangle slot in progress (its byte offset in tri_geo), or -1
    uint32_t mask;    // mesh roots still to visit (scenes of up to 32 meshes; larger ones start at the top-level tree)
    float best_t;
    int best_rank;
};

// Root boxes of a scene of up to 32 meshes (LDS broadcast reads) -> bit mask of the meshes whose
// root box the ray's line touches.  A leaf root is tested as the reference does (:586); for an
// interior root the test is an exact shortcut: child boxes nest inside it and float
// subtract/multiply are monotonic, so no child test could pass when the root test fails.
template <bool STATS, int CULL, bool FINITE, bool ROCT>
__device__ __forceinline__ uint32_t root_mask_loop(const float4* s_roots, int jn, v3 org, v3 inv, float best_t, bool sah, Counters& cnt, int& nearest) {
    uint32_t mask = 0;
    // Roots are walked from the last to the first so that the mask is built as mask + mask + take (one add-with-carry instead of
    // select + or); "<=" against FLT_MAX then names the same root as "<" against +inf did walking upwards: the FIRST of the
    // nearest ones, none for an entry distance of +inf or NaN (the order is a heuristic only, but the oracle mirrors it for
    // its counters).
    float near_t = 0x1.fffffep127f;
    const float mi = sah_slack(org, inv), mo = sah_org(org), beyond = fma_(best_t, RT_CULL_REL, best_t) + RT_CULL_ABS;
    // One box per trip, not unrolled: the kernel is issue bound, so overlapping the LDS latency
    // of several boxes buys nothing, while two or three boxes in flight cost spilled registers
    // (0.8 GB of scratch writes per frame, measured).
    // the root's LDS address in a VECTOR register (the empty asm hides that it is uniform): address of the reads and value of
    // the `nearest` select at once - as a scalar it is copied to a vector register twice per trip
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(3))) f4v* lds_f4;
    uint32_t base = (uint32_t)(uintptr_t)(lds_f4)s_roots;
    if (ROCT && FINITE) {
        // the table exists once per direction octant (variant v = jn entries, v * jn * 32 bytes on; variant 0 is the plain one):
        // the lane reads the one of its ray, whose boxes are in (near, far) order per axis (box_test_oct)
        const uint32_t vb = (uint32_t)jn * 32u;
        base += (oct_axis_bit(inv.x) ? vb : 0u) + (oct_axis_bit(inv.y) ? 2u * vb : 0u) + (oct_axis_bit(inv.z) ? 4u * vb : 0u);
    }
    uint32_t at = base + (uint32_t)(jn - 1) * 32u, near_at = base - 32u;
    asm volatile("" : "+v"(at));
#pragma nounroll
    for (int j = jn - 1; j >= 0; j--) {
        const f4v lo_ = *(lds_f4)(uintptr_t)at, hi_ = *(lds_f4)(uintptr_t)(at + 16);
        const float4 lo = make_float4(lo_.x, lo_.y, lo_.z, lo_.w), hi = make_float4(hi_.x, hi_.y, hi_.z, hi_.w);
        float ta, tb;
        uint64_t m_take;
        if (!sah) m_take = lane_mask(FINITE ? box_test_finite(org, inv, lo, hi, ta, tb) : box_test(org, inv, lo, hi, ta, tb));
        else if (FINITE && ROCT) { box_test_oct(org, inv, lo, hi, ta, tb); m_take = lane_mask(sah_widened_hit(ta, tb, hi.w, mo, mi)); }
        else if (FINITE) { box_test_finite(org, inv, lo, hi, ta, tb); m_take = lane_mask(sah_widened_hit(ta, tb, hi.w, mo, mi)); }
        else m_take = lane_mask(sah_box_test_any(org, inv, lo, hi, hi.w, mo, ta, tb));
        // a mesh without triangles: its box is all NaN (k_pack_nodes), which the min/max form can never enter; the ladder can
        const bool real = (FINITE && !STATS) || __float_as_int(lo.w) != RT_NONE;
        if (!(FINITE && !STATS)) m_take &= lane_mask(real);
        if (CULL == 1) m_take &= lane_mask(!(tb < RT_CULL_BEHIND));
        if (CULL) m_take &= lane_mask(!(ta > beyond));
        if (STATS) cnt.box += real;
        mask = shl1_add(mask, m_take);
        const bool nearer = in_mask(m_take & lane_mask(ta <= near_t));
        near_at = nearer ? at : near_at;
        near_t = nearer ? ta : near_t;
        at -= 32;
    }
    nearest = (int)(near_at - base) >> 5;
    return mask;
}

template <bool STATS, int CULL, bool ROCT>
__device__ __forceinline__ uint32_t root_mask(int jn, const float4* s_roots, v3 org, v3 inv, float best_t, bool sah, Counters& cnt, int& nearest) {
    // wave-uniform choice of the slab-test form (see box_test_finite)
    const bool all_finite = lane_mask(!(__builtin_isfinite(inv.x) && __builtin_isfinite(inv.y) && __builtin_isfinite(inv.z))) == 0;
    return all_finite ? root_mask_loop<STATS, CULL, true, ROCT>(s_roots, jn, org, inv, best_t, sah, cnt, nearest)
                      : root_mask_loop<STATS, CULL, false, ROCT>(s_roots, jn, org, inv, best_t, sah, cnt, nearest);
}

// SAH structure: the reference only ever tests a triangle whose leaf box IN ITS OWN TREE the ray's
// slab test passes (the boxes above it nest and float subtract/multiply are monotonic, so they
// pass too).  The winner found through our tree is checked against that box (in its tri_hit
// record); in the rounding-level case (~1e-9 of queries) that it fails, the query is run again with
// the check applied to every hit (`gated`), so that the candidate set is exactly the reference's
// (DESIGN.md 5).
__device__ __forceinline__ bool ref_leaf_box_of(float4 h1, float4 h2, float4 h3, v3 org, v3 inv) {
    float ta, tb;
    return box_test(org, inv, make_float4(h1.w, h2.w, h3.x, 0.f), make_float4(h3.y, h3.z, h3.w, 0.f), ta, tb);
}
__device__ __forceinline__ bool ref_leaf_box_passes(const DevScene& S, int slot, v3 org, v3 inv) {
    const float4* h = (const float4*)((const char*)S.tri_hit + ((uint32_t)slot << 6));
    return ref_leaf_box_of(h[1], h[2], h[3], org, inv);
}

// Device memory reached through a pointer that was read back from the LDS copy of the kernel arguments: say so where
// it is free (fold_job), so that the compiler emits global_load/global_store instead of flat ones.  (In the path block
// itself the same qualification was measured at +4 % - the scheduler then orders that block's loads worse - and is not used.)
#define RT_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ RT_GLOBAL T* as_global(T* p) { return (RT_GLOBAL T*)p; }

// pixel coordinates of a lane's sample: x in the low half, y in the high half (images are <= 32768 a side)
__device__ __forceinline__ int px_of(uint32_t pxy) { return (int)(pxy & 0xffffu); }
__device__ __forceinline__ int py_of(uint32_t pxy) { return (int)(pxy >> 16); }

// Every sample of the job in `slot` has finished: fold its radiance into the accumulator, for each pixel the passes
// of the launch in order, with the reference's expressions (raytracer.fs.glsl:1008-1021: divide by RAYS_PER_PIXEL,
// running mean with weight 1/pass; RT_ACCUM_SUM adds instead).  Wave-uniform call; a lane takes one (pixel, channel)
// of the tile.  The weight of pass p is the same for every lane, so lane j computes the one of pass pb + j once per
// 64 passes and the loop reads it with v_readlane.  RT_FOLD_BATCH passes per trip: their loads are issued together and
// unconditionally (a lane without an item reads element 0, a pass beyond the last re-reads the last), then the
// recurrence runs over them in order.  (Inlined: as a called function it cost the caller one spilled register.)
__device__ __forceinline__ float* stage_of_wave(const TraceArgs& P, int wave) {
    return P.stage + (size_t)(blockIdx.x * (RT_BLOCK / 64) + wave) * 2 * P.job_cap * 3;
}

#ifndef RT_FOLD_BATCH
#define RT_FOLD_BATCH 4   // passes whose loads are in flight together (8 costs the path block a spilled register)
#endif
template <bool ONE_RAY, bool SUM>
__device__ __forceinline__ float fold_passes(float col, const RT_GLOBAL float* st, uint32_t idx, uint32_t stride, uint32_t n_passes, uint32_t first,
                                             uint32_t pass_stride, float rpp, int lane) {
    // st is wave-uniform, idx the lane's element of pass 0: st[idx + pass * stride] = one 32-bit add and one load per pass
    for (uint32_t pb = 0; pb < n_passes; pb += 64) {
        const float wl = 1.0f / (float)(first + (pb + (uint32_t)lane) * pass_stride);      // :1015, pass pb + lane
        const uint32_t pe = min(64u, n_passes - pb);
        for (uint32_t j0 = 0; j0 < pe; j0 += RT_FOLD_BATCH) {
            float tv[RT_FOLD_BATCH];
            const uint32_t o0 = (pb + j0) * stride;
            if (j0 + RT_FOLD_BATCH <= pe) {
#pragma unroll
                for (uint32_t u = 0; u < RT_FOLD_BATCH; u++) tv[u] = st[idx + (o0 + u * stride)];
            } else {
#pragma unroll
                for (uint32_t u = 0; u < RT_FOLD_BATCH; u++) tv[u] = st[idx + (pb + min(j0 + u, pe - 1)) * stride];
            }
#pragma unroll
            for (uint32_t u = 0; u < RT_FOLD_BATCH; u++) {
                if (j0 + u < pe) {                                                        // wave-uniform
                    const float total = ONE_RAY ? tv[u] : tv[u] / rpp;                   // :1009 (x / 1 is x)
                    if (SUM) col = col + total;
                    else {
                        const float w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wl), (int)(j0 + u))), om = 1.0f - w;
                        col = col * om + total * w;                                      // :1016-1021
                    }
                }
            }
        }
    }
    return col;
}

__device__ __forceinline__ void fold_job(const TraceArgs* Pp, const WaveJobs* wj, int slot, float* stage_wave) {
#ifdef RT_NO_FOLD   // timing experiment only: what the fold itself costs
    return;
#endif
#ifdef RT_FOLD_NEVER   // timing experiment only: the fold's code is there (registers, scheduling) but never runs
    if (Pp->max_bounces != 12345) return;
#endif
    const TraceArgs& P = *Pp;
    const int lane = threadIdx.x & 63;
    const int tw = 1 << P.tile_lx, npx = tw << P.tile_ly;
    const int ox = wj->ox[slot], oy = wj->oy[slot];
    const int W = P.cam.width, H = P.cam.height;
    const uint32_t n_passes = P.n_passes, first = P.first_pass, pstride = P.pass_stride;
    const bool one_ray = P.rays_per_pixel == 1, sum_mode = P.sum_mode != 0;
    const float rpp = (float)P.rays_per_pixel;
    RT_GLOBAL float* accum = as_global(P.accum);
    const RT_GLOBAL float* st = as_global(stage_wave) + (size_t)slot * P.job_cap * 3;
    for (int base = 0; base < 3 * npx; base += 64) {
        const int item = base + lane;
        const int lp = item / 3, ch = item - 3 * lp;
        const int x = ox + (lp & (tw - 1)), y = oy + (lp >> P.tile_lx);
        const bool live = item < 3 * npx && x < W && y < H;
        RT_GLOBAL float* acc = accum + ((size_t)(live ? y : 0) * W + (live ? x : 0)) * 3 + ch;
        float col = live ? *acc : 0.0f;
        const uint32_t idx = live ? (uint32_t)item : 0u;
        const uint32_t stride = (uint32_t)npx * 3u;                                      // floats between the passes of one item
        if (sum_mode) col = one_ray ? fold_passes<true, true>(col, st, idx, stride, n_passes, first, pstride, rpp, lane)
                                    : fold_passes<false, true>(col, st, idx, stride, n_passes, first, pstride, rpp, lane);
        else col = one_ray ? fold_passes<true, false>(col, st, idx, stride, n_passes, first, pstride, rpp, lane)
                           : fold_passes<false, false>(col, st, idx, stride, n_passes, first, pstride, rpp, lane);
        if (live) *acc = col;
    }
}

template <bool STATS, int CULL, bool FIRST_HIT, bool SAH, bool DEEP, bool MLDS, bool ROCT>
__device__ __forceinline__ void trace_pixels(const TraceArgs& a, float4* s_mem) {
    const DevScene& S = a.scene;
    // ---- LDS: mesh root table (2 float4 per mesh), then the per-lane columns
    // ROCT instantiations (a small SAH scene with a root table; chosen per launch in rt_hip.hip::launch_mats where the layout fits): the table in its eight octant variants
    // (box_test_oct; k_pack_nodes writes them, variant 0 first), paid for by keeping the rank -> slot table as 16-bit entries
    static_assert(!ROCT || (SAH && !DEEP && !FIRST_HIT && !STATS), "ROCT: render kernels of small SAH scenes only");
    float4* s_roots = s_mem;
    const int n_root4 = S.tlas_root >= 0 ? 0 : 2 * S.n_roots * (ROCT ? 8 : 1);
    for (int i = threadIdx.x; i < n_root4; i += RT_BLOCK) s_roots[i] = S.roots[i];
    // ... the mesh materials of the same scenes (80 B each): the path block reads the hit material after the winner's
    // record, a dependent memory round trip that an LDS copy shortens ...
    const int n_mat4 = S.tlas_root >= 0 ? 0 : 5 * S.n_meshes;
    float4* s_mats4 = s_mem + n_root4;
    for (int i = threadIdx.x; i < n_mat4; i += RT_BLOCK) s_mats4[i] = ((const float4*)S.mats)[i];
    // MLDS instantiations (a root-table scene without spheres: rt_hip.hip::launch_accel): every material the path block reads
    // is in the LDS copy, and the compiler knows it - ds_read instead of flat_load through a pointer that could be either
    // (16 of the path block's 28 flat loads; -1.0 % on the bench frame, profiles/r03_k_trace_isa.txt)
    const MatDev* s_mats = MLDS ? (const MatDev*)s_mats4 : (n_mat4 ? (const MatDev*)s_mats4 : S.mats);
    s_mem += n_mat4;
    // ... and, for a small SAH scene, the rank -> slot table: the first of the path block's dependent loads (winner's
    // rank -> slot -> its records -> its material), each a round trip to the L2 otherwise
    int* s_rank = (int*)(s_mem + n_root4);
    if (ROCT) for (int i = threadIdx.x; i < S.rank_lds; i += RT_BLOCK) ((uint16_t*)s_rank)[i] = (uint16_t)S.rank_slot[i];   // (<= 2048 slots)
    else for (int i = threadIdx.x; i < S.rank_lds; i += RT_BLOCK) s_rank[i] = S.rank_slot[i];
    // (making the LDS home of this table a compile-time fact like MLDS - ds_read instead of flat_load for the first hop of the
    // path block's dependent chain - measured +0.5 %: the block's loads are then scheduled worse; profiles/r03_k_trace_isa.txt)
    const int* rank_slot = S.rank_lds ? s_rank : S.rank_slot;
    // (ROCT: the same either-or pointer, read as 16-bit entries - the host only picks the instantiation when the table is in LDS)
    const auto slot_of_rank = [&](int r) { return ROCT ? (int)((const uint16_t*)rank_slot)[r] : rank_slot[r]; };
    s_mem += ROCT ? (S.rank_lds + 7) / 8 : (S.rank_lds + 3) / 4;
    // camera and atmosphere constants: read from LDS where the path block needs them, instead of
    // ~50 scalar registers held (and spilled) across the traversal loop
    PathConsts* s_pc = (PathConsts*)(s_mem + n_root4);
    if (threadIdx.x == 0) s_pc->a = a;
    const TraceArgs& P = s_pc->a;          // the LDS copy: use it wherever only the path block needs a value
    const DevScene& PS = P.scene;
    const DevCamera& cam = P.cam;
    const MatDev& atmo = P.atmo;
    // per-lane column: leaf queue entries [0, LQ), node stack entries [LQ, LQ + stack_depth);
    // entry e at s_stack[e * RT_BLOCK].  CULL builds keep the entry distance of each queued LEAF in a second
    // plane so that it can be re-culled when it is taken up (triangle tests per ray 4.6 -> 3.3).  Deferred interior
    // children are not re-culled: their own children are, one step later, and measured that is as good (box tests per
    // ray 22.5 -> 22.7) while the plane cost 2 bytes of LDS per entry and five instructions per pop (round 2).
    constexpr int LQ = RT_LQ_OF(SAH);
    const int n_ent = S.stack_depth + LQ;
    // (the leaf queue comes FIRST in a column: the node stack then starts at a compile-time offset from the same base
    // register, where "stack first" needs a second base register per plane for a run-time stack depth)
    // SAH + culling with the whole node stack in LDS: a queued leaf is ONE 8-byte entry (reference, entry distance as float32) -
    // one address and one LDS instruction per push / take-up instead of two of each (-0.5 % on the bench frame); 1.5 KB more
    // LDS per workgroup, which a scene that needs the DEEP instantiation rather spends on stack entries (dragon scene: +1.8 %
    // with two entries fewer in LDS), so those keep the 4 + 2-byte planes.  rt_hip.hip::trace_lds_bytes is the other half.
    constexpr bool LQ64 = RT_LQ64_OF(SAH, CULL, DEEP);
    // The node stack pointer and the leaf queue pointer are kept as the LDS ADDRESS of the lane's TOP entry (entry e of
    // lane t lies e planes of RT_BLOCK entries above the lane's first one; empty = one plane below the first): a push is one
    // add and a store, a pop a load and one subtract, both in place, and the emptiness tests compare against a uniform bound
    // (a lane's first entry lies less than one plane above the base of the columns; signed compares, the empty address of a
    // low column may be negative).  As an index each access costs a shift-add and a copy on top: five vector instructions
    // per node step, one per triangle step.  The leaf queue pointer runs over the 8-byte entries (LQ64), else over the 16-bit
    // distance plane of the culling builds (the 4-byte entry of the same index is then at twice the address plus a uniform
    // constant: one shift-add), else over the 4-byte entries themselves.
    typedef int i2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) int* lds_i;
    typedef __attribute__((address_space(3))) i2v* lds_i2;
    typedef __attribute__((address_space(3))) unsigned short* lds_h;
    constexpr bool LQ16 = !LQ64 && CULL != 0;
    constexpr uint32_t ST_PLANE = 4u * RT_BLOCK, LQ_PLANE = (LQ64 ? 8u : LQ16 ? 2u : 4u) * RT_BLOCK;
    const uint32_t col_base = (uint32_t)(uintptr_t)(lds_i)(int*)(s_mem + n_root4 + RT_PC_F4);        // uniform
    const uint32_t lq_base = LQ16 ? col_base + (uint32_t)n_ent * ST_PLANE : col_base;                  // plane the queue pointer runs over
    const uint32_t lq_int_k = col_base - 2u * lq_base;                                                // LQ16: 4-byte entry at 2 * pointer + this
    const uint32_t st_base = col_base + (LQ64 ? 8u : 4u) * LQ * RT_BLOCK;
    const uint32_t st_lim = st_base + (uint32_t)S.stack_depth * ST_PLANE;                              // DEEP: first entry beyond the LDS ones
    // DEEP: the planes go on in the workgroup's region of stack_ovf, so an entry beyond st_lim is at this uniform base plus
    // the same 32-bit address - no index arithmetic, the store / load takes the pointer register as its offset
    const uintptr_t ovf_base = DEEP ? (uintptr_t)a.stack_ovf + (size_t)blockIdx.x * a.ovf_block_bytes - st_lim : 0;
    const int st_empty = (int)(st_base + threadIdx.x * 4u) - (int)ST_PLANE;
    const int lq_empty = (int)(lq_base + threadIdx.x * (LQ_PLANE / RT_BLOCK)) - (int)LQ_PLANE;
    const auto lq_nonempty = [&](int lq) { return lq >= (int)lq_base; };
    const auto lq_room_for_two = [&](int lq) { return lq < (int)(lq_base + (LQ - 2) * LQ_PLANE); };
    const auto lq_store = [&](int lq, int ref, float dist) {          // the slot above the top entry
        const uint32_t at = (uint32_t)lq + LQ_PLANE;
        if (LQ64) *(lds_i2)(uintptr_t)at = (i2v){ref, __float_as_int(dist)};
        else if (LQ16) {
            *(lds_i)(uintptr_t)(2u * at + lq_int_k) = ref;
            *(lds_h)(uintptr_t)at = pack_dist(dist);
        } else *(lds_i)(uintptr_t)at = ref;
    };
    __syncthreads();

    const int lane = threadIdx.x & 63;
    Counters cnt = {0, 0, 0, 0};
    SchedStats sch = {0, 0, 0, 0, 0, 0};
    struct { unsigned long long n, band, behind; } aud = {0, 0, 0};   // culling audit (STATS builds of the exact mode)

    // ---- wave-uniform work cursor ---------------------------------------------------------
    // Work is handed out in jobs = one small tile of pixels with all passes of the launch (~512 samples),
    // taken from a launch-wide counter by whichever wave runs dry; a lane whose sample is finished
    // takes the next one of its wave's job (pass-major), so no lane waits for a slower neighbour and
    // the launch ends within one job's time on every SIMD.
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // uniform: say so, or every address derived from it sits in vector registers
    WaveJobs* wj = &s_pc->wj[wave];
    if (lane == 0) { wj->done[0] = wj->done[1] = 0; wj->expected[0] = wj->expected[1] = 0; }
    // (job shape and staging base are read from the LDS argument copy where the path block needs them: scalar
    // registers are as scarce in this kernel as vector ones)
    uint32_t w_left = 0;       // samples of the wave's job not handed out yet
    int w_x0 = 0, w_y0 = 0, w_slot = 0;
    bool w_dry = false;        // the launch's counter is exhausted

    // ---- per-lane state -----------------------------------------------------------------
    int mode = M_PATH;
    bool have = false;         // holds a sample: pixel pxy, staging element `elem` of its wave (slot * job_cap + pass * tile pixels + pixel of the tile)
    uint32_t pxy = 0, elem = 0;
    uint32_t sb = 0;           // low half: paths of the current sample started; high half: bounce of the path in flight
                               // (one register: the path block is where this kernel runs out of them; both <= 65535, rt_render checks)
    uint32_t rng = 0;
    bool fresh = true;         // no path in flight: the path block generates a camera sample
    v3 org = V3(0.f, 0.f, 0.f), dir = V3(0.f, 0.f, 1.f), inv = V3(0.f, 0.f, 0.f);
    v3 thr = V3(1.f, 1.f, 1.f), inc = V3(0.f, 0.f, 0.f);
    int prev_id = -1, prev_mat = -1;
    float ior = 1.0f;
    bool in_solid = false;
    int hb_key = 0;            // hitBehind test of the ray in flight: the mesh it is inside, or 0 (set at query start)
    constexpr bool sah = SAH;  // the scene was built with RT_ACCEL_SAH (the tri_hit records carry the reference's leaf boxes)
    bool gated = false;        // SAH: the query in flight is a repeat with the reference-leaf-box gate on every hit
    Trav tv;
    tv.cur = RT_NONE; tv.sp = st_empty; tv.lq = lq_empty; tv.k = -1; tv.mask = 0;
    tv.best_t = 1e20f; tv.best_rank = RT_MISS_RANK;
    // Two per-lane flags of the query in flight, kept as lane masks of the wave (64-bit scalars that are only ever
    // assigned where the wave's control flow is uniform, so that they stay in scalar registers): node work may remain;
    // 1/d has a non-finite component (the query takes the reference's compare ladder instead of the min/max form).
    uint64_t m_live = 0, m_nonfinite = 0;
    uint64_t m_gated = 0;      // SAH: lanes whose query in flight is a repeat with the reference-leaf-box gate on every hit (`gated`)
    // `mode` as the traversal loop sees it, as lane masks taken where the path block ends: lanes with a query in flight, lanes
    // waiting for the path block.  A lane whose query finishes inside the loop is marked in m_fin (scalar logic on masks the
    // vote computes anyway); its `mode` register is brought up to date once, when the loop is left - a compare and a select
    // fewer in every trip of the vote.
    uint64_t m_trav = 0, m_path = ~0ull, m_fin = 0, m_pw = ~0ull;
    uint64_t m_k = 0;          // lanes with a triangle in progress (tv.k >= 0): changes in triangle steps and at a query's start only

    while (true) {
        // =============== traversal phase: node steps / triangle steps chosen by vote ===============
        while (true) {
            const uint64_t bT = m_k | lane_mask(lq_nonempty(tv.lq));                    // has leaf work
            const uint64_t bN = m_live & lane_mask(lq_room_for_two(tv.lq));             // can take a node step
            const bool can_node = in_mask(bN);
            m_fin = m_trav & ~bT & ~m_live;                                             // the query is finished
            // The vote runs once per node step / triangle burst on the scalar unit, and a scalar instruction of this loop costs the
            // frame 1.5 x what a vector one does (profiles/r03_k_trace_isa.txt, calibration), so it is written for the scalar unit:
            // ONE way out of the loop - "fewer than refill_below lanes traverse and some lane waits for the path block", where "no
            // lane traverses" is the case 0 < refill_below (m_pw: all ones when no query is in flight at all, else the lanes waiting
            // since the last path block; with queries in flight and none left to step, m_fin is not empty) - as one select and one
            // compare; and without a bias a node step is taken when cN > cT (lanes exist, so cT == 0 implies cN > 0).
            const int cAny = lanes_in(bN | bT), cGo = (m_fin | m_pw) != 0 ? cAny : 64;
            if (cGo < a.refill_below) break;
            const int cN = lanes_in(bN), cT = lanes_in(bT);
#ifdef RT_NODE_BIAS_KNOB    // sweeps only (tools/sweep_sched.sh; measured flat within +-8 lanes, worse beyond): a step never runs without lanes
            const bool do_node = (cN ? cN + a.node_bias : -(1 << 20)) > (cT ? cT : -(1 << 30));
#else
            const bool do_node = cN > cT;
#endif
            if (STATS && do_node) { sch.node_steps++; sch.node_lanes += cN; }
            if (do_node) {
                if (can_node) {
                    // A lane that can take a node step HOLDS a record: when a step leaves it without a child to go on with, it
                    // takes the next one up at once (end of this block).  So "node work remains" (m_live) is just "a record is in
                    // hand" - one compare per step where the step used to check the record, the stack and the root mask (-1.6 %).
                    {                                               // :562-584, both children of `cur`
                        const float4* n = (const float4*)((const char*)S.nodes + (uint32_t)tv.cur);
                        const float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
                        const int ref1 = __float_as_int(n0.w), ref2 = __float_as_int(n1.w);
                        float t1a, t1b, t2a, t2b;
                        // Which child is hit / culled / a leaf / the nearer one is kept as LANE MASKS (64-bit scalars, one
                        // ballot per compare) and combined on the scalar unit: as booleans the compiler merges them across
                        // the branches below as 0/1 values in vector registers and compares them again - a dozen vector
                        // instructions of a step that is bound by vector issue.
                        uint64_t m_h1, m_h2;
                        // An absent child (reference RT_NONE: a node with one child, an empty leaf) carries an all-NaN box and slack
                        // (k_pack_nodes): in the min/max form every slab value, hence the interval, is NaN, and NaN <= x is false -
                        // the test fails by itself, no compare of the reference needed (two vector instructions per node step: -0.9 %).
                        // The compare ladder reads NaN as "not disjoint" (as the reference does), so those branches do test it.
                        // (m_nonfinite holds the lanes with a query in flight only, so the whole mask is tested: a compare and a branch;
                        // a lane outside this step can send the step down the ladder, which is exact for every ray.  The hot path
                        // falls through: two taken branches fewer.)
                        if (__builtin_expect(m_nonfinite == 0, 1)) {
                            box_test_finite(org, inv, n0, n1, t1a, t1b);
                            box_test_finite(org, inv, n2, n3, t2a, t2b);
                            if (sah) {
                                const float mi = sah_slack(org, inv), mo = sah_org(org);
                                m_h1 = lane_mask(t1a <= fma_(n2.w + mo, mi, t1b));       // sah_widened_hit
                                m_h2 = lane_mask(t2a <= fma_(n3.w + mo, mi, t2b));
                            } else {
                                m_h1 = lane_mask(t1a <= t1b);
                                m_h2 = lane_mask(t2a <= t2b);
                            }
                        } else if (!sah) {                          // a lane has an exactly zero direction component
                            m_h1 = lane_mask(box_test(org, inv, n0, n1, t1a, t1b)) & lane_mask(ref1 != RT_NONE);
                            m_h2 = lane_mask(box_test(org, inv, n2, n3, t2a, t2b)) & lane_mask(ref2 != RT_NONE);
                        } else {
                            const float mo = sah_org(org);
                            m_h1 = lane_mask(sah_box_test_any(org, inv, n0, n1, n2.w, mo, t1a, t1b)) & lane_mask(ref1 != RT_NONE);
                            m_h2 = lane_mask(sah_box_test_any(org, inv, n2, n3, n3.w, mo, t2a, t2b)) & lane_mask(ref2 != RT_NONE);
                        }
                        if (CULL) {                                 // culled(): NaN never culls
                            const float beyond = fma_(tv.best_t, RT_CULL_REL, tv.best_t) + RT_CULL_ABS;
                            if (CULL == 1) {
                                m_h1 &= ~(lane_mask(t1b < RT_CULL_BEHIND) | lane_mask(t1a > beyond));
                                m_h2 &= ~(lane_mask(t2b < RT_CULL_BEHIND) | lane_mask(t2a > beyond));
                            } else {                                // the band alone (rt_set_cull(ctx, RT_CULL_BAND))
                                m_h1 &= ~lane_mask(t1a > beyond);
                                m_h2 &= ~lane_mask(t2a > beyond);
                            }
                        }
                        if (STATS) cnt.box += (ref1 != RT_NONE) + (ref2 != RT_NONE);
                        // Far child first, so that the near one is taken up first.  A child that is missed or culled becomes "no
                        // record" BEFORE the two are ordered, and what each ordered child is - a leaf (reference above 0x80000000
                        // as unsigned), an interior record (>= 0) or nothing - is read off its reference with one compare: the
                        // same decisions as lane-mask algebra on (hit, leaf, nearer) cost fourteen scalar instructions per step,
                        // and a scalar instruction of this loop costs more than a vector one (profiles/r03_k_trace_isa.txt).
                        const bool second_near = t2a < t1a;
                        const int r1 = in_mask(m_h1) ? ref1 : RT_NONE, r2 = in_mask(m_h2) ? ref2 : RT_NONE;
                        const int rn = second_near ? r2 : r1, rf = second_near ? r1 : r2;
                        const float tn = second_near ? t2a : t1a, tf = second_near ? t1a : t2a;
                        // (the entry is written whether the child is a leaf or not - a lane takes part in a node step only with two
                        // free slots above its top entry - and only the pointer moves conditionally: no divergent region, i.e. three
                        // scalar instructions fewer per push for one more vector one)
                        lq_store(tv.lq, rf, tf);
                        tv.lq += (uint32_t)rf > (uint32_t)RT_NONE ? (int)LQ_PLANE : 0;
                        lq_store(tv.lq, rn, tn);
                        tv.lq += (uint32_t)rn > (uint32_t)RT_NONE ? (int)LQ_PLANE : 0;
                        if ((rn | rf) >= 0) {                       // both interior: the far one waits on the node stack
                            if (!DEEP || tv.sp + (int)ST_PLANE < (int)st_lim) *(lds_i)(uintptr_t)((uint32_t)tv.sp + ST_PLANE) = rf;
                            else *(int*)(ovf_base + (uint32_t)(tv.sp + (int)ST_PLANE)) = rf;
                            bump_in_place<ST_PLANE>(tv.sp);
                        }
                        tv.cur = rn >= 0 ? rn : (rf >= 0 ? rf : RT_NONE);
                        // ---- no child to go on with: take up the next record NOW (a deferred far child, else the next mesh root),
                        // so that between steps "node work remains" is simply "a record is in hand" ----
                        if (tv.cur == RT_NONE) {
                            if (tv.sp >= (int)st_base) {
                                if (!DEEP || tv.sp < (int)st_lim) tv.cur = *(lds_i)(uintptr_t)(uint32_t)tv.sp;
                                else                                    // a rare deep entry (see stack_ovf)
                                    tv.cur = *(const int*)(ovf_base + (uint32_t)tv.sp);
                                tv.sp -= (int)ST_PLANE;
                            } else if (tv.mask) {                       // (a root in the table is never a leaf: rt_set_scene wraps those)
                                const int b = __builtin_ctz(tv.mask);
                                tv.mask &= tv.mask - 1;
                                tv.cur = __float_as_int(s_roots[2 * b].w);
                            }
                        }
                        cal_valu<RT_CAL_NODE_VALU>();
                        cal_salu<RT_CAL_NODE_SALU>();
                    }
                }
                m_live &= ~(bN & lane_mask(tv.cur == RT_NONE));
            } else {
                // triangle burst: stay on triangles while at least half of the lanes that were
                // traversing at the vote still have leaf work (an upper bound of the lanes a node
                // step could serve then outnumbers them and the full vote is taken again)
                const int lim = max(a.burst8 * (cN + cT), 1);           // (>= 1: a step never runs without lanes)
                int cTn;
                uint64_t m_leaf = bT;                                   // lanes with leaf work: the vote's mask, then each step's own
                do {
                    if (STATS) { sch.tri_steps++; sch.tri_lanes += __popcll(m_leaf); }
                    {
                        // (no region around the lanes with leaf work: the two conditions below are false for every other lane - m_leaf is
                        // "a triangle in progress or a queued leaf" - and a region costs three scalar instructions per step)
                        if (in_mask(m_leaf & ~m_k)) {                   // take up the most recently found leaf
                            const float beyond = fma_(tv.best_t, RT_CULL_REL, tv.best_t) + RT_CULL_ABS;
                            if (LQ64) {
                                const i2v e = *(lds_i2)(uintptr_t)(uint32_t)tv.lq;
                                tv.k = RT_LEAF_OFFSET(e.x);
                                if (__int_as_float(e.y) > beyond) tv.k = -1;
                            } else if (LQ16) {
                                tv.k = RT_LEAF_OFFSET(*(lds_i)(uintptr_t)(2u * (uint32_t)tv.lq + lq_int_k));
                                if (unpack_dist(*(lds_h)(uintptr_t)(uint32_t)tv.lq) > beyond) tv.k = -1;
                            } else tv.k = RT_LEAF_OFFSET(*(lds_i)(uintptr_t)(uint32_t)tv.lq);
                            tv.lq -= (int)LQ_PLANE;
                        }
                        if (tv.k >= 0) {                                // one triangle of the leaf (:593-630)
                            // tv.k is the slot's 32-bit byte offset from a uniform base: no shift, no 64-bit address arithmetic
                            const float4* g = (const float4*)((const char*)S.tri_geo + (uint32_t)tv.k);
                            const float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
                            // all rejects are folded into one predicate and the three edge tests run together:
                            // one divergent region instead of five nested ones (scalar-unit pressure)
                            const int rank_k = __float_as_int(g3.z);       // rank in the reference's traversal order
                            const bool tested = rank_k != prev_id;
                            if (STATS) cnt.tri += tested;
                            // hitBehind for this triangle's mesh (:605-615; in a solid: same mesh only [9.6])
                            // = in_solid ? mesh == prev_mesh : !transparentFromBehind, through the per-ray key/mask
                            const bool hb = ((__float_as_int(g3.w) ^ hb_key) & (in_solid ? ~RT_TFB_BIT : RT_TFB_BIT)) == 0;
                            const v3 A = V3(g0.x, g0.y, g0.z), B = V3(g1.x, g1.y, g1.z), C = V3(g2.x, g2.y, g2.z);
                            const v3 N = V3(g0.w, g1.w, g2.w);
                            const float nd = dot3(N, dir);
                            const float t = -(dot3(N, org) + g3.x) / nd;
                            bool ok = tested && !(__builtin_fabsf(nd) < 1e-6f) && !(nd > 0.0f && !hb);
                            ok = ok && !(t < 1e-6f);                        // covers the reference's `t < 0` reject too
                            // closer than (or tied with a later-ranked) current best; NaN t never wins
                            // (as lane masks: the short-circuit form is a divergent region of its own, seven scalar instructions)
                            ok = ok && in_mask(lane_mask(t < tv.best_t) | (lane_mask(t == tv.best_t) & lane_mask(rank_k < tv.best_rank)));
                            if (ok) {
                                const v3 P = madd3(org, t, dir);
                                const float e0 = dot3(N, cross3(B - A, P - A));
                                const float e1 = dot3(N, cross3(C - B, P - B));
                                const float e2 = dot3(N, cross3(A - C, P - C));
                                bool inside = !(e0 < 0.0f) && !(e1 < 0.0f) && !(e2 < 0.0f);
                                // SAH structure, a query that is being repeated (see the path block): a hit
                                // counts only if the ray passes the box of the triangle's leaf in the
                                // REFERENCE's tree
                                if (SAH && m_gated != 0) {                  // (wave-level: rare, one compare and one branch)
                                    if (in_mask(m_gated & lane_mask(inside))) inside = ref_leaf_box_passes(S, (int)((uint32_t)tv.k >> 6), org, inv);
                                }
                                tv.best_t = inside ? t : tv.best_t;
                                tv.best_rank = inside ? rank_k : tv.best_rank;
                            }
                            tv.k = __float_as_int(g3.y);                    // the leaf's next slot, or -1
                            cal_valu<RT_CAL_TRI_VALU>();
                        }
                    }
                    m_k = lane_mask(tv.k >= 0);
                    m_leaf = m_k | lane_mask(lq_nonempty(tv.lq));
                    cTn = lanes_in(m_leaf);
                } while (8 * cTn >= lim);
            }
        }
        if (in_mask(m_fin)) mode = M_PATH;
        if (lane_mask(mode == M_PATH) == 0) break;                  // every lane is M_DONE (a lane in M_WAIT implies samples in flight, i.e. lanes that traverse)

        // =============== path phase: shade the finished query, start the next one ===============
        __builtin_amdgcn_s_setprio(RT_PRIO_PATH);
        if (STATS) { sch.path_steps++; sch.path_lanes += __popcll(lane_mask(mode == M_PATH)); }
        bool need = mode == M_WAIT, start = false, redo = false, completed = false;
        if (mode == M_PATH) {
            if (!fresh) {
                if (FIRST_HIT) {
                    int id = -1;
                    if (tv.best_rank != RT_MISS_RANK) {
                        if (tv.best_rank < 0) id = -2 - (tv.best_rank - RT_SPHERE_RANK);
                        else {
                            const int slot = SAH ? slot_of_rank(tv.best_rank) : tv.best_rank;   // (reference structure: slots are in rank order)
                            id = PS.slot_id[slot];
                            if (SAH && !gated) redo = !ref_leaf_box_passes(PS, slot, org, inv) || P.recheck_all != 0;
                        }
                    }
                    if (redo) start = true;
                    else {
                        const size_t pix = (size_t)py_of(pxy) * cam.width + px_of(pxy);
                        P.ids[pix] = id;
                        P.ts[pix] = tv.best_t;
                        have = false;
                        fresh = true;
                        sb = 0;
                    }
                } else if (tv.best_rank == RT_MISS_RANK) {         // miss: environment light is vec3(0) (:279-282)
                    fresh = true;
                } else {
                    // ---- rebuild the HitInfo of the winner ----
                    const float best_t = tv.best_t;
                    const v3 hit_point = madd3(org, best_t, dir);
                    v3 normal;
                    bool backside = false;
                    int hit_id = -1, hit_mat;
                    if (tv.best_rank < 0) {
                        const int si = tv.best_rank - RT_SPHERE_RANK;
                        const SphereDev& sph = PS.spheres[si];
                        normal = normalize3(hit_point - V3(sph.position[0], sph.position[1], sph.position[2]));
                        hit_mat = 256 + si;
                    } else {
                        const int slot = SAH ? slot_of_rank(tv.best_rank) : tv.best_rank;
                        const float4* g = S.tri_geo + 4 * (size_t)slot;
                        const float4 g0 = g[0], g1 = g[1], g2 = g[2];
                        const v3 A = V3(g0.x, g0.y, g0.z), B = V3(g1.x, g1.y, g1.z), C = V3(g2.x, g2.y, g2.z);
                        const v3 N = V3(g0.w, g1.w, g2.w);
                        backside = dot3(N, dir) > 0.0f;
                        // (:364-379 divide half the cross products' lengths by the triangle's area, itself half a length: the halvings
                        // are exact - a length is 0 or >= sqrt of the smallest subnormal, ~3.7e-23 - so the quotient of the halves is
                        // the quotient of the lengths, same rounding; three multiplies fewer)
                        const float len_n = length3(N);
                        const float w = length3(cross3(B - A, hit_point - A)) / len_n;
                        const float u = length3(cross3(C - B, hit_point - B)) / len_n;
                        const float v = 1.0f - u - w;
                        const float4* nn = PS.tri_hit + 4 * (size_t)slot;
                        const float4 na = nn[0], nb = nn[1], nc = nn[2];
                        // the reference's leaf box sits in the same 64-byte record as the normals: no extra memory round trip, no extra line
                        if (SAH && !gated) redo = !ref_leaf_box_of(nb, nc, nn[3], org, inv) || P.recheck_all != 0;
                        normal = V3(na.x * u + nb.x * v + nc.x * w, na.y * u + nb.y * v + nc.y * w,
                                    na.z * u + nb.z * v + nc.z * w);
                        hit_id = tv.best_rank;                       // (= the record's rank field g3.z, without waiting for it)
                        hit_mat = __float_as_int(na.w);
                    }
                    if (STATS) cnt.hits += !redo;
                    if (STATS && !CULL && !redo && tv.best_rank >= 0) {
                        // Culling audit (counting launches of the exact mode only; the oracle's rto_cull_audit): could a culling
                        // rule skip the WINNER's own bounding box, with the winner's own distance as the best in hand?  That box
                        // lies inside every box any structure over the same triangles puts the triangle in, and float
                        // subtract / multiply are monotonic, so its interval lies inside theirs: no flagged query means no
                        // culling traversal of either structure can lose the reference's winner on this launch's rays.
                        const int slot = SAH ? slot_of_rank(tv.best_rank) : tv.best_rank;
                        const float4* g = S.tri_geo + 4 * (size_t)slot;
                        const float4 a0 = g[0], a1 = g[1], a2 = g[2];
                        const float4 lo = make_float4(__builtin_fminf(a0.x, __builtin_fminf(a1.x, a2.x)), __builtin_fminf(a0.y, __builtin_fminf(a1.y, a2.y)),
                                                      __builtin_fminf(a0.z, __builtin_fminf(a1.z, a2.z)), 0.f);
                        const float4 hi = make_float4(__builtin_fmaxf(a0.x, __builtin_fmaxf(a1.x, a2.x)), __builtin_fmaxf(a0.y, __builtin_fmaxf(a1.y, a2.y)),
                                                      __builtin_fmaxf(a0.z, __builtin_fmaxf(a1.z, a2.z)), 0.f);
                        float own_lo, own_hi;
                        box_test(org, inv, lo, hi, own_lo, own_hi);      // the ladder's interval, hit or not
                        aud.n++;
                        aud.band += own_lo > fma_(best_t, RT_CULL_REL, best_t) + RT_CULL_ABS;
                        aud.behind += own_hi < RT_CULL_BEHIND;
                    }
                    const MatDev* m = (MLDS || hit_mat < 256) ? &s_mats[hit_mat] : &PS.spheres[hit_mat - 256].material;
                    const float alpha = m->color[3];
                    bool ended = false, passthrough = redo;   // a query to be repeated changes nothing here
                    if (!redo && alpha < 0.99999f) {                        // :725-736
                        if (random_value(rng) > alpha) {
                            passthrough = true;
                            if ((sb >> 16) == (uint32_t)P.max_bounces) ended = true;
                            else { org = hit_point; prev_id = hit_id; prev_mat = hit_mat; }
                        }
                    }
                    if (!passthrough) {
                        const float es = m->emissionStrength;
                        inc = inc + V3(m->emissionColor[0] * es, m->emissionColor[1] * es, m->emissionColor[2] * es) * thr;
                        const float nf = backside ? -1.0f : 1.0f;
                        const v3 nrm = normal * nf;
                        const float n2 = (in_solid && backside) ? atmo.ior : m->ior;
                        const float n1 = ior;
                        v3 diffuse = normalize3((-nrm) + random_direction(rng));
                        const float costheta = clampf(dot3(normalize3(nrm), normalize3(-dir)), -1.0f, 1.0f);
                        float r0 = (n1 - n2) / (n1 + n2);
                        r0 = r0 * r0;
                        const float om = 1.0f - costheta;
                        const float om5 = ((om * om) * (om * om)) * om;
                        const float schlick = r0 + (1.0f - r0) * om5;
                        bool specular = random_value(rng) < m->specularStrength;
                        v3 spec_dir = V3(0.0f, 0.0f, 0.0f);
                        if (specular) {
                            spec_dir = reflect3(dir, nrm);
                            diffuse = -diffuse;
                        } else if (random_value(rng) > schlick) {
                            if (random_value(rng) < m->transmission) {   // :822-847
                                float pc0 = atmo.color[0], pc1 = atmo.color[1], pc2 = atmo.color[2];
                                float pc3 = atmo.color[3], ptr = atmo.transmission;
                                if (prev_mat >= 0) {
                                    const MatDev* pm = (MLDS || prev_mat < 256) ? &s_mats[prev_mat] : &PS.spheres[prev_mat - 256].material;
                                    pc0 = pm->color[0]; pc1 = pm->color[1]; pc2 = pm->color[2]; pc3 = pm->color[3];
                                    ptr = pm->transmission;
                                }
                                const float at = -det_log(ptr);
                                const float e = det_exp(-at * best_t);
                                const v3 tc = V3(pc0, pc1, pc2) * e;
                                thr = thr * (tc * pc3);
                                if (backside) { in_solid = false; ior = atmo.ior; }
                                else { in_solid = true; ior = m->ior; }
                            } else {
                                diffuse = -diffuse;
                                if (backside) { in_solid = true; ior = m->ior; }
                            }
                        } else {
                            specular = true;
                            spec_dir = reflect3(dir, nrm);
                            diffuse = -diffuse;
                        }
                        if ((sb >> 16) == (uint32_t)P.max_bounces) ended = true;
                        else {
                            org = hit_point;
                            if (specular) {
                                const float sa = m->smoothness, sd = 1.0f - sa;
                                dir = V3(diffuse.x * sd + spec_dir.x * sa, diffuse.y * sd + spec_dir.y * sa,
                                         diffuse.z * sd + spec_dir.z * sa);
                                thr = thr * V3(m->specularColor[0], m->specularColor[1], m->specularColor[2]);
                            } else {
                                dir = diffuse;
                                thr = thr * V3(m->color[0], m->color[1], m->color[2]);
                            }
                            prev_id = hit_id;
                            prev_mat = hit_mat;
                        }
                    }
                    if (ended) fresh = true;
                    else { sb += redo ? 0u : 0x10000u; start = true; }
                }
                if (fresh && !FIRST_HIT) {                         // the path is over (:989): totalIncomingLight += inc
                    // the sample's running sum lives in the staging slot of its job (three fewer live registers):
                    // element (pass, pixel of the tile); fold_job divides by RAYS_PER_PIXEL (:1009)
                    float* o = stage_of_wave(P, wave) + (size_t)elem * 3;
                    const v3 sum = (sb & 0xffffu) == 1u ? V3(0.f, 0.f, 0.f) : V3(o[0], o[1], o[2]);
                    o[0] = sum.x + inc.x; o[1] = sum.y + inc.y; o[2] = sum.z + inc.z;
                    if ((sb & 0xffffu) == (uint32_t)P.rays_per_pixel) { sb = 0; have = false; completed = true; }   // the sample is complete (:1008-1010)
                }
            }
            need = fresh && !have;
        }
        if (!FIRST_HIT) {                                          // samples finished in this block, per job slot (wave-uniform)
            // (one ballot of the flag, split by a compare: the ballot of a boolean that was merged across branches costs a select
            // and a compare, that of a compare only the compare)
            const uint64_t c_all = lane_mask(completed), c_lo = lane_mask(elem < P.job_cap);
            const uint64_t c0 = c_all & c_lo, c1 = c_all & ~c_lo;
            if (lane == 0) {
                if (c0) wj->done[0] = (uint16_t)(wj->done[0] + __popcll(c0));
                if (c1) wj->done[1] = (uint16_t)(wj->done[1] + __popcll(c1));
            }
        }
        // ---- hand out samples: wave-uniform code, reached by every lane of the wave ----
        for (uint64_t nm = lane_mask(need); nm != 0; nm = lane_mask(need)) {
            if (w_left == 0) {                                     // the job in hand is handed out completely
                int free_slot = 0;
                if (!FIRST_HIT) {
                    // a job whose samples have all finished is folded into the accumulator and its slot given back
                    free_slot = -1;
#pragma nounroll
                    for (int k = 1; k >= 0; k--) {
                        const uint32_t expc = (uint32_t)__builtin_amdgcn_readfirstlane((int)wj->expected[k]);
                        if (expc && (uint32_t)__builtin_amdgcn_readfirstlane((int)wj->done[k]) == expc) {
                            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the staged radiance was written by other lanes of this wave
                            fold_job(&P, wj, k, stage_of_wave(P, wave));
                            if (lane == 0) wj->expected[k] = 0;
                            free_slot = k;
                        } else if (!expc) free_slot = k;
                    }
                }
                if (w_dry) { if (need) mode = M_DONE; break; }
                if (free_slot < 0) { if (need) mode = M_WAIT; break; }   // both jobs still have samples in flight: ask again later
                unsigned int c = 0;
                if (lane == 0) c = atomicAdd(P.work_counter, 1u);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c >= (unsigned int)__builtin_amdgcn_readfirstlane((int)P.n_jobs)) { w_dry = true; continue; }
                const uint32_t n_tx = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.n_tx);
                const uint32_t tile_lx = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.tile_lx), tile_ly = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.tile_ly);
                const uint32_t ty = c / n_tx;
                w_x0 = (int)((c - ty * n_tx) << tile_lx);
                w_y0 = (int)(ty << tile_ly);
                w_slot = free_slot;
                w_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)P.n_passes) << (tile_lx + tile_ly);
                if (!FIRST_HIT && lane == 0) {
                    // samples that will be traced: the tile's pixels inside the image x all passes
                    const uint32_t cw = (uint32_t)min(1 << tile_lx, cam.width - w_x0), chh = (uint32_t)min(1 << tile_ly, cam.height - w_y0);
                    wj->ox[free_slot] = (uint16_t)w_x0;
                    wj->oy[free_slot] = (uint16_t)w_y0;
                    wj->expected[free_slot] = (uint16_t)(cw * chh * P.n_passes);     // <= 1024 (RT_MAX_LAUNCH_PASSES, job shape)
                    wj->done[free_slot] = 0;
                }
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nm, 0u));
            const uint32_t avail = w_left;
            if (need && rank < avail) {
                const uint32_t tile_lx = P.tile_lx, job_px_log = tile_lx + P.tile_ly;
                const uint32_t o = (P.n_passes << job_px_log) - w_left + rank;   // pass-major: (pass, pixel of the tile)
                const uint32_t lp = o & ((1u << job_px_log) - 1u);
                const int x = w_x0 + (int)(lp & ((1u << tile_lx) - 1u));
                const int y = w_y0 + (int)(lp >> tile_lx);
                need = false;                                      // consumed, inside the image or not
                if (x < cam.width && y < cam.height) {
                    have = true;
                    pxy = (uint32_t)x | ((uint32_t)y << 16);
                    elem = o + (w_slot ? P.job_cap : 0u);
                    // per-pass seed of the pixel (:956-957); the state runs on across the RAYS_PER_PIXEL samples
                    rng = (uint32_t)y * (uint32_t)cam.width + (uint32_t)x + (P.first_pass + (o >> job_px_log) * P.pass_stride) * 747796405u;
                    mode = M_PATH;                                 // (a lane that was waiting for a job slot)
                }
            }
            w_left -= min((uint32_t)__popcll(nm), avail);
        }
        // a lane that consumed a sample outside the image, or none at all, has nothing to start in this block
        if (mode == M_WAIT && have) mode = M_PATH;
        if (mode == M_PATH) {
            if (fresh && have) {
                {
                    const v3 view_point = view_point_of(cam, px_of(pxy), py_of(pxy));
                    // camera sample (:972-987); four draws: DoF x,y then AA x,y
                    const float r1 = random_value(rng), r2 = random_value(rng), r3 = random_value(rng), r4 = random_value(rng);
                    const float nearp = cam.view_params[2];
                    const v3 dofo = V3(((r1 * 2.0f - 1.0f) * cam.dof) * nearp, ((r2 * 2.0f - 1.0f) * cam.dof) * nearp,
                                       ((0.0f * 2.0f - 1.0f) * cam.dof) * nearp);
                    const v3 aao = V3((r3 * 2.0f - 1.0f) * cam.aa, (r4 * 2.0f - 1.0f) * cam.aa, (0.0f * 2.0f - 1.0f) * cam.aa);
                    dir = normalize3((view_point + dofo) + aao);
                    org = V3(cam.cam_pos[0], cam.cam_pos[1], cam.cam_pos[2]) - dofo;
                    prev_id = -1;
                    prev_mat = -1;
                    ior = atmo.ior;
                    in_solid = false;
                    thr = V3(1.f, 1.f, 1.f);
                    inc = V3(0.f, 0.f, 0.f);
                    sb = (sb & 0xffffu) + 1u;                            // bounce = 0, one more path of the sample
                    fresh = false;
                    start = true;
                }
            }
            if (start) {                                           // begin one closest-hit query (:518-541)
                cal_valu<RT_CAL_PATH_VALU>();
                if (STATS) cnt.rays += !redo;
                gated = redo;
                inv = V3(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
                tv.best_t = 1e20f;
                tv.best_rank = RT_MISS_RANK;
                // in a solid only the mesh the ray is inside shows its back faces (prev_mat = -1 or a
                // sphere never equals a mesh index); outside, every mesh that is not transparentFromBehind
                hb_key = in_solid ? prev_mat : 0;
                const int n_sph = __builtin_amdgcn_readfirstlane(PS.n_spheres);
                for (int i = 0; i < n_sph; i++) {
                    const SphereDev& sph = PS.spheres[i];
                    const v3 c = V3(sph.position[0], sph.position[1], sph.position[2]);
                    const v3 oo = org - c;
                    const float qa = dot3(dir, dir);
                    const float qb = 2.0f * dot3(oo, dir);
                    const float qc = dot3(oo, oo) - sph.radius * sph.radius;
                    const float disc = qb * qb - (4.0f * qa) * qc;
                    if (disc >= 0.0f) {
                        const float dst = (-qb - __builtin_sqrtf(disc)) / (2.0f * qa);
                        if (dst >= 0.0f && !(dst < 1e-6f) && dst < tv.best_t) { tv.best_t = dst; tv.best_rank = RT_SPHERE_RANK + i; }
                    }
                }
                tv.sp = st_empty;
                tv.lq = lq_empty;
                tv.k = -1;
                const int tlas = __builtin_amdgcn_readfirstlane(PS.tlas_root);
                if (tlas >= 0) {                                   // many meshes: start at the top-level tree
                    tv.cur = tlas;
                    tv.mask = 0;
                } else {
                    tv.cur = RT_NONE;
                    int nearest;
                    tv.mask = root_mask<STATS, CULL, ROCT>(__builtin_amdgcn_readfirstlane(PS.n_roots), s_roots, org, inv, tv.best_t, sah, cnt, nearest);
                    // the mesh whose root box the ray enters first is visited first (a hit there culls the others)
                    // (entry distances of +inf / NaN leave `nearest` unset: then the first root of the mask; roots are interior records)
                    if (tv.mask) {
                        const int b = nearest >= 0 ? nearest : __builtin_ctz(tv.mask);
                        tv.cur = __float_as_int(s_roots[2 * b].w);
                        tv.mask &= ~(1u << b);
                    }
                }
                mode = M_TRAV;
            }
        }
        __builtin_amdgcn_s_setprio(RT_PRIO_TRAV);
        // The wave is converged here: take the two lane masks anew, for the queries that have just started and (to the
        // same values: node work remains <=> a record in hand, a deferred one or a root left) for those still in flight.
        // (a lane without a query keeps the 1/d of its last one: masked out of m_nonfinite)
        m_trav = lane_mask(mode == M_TRAV);
        m_k = lane_mask(tv.k >= 0);
        m_path = lane_mask(mode == M_PATH);
        m_pw = m_trav ? m_path : ~0ull;
        m_live = m_trav & lane_mask(tv.cur != RT_NONE);
        m_gated = SAH ? lane_mask(gated) : 0;
        m_nonfinite = m_trav & ~(lane_mask(__builtin_isfinite(inv.x)) & lane_mask(__builtin_isfinite(inv.y)) & lane_mask(__builtin_isfinite(inv.z)));
    }
    if (!FIRST_HIT) {
        // every job was folded when its last sample finished (the lane that finishes it asks for the next sample in
        // the same block, which is where finished jobs are folded); a slot still in use here would be a bookkeeping
        // error: rt_get_stats reports it
        if (lane == 0 && (wj->expected[0] | wj->expected[1]) != 0) atomicAdd(a.counters + 15, 1ull);
    }
    if (STATS) {
        unsigned long long v[4] = {cnt.rays, cnt.box, cnt.tri, cnt.hits};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned long long x = v[q];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if ((threadIdx.x & 63) == 0 && x) atomicAdd(a.counters + q, x);
        }
        if (!CULL && !FIRST_HIT) {
            unsigned long long u[3] = {aud.n, aud.band, aud.behind};
#pragma unroll
            for (int q = 0; q < 3; q++) {
                unsigned long long x = u[q];
                for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
                if ((threadIdx.x & 63) == 0 && x) atomicAdd(a.counters + 10 + q, x);
            }
        }
        if ((threadIdx.x & 63) == 0) {
            const unsigned long long w[6] = {sch.node_steps, sch.node_lanes, sch.tri_steps, sch.tri_lanes, sch.path_steps, sch.path_lanes};
#pragma unroll
            for (int q = 0; q < 6; q++) atomicAdd(a.counters + 4 + q, w[q]);
        }
    }
}
