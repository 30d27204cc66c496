/* rt_hip.h - C-ABI of the MI355X rendering core (librt_hip.so).
 *
 * This is the drop-in boundary for the ONE hot path of dw-evans/raytracer: what the
 * reference does between `RayTracerDynamic.configure_buffers_and_uniforms`
 * (scripts/application.py:919-1002) and the per-pass `vao.render` that runs
 * shaders/raytracer.fs.glsl (scripts/application.py:1052-1058), plus the numba scene-prep
 * it calls on the way (scripts/numba_utils/classes.py:254-257, 315-342;
 * scripts/scenes/chunker.py:94-106).  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - every function returns 0 on success, a negative RT_E_* code on failure; the text of
 *    the last failure of a context is `rt_last_error(ctx)`;
 *  - host buffers are caller-owned and copied during the call; the context owns its
 *    device memory; one context per GPU per host thread, calls are not re-entrant;
 *  - images are W*H*3 float32, row 0 = BOTTOM scanline (what `fbo.read` returns in the
 *    reference, application.py:1104-1107);
 *  - all arithmetic on the path is IEEE binary32 in the expression order written down in
 *    DESIGN.md ("arithmetic contract"), so results are bit-identical to oracle/rt_oracle.c.
 *
 * The CPU oracle (oracle/librt_oracle.so) exports the same symbols so that the test
 * harness can drive both through one binding; the product never loads it.
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum {
    RT_OK = 0,
    RT_E_ARG = -1,     /* bad argument (null pointer, negative size, index out of range) */
    RT_E_STATE = -2,   /* call order: no scene / no camera / no image size yet            */
    RT_E_DEVICE = -3,  /* HIP runtime error (text in rt_last_error)                        */
    RT_E_LIMIT = -4,   /* scene exceeds a stated limit (BVH depth > RT_MAX_STACK-1 ...)    */
    RT_E_NOMEM = -5
};

/* rt_render flags */
enum {
    RT_ACCUM_MEAN = 0, /* reference running mean: c = c*(1-1/p) + L*(1/p), p = pass index
                          (raytracer.fs.glsl:1013-1025); needs pass_stride == 1            */
    RT_ACCUM_SUM = 1   /* c += L  (spp-sharding across GPUs; combine with an all-reduce)   */
};

#define RT_MAX_STACK 32 /* traversal stack entries; reference uses 16 (raytracer.fs.glsl:553)
                           but its builder allows depth 20 (chunker.py:24) -> needs 21      */

/* Work counters of the render/first-hit launches since the last rt_reset_stats.
 * ray = one closest-hit query = one calculateRayCollision (raytracer.fs.glsl:518-641). */
typedef struct rt_stats_t {
    uint64_t rays;        /* closest-hit queries issued                                   */
    uint64_t box_tests;   /* boundingBoxIntersect evaluations  (B)                        */
    uint64_t tri_tests;   /* rayTriangle evaluations           (T)                        */
    uint64_t shaded_hits; /* hits that reached the material code (H)                      */
    uint64_t samples;     /* camera samples started (pixels * rays_per_pixel * passes)    */
    double kernel_ms;     /* device time of those launches (hipEvent), 0 for the oracle   */
    uint64_t launches;
    /* wave scheduler statistics of counting launches (0 for the oracle): block executions and
     * the lanes active in them: {node_steps, node_lanes, tri_steps, tri_lanes, path_steps, path_lanes} */
    uint64_t sched[6];
    /* culling audit of counting launches in the exact mode (rt_set_cull(ctx, RT_CULL_OFF); 0 otherwise and for
     * the oracle, whose twin is rto_cull_audit): {closest-hit queries won by a triangle, of these: the winner's own
     * bounding box lies beyond the distance band of rule (a), ... is left before t = 5e-7 (rule (b))}.  Zero in [1]
     * and [2] = no culling traversal can lose a winner on the rays of those launches. */
    uint64_t cull_audit[3];
} rt_stats_t;

/* What rt_set_scene built and how the most recent k_trace launch was laid out (for tests and
 * tuning; NO counterpart in the reference).  Oracle: the scene counts, everything else 0. */
typedef struct rt_info_t {
    int32_t accel;              /* RT_ACCEL_* of the scene in place                                        */
    int32_t n_slots;            /* triangle slots (triangles reachable through the BVHs)                   */
    int32_t n_records;          /* interior traversal records (64 B each), top-level tree included         */
    int32_t top_level_tree;     /* 1: more meshes than the LDS root mask holds, traversal starts at a tree */
    int32_t stack_entries;      /* node stack entries a lane can need (tree depth + 1)                     */
    int32_t stack_entries_lds;  /* of these, kept in LDS by the most recent launch (rest: global memory)   */
    int32_t lds_bytes;          /* dynamic LDS per workgroup of that launch                                */
    int32_t workgroups_per_cu;  /* resident workgroups per CU of that launch                               */
    int32_t top_level_rebuilds; /* times the top-level tree was rebuilt from posed boxes                   */
    int32_t n_roots;            /* entries of the root table: one per mesh, fewer when small meshes share a tree */
    int32_t root_table_variants;/* copies of the root table that launch's kernel kept in LDS: 8 = one per direction
                                   octant (small SAH scenes), 1 = the plain table, 0 = no launch yet / oracle */
    int32_t reserved[5];
} rt_info_t;

/* replaces: ProgramABC.__init__/initialise creating the GL context (application.py:499-575) */
int rt_create(int device, rt_ctx** out);
int rt_destroy(rt_ctx* ctx);
const char* rt_last_error(const rt_ctx* ctx);
/* "hip" for librt_hip.so, "oracle" for the CPU oracle */
const char* rt_backend_name(void);

/* replaces: the buffer uploads of configure_buffers_and_uniforms (application.py:924-987):
 *   tris28        N x 28 float32 std140 `Triangle` image  (SSBO 9,  raytracer.fs.glsl:93-110)
 *   meshes128     128-byte std140 `Mesh` records          (UBO 10,  :186-198)
 *   nodes48       48-byte std140 `GraphNode` records      (SSBO 12, :136-149)
 *   leaf_tri_ids  flat int32 leaf triangle-id list        (SSBO 13, :151-153)
 *   spheres96     96-byte std140 `Sphere` records         (UBO 1,   :80-91), may be NULL/0
 *   atmosphere80  one 80-byte `Material`                  (UBO 11,  :53-74)
 * Triangle positions given here are the REST pose of every mesh when poses are later
 * applied with rt_update_mesh_pose, or final world positions if that is never called.
 * The library re-lays the data out (DESIGN.md "data layout in HBM"), validates every index
 * and the BVH depth, and refits all node boxes from the triangles. */
int rt_set_scene(rt_ctx* ctx, const float* tris28, int n_tris, const void* meshes128, int n_meshes,
                 const void* nodes48, int n_nodes, const int32_t* leaf_tri_ids, int n_ids,
                 const void* spheres96, int n_spheres, const void* atmosphere80);

/* replaces: re-upload of UBO 10/11 each frame for animated materials (application.py:964-976) */
int rt_set_materials(rt_ctx* ctx, const void* meshes128, int n_meshes, const void* atmosphere80);

/* replaces: numba update_triangles_to_csys (numba_utils/classes.py:195-224, 254-257) +
 * chunker refit (chunker.py:94-106, 197-214) + triangles_to_array re-pack (:315-342) for one
 * mesh, on the device.  T16 = row-major float32 4x4, rotation in [0:3,0:3], translation in
 * ROW 3 (row-vector convention, numba_utils/classes.py:44-48). */
int rt_update_mesh_transform(rt_ctx* ctx, int mesh, const float T16[16]);
/* same, from pos (x,y,z) + quaternion (x,y,z,w) with the dtype rules of
 * numba_utils/functions.py:35-88 applied on the host side of the library */
int rt_update_mesh_pose(rt_ctx* ctx, int mesh, const float pos[3], const float quat[4]);
/* read back the posed std140 triangle image / node image (tests, picking) */
int rt_read_tris28(rt_ctx* ctx, float* out_tris28);
int rt_read_nodes48(rt_ctx* ctx, void* out_nodes48);

/* replaces: chunker.chunk_mesh_bvh + BVHParentNode split rule (scenes/chunker.py:17-25, 113-245)
 * and the flat leaf id list of BVHGraph.register_all_and_update_node_tri_ids (:79-86) for ONE
 * mesh, on the host, natively (the Python builder cannot do 10^6 triangles).  No context and no
 * GPU needed.  pos9 = n_tris x 9 float32 (A, B, C of each triangle, current pose).  Node ids
 * start at first_node_id (creation order: both children before either subtree), leaf triangle
 * ids are first_tri_id + local index, leaf offsets start at first_leaf_offset.  Writes
 * *n_nodes_out std140 nodes (48 B, chunker.py:247-257) and exactly n_tris leaf ids.
 * RT_E_LIMIT if nodes_cap is too small (2*n_tris always suffices). */
int rt_bvh_build(const float* pos9, int n_tris, int32_t first_node_id, int32_t first_tri_id,
                 int32_t first_leaf_offset, void* nodes48_out, int nodes_cap,
                 int32_t* leaf_tri_ids_out, int32_t* n_nodes_out);
/* replaces: BVHGraph.update_graph_node_aabbs_for_changed_triangles (chunker.py:94-106, 197-214)
 * on the host: boxes of every node from the current positions, bottom-up (exact: min/max). */
int rt_bvh_refit(const float* pos9, int n_tris, int32_t first_tri_id, int32_t first_leaf_offset,
                 void* nodes48, int n_nodes, const int32_t* leaf_tri_ids);

/* NO counterpart in the reference: which per-mesh tree the NEXT rt_set_scene builds on the device.
 * RT_ACCEL_REFERENCE (default): the uploaded nodes, i.e. the reference's own BVH - the parity
 * structure, results identical to the reference's traversal.  RT_ACCEL_SAH: a binned-SAH tree with
 * leaves of <= 3 triangles over the same triangles ("performance
 * structure").  Same results: equal-distance hits resolve in the reference's order, the slab tests
 * are widened so that no triangle the reference could hit is missed, and a hit counts only if the
 * ray passes the box of the triangle's leaf in the REFERENCE's tree, so the candidate set is the
 * reference's (0 differing pixels over 2.1e10 queries).  The uploaded nodes are still
 * required: they define ranks and leaf boxes.  rt_read_nodes48 is unavailable for such a scene.
 * Oracle: accepted and ignored. */
#define RT_ACCEL_REFERENCE 0
#define RT_ACCEL_SAH 1
int rt_set_accel(rt_ctx* ctx, int accel);

/* replaces: the camera uniform writes of calculate_frame (application.py:1175-1180):
 * ViewParams, CamLocalToWorldMatrix (numpy row-major float32 bytes), CamGlobalPos,
 * depthOfFieldStrength, antialiasStrength */
int rt_set_camera(rt_ctx* ctx, const float view_params[3], const float local_to_world16[16],
                  const float cam_pos[3], float dof, float aa);

/* replaces: texA/texB creation (application.py:893-900); clears the accumulator */
int rt_resize(rt_ctx* ctx, int width, int height);
int rt_reset_accum(rt_ctx* ctx);
/* render into caller-provided DEVICE memory (W*H*3 float32, e.g. a torch tensor that a
 * later all-reduce works on); NULL returns to the internal buffer.  Oracle: host memory. */
int rt_bind_accum(rt_ctx* ctx, void* device_ptr);

/* replaces: render_with_chunk's pass loop (application.py:1190-1207): passes
 * first_pass, first_pass+stride, ... (n_passes of them) are the `frameNumber` values that
 * seed the per-pixel RNG (raytracer.fs.glsl:956-957); rays_per_pixel / max_bounces are the
 * RAYS_PER_PIXEL / MAX_BOUNCES uniforms (application.py:1182-1183); each is limited to 65535
 * (RT_E_LIMIT beyond: the kernel keeps both counters of a path in one register). */
int rt_render(rt_ctx* ctx, uint32_t first_pass, uint32_t n_passes, uint32_t pass_stride,
              int rays_per_pixel, int max_bounces, int flags);

/* NO counterpart in the reference.  For callers that cannot allocate device memory while they
 * call - code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc:
 * rt_render with every scratch buffer of its launches (the work counter, the node stack entries
 * beyond the LDS ones, the waves' staging slots for the radiance of the samples in flight) in one
 * file-scope device array of RT_STATIC_SCRATCH_BYTES instead of hipMalloc'ed memory.  Same
 * arguments, same image bit for bit; calls no hipMalloc/hipFree, creates no events (kernel_ms of
 * rt_get_stats does not grow).  Limits: the array holds the staging slots of 8192 waves at 256
 * samples per job, which is more than a launch uses, so any image size rt_resize accepts and any
 * number of passes go through (more than 1024 passes are split into launches inside the call); a
 * launch whose scratch would not fit - only a BVH whose node stack exceeds LDS by far, or more than
 * 256 passes per launch, which need larger staging slots - runs on fewer workgroups, and
 * is refused with RT_E_LIMIT if not even one workgroup fits.  Every context on a GPU shares that
 * GPU's array, which keeps the last call's contents: calls that overlap on one GPU (two contexts,
 * two streams) race; calls in stream order on one context do not.  Scene, camera, image size and
 * accumulator must have been set up before (those calls allocate).  Pose updates must have been
 * flushed: with an rt_update_mesh_pose / rt_update_mesh_transform pending the call returns
 * RT_E_STATE (applying them copies from pageable host memory and synchronises the stream, neither
 * of which is legal under stream capture) - call rt_prepare first, outside the capture.  A pending
 * rt_set_materials alone is fine (one more kernel launch, nothing else). */
#define RT_STATIC_SCRATCH_BYTES (64u << 20)
int rt_render_static(rt_ctx* ctx, uint32_t first_pass, uint32_t n_passes, uint32_t pass_stride,
                     int rays_per_pixel, int max_bounces, int flags);

/* replaces: the place where the reference runs its scene prep - update_triangles_to_csys, the BVH
 * refit and the re-pack at the top of configure_buffers_and_uniforms (application.py:939-959),
 * i.e. BEFORE the pass loop.  Applies every pending rt_update_mesh_pose / rt_update_mesh_transform
 * / rt_set_materials on the device now (pose kernel, bottom-up refit, record re-pack; may copy from
 * host memory and synchronise the stream).  rt_render / rt_first_hit / rt_read_* do this implicitly;
 * rt_render_static refuses to (see above).  RT_E_STATE without a scene. */
int rt_prepare(rt_ctx* ctx);

/* replaces: the MAX_CYCLES uniform (application.py:132, 997): a pass whose index (frameNumber)
 * is larger leaves the image as it is (raytracer.fs.glsl:931-935) - rt_render skips it.
 * Default: no cap (0xffffffff); the Application mirror sets the reference's 2048*2048. */
int rt_set_max_cycles(rt_ctx* ctx, uint32_t max_cycles);

/* primary ray of sample 0 of `pass` per pixel: closest triangle id (-1 = miss; spheres
 * report -2-sphere_index) and distance (1e20 on a miss).  No reference counterpart: this is
 * the first-hit parity probe BASELINE.json asks for. */
int rt_first_hit(rt_ctx* ctx, uint32_t pass, int32_t* tri_id_out, float* t_out);

/* replaces: fbo.read (application.py:1104).  rt_read_accum copies the raw accumulator;
 * rt_read_mean divides a SUM accumulator by the passes accumulated (identity in MEAN mode) */
int rt_read_accum(rt_ctx* ctx, float* out_whc, uint64_t* n_passes_out);
int rt_read_mean(rt_ctx* ctx, float* out_whc);
/* tell the context how many passes an externally combined (all-reduced) SUM now holds */
int rt_set_accum_passes(rt_ctx* ctx, uint64_t n_passes);
/* NO counterpart in the reference (it has no multi-GPU path).  The one exchange step of the spp-sharded path (SURVEY.md
 * 8(e)) for a host without torch: ncclAllReduce(sum, float32) of the W*H*3 accumulator over `nccl_comm` (an ncclComm_t
 * the caller created, one rank per GPU), in place, on the context's stream - ordered after the renders, no host
 * synchronisation - and the accumulator then counts as holding `total_passes` passes (what rt_read_mean divides by).
 * The shards must have been rendered with RT_ACCUM_SUM.  RCCL is resolved at the first call (an instance the process
 * has already loaded - the one the communicator came from - else librccl.so.1; RT_RCCL_LIB names another): the library
 * does not link it.  RT_E_DEVICE if RCCL cannot be found or the collective fails.  Oracle: RT_E_STATE (single process). */
int rt_allreduce_sum(rt_ctx* ctx, void* nccl_comm, uint64_t total_passes);

/* Traversal mode.  RT_CULL_OFF (default) = STRICT: every box the reference would open is opened,
 * so results equal the oracle bit-for-bit.  RT_CULL_ON: boxes (a) wholly beyond the current closest
 * hit (with a 2^-10 relative+absolute safety band) or (b) left by the ray before t = 5e-7 are
 * skipped.  RT_CULL_BAND: rule (a) alone - it can drop a hit only if the triangle's computed distance
 * is short of its own box by more than the band, i.e. a rounding error of the plane distance beyond
 * 2^-10 (1 + t): rays within ~0.2 degrees of a surface's plane at the coordinates of the shipped
 * scenes (bound in DESIGN.md 5); rule (b) has a real, found, one-ulp failure case (ibid.).
 * RT_E_ARG for any other value.  No reference counterpart (the reference never culls,
 * raytracer.fs.glsl:558-632).  Oracle: accepted and ignored. */
#define RT_CULL_OFF 0
#define RT_CULL_ON 1
#define RT_CULL_BAND 2
int rt_set_cull(rt_ctx* ctx, int mode);

/* counters: enable => launches use the counting kernel variant (slower; not for timing) */
int rt_enable_stats(rt_ctx* ctx, int enable);
int rt_reset_stats(rt_ctx* ctx);
int rt_get_stats(rt_ctx* ctx, rt_stats_t* out);
int rt_get_info(rt_ctx* ctx, rt_info_t* out);

/* launch on a caller-provided hipStream_t (e.g. torch's current stream, so that a
 * torch.distributed all-reduce of the bound accumulator is ordered after the render
 * without a host sync).  NULL is HIP's null (legacy default) stream - which is what torch's
 * default stream is; RT_STREAM_OWN returns to the context's own stream.  Oracle: no-op. */
#define RT_STREAM_OWN ((void*)(intptr_t)-1)
int rt_set_stream(rt_ctx* ctx, void* hip_stream);
/* block until all work of the context is done (hipStreamSynchronize) */
int rt_sync(rt_ctx* ctx);
/* the hipStream_t the context launches on, as an opaque pointer (for event timing) */
void* rt_stream(rt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
