/* rt_oracle.c - CPU ORACLE for the rendering hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or
 * call this file; the product (raytracer_amd/, librt_hip.so) never does.
 *
 * PARITY UNPINNED: the reference (dw-evans/raytracer) has no tests, golden vectors or
 * fixtures for this path, its renderer is an OpenGL 4.6 fragment shader that cannot run in
 * the build container (no GL, no moderngl/numba/pyrr/trimesh), and none of its code is
 * compiled C.  This file is therefore a restatement written by reading the reference, and
 * is pinned only by (a) hand-derived known-answer tests and (b) an independent numpy
 * restatement (oracle/np_ref.py) agreeing bit-for-bit - see tests/test_oracle_*.py.
 *
 * What it restates (line numbers: /root/reference/shaders/raytracer.fs.glsl unless noted):
 *   main()                 :928-1044   ray generation, per-pass seed, running mean
 *   randomValue/Direction  :457-492
 *   boundingBoxIntersect   :421-453
 *   calculateRayCollision  :518-641    spheres, per-mesh BVH walk with an explicit stack
 *   rayTriangle            :305-409    plane + three edge tests (NOT Moller-Trumbore)
 *   raySphere              :251-276
 *   traceRay               :691-915
 *   std140 inputs          :53-198     consumed AS UPLOADED (AoS), unlike the product
 *   update_triangles_to_csys / refit   scripts/numba_utils/classes.py:195-224,
 *                                      scripts/scenes/chunker.py:197-214
 *
 * Arithmetic contract (shared, by specification, with raytracer_amd/csrc - the two
 * implementations share NO code): IEEE binary32, round-to-nearest-even, no implicit
 * contraction (-ffp-contract=off), subnormals kept.  Fused multiply-adds appear only where
 * written as fmaf().  Vector helpers:
 *   dot(a,b)    = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   cross(a,b)  = (fma(a.y,b.z,-(a.z*b.y)), fma(a.z,b.x,-(a.x*b.z)), fma(a.x,b.y,-(a.y*b.x)))
 *   length(a)   = sqrt(dot(a,a));  normalize(a) = a * (1/length(a))
 *   a / b in the slab test is a * (1/b) with 1/b correctly rounded (GLSL 4.60 s4.7.1 allows
 *   exactly this); RT_ORACLE_TRUE_DIV=1 in the environment switches the slab test to true
 *   division so that the effect of this choice can be measured (tests/test_oracle_div.py).
 *   sin/cos/acos/exp/log are the fixed polynomial routines det_* below (GLSL leaves their
 *   precision to the driver).
 * Variants (rto_set_variant, or RT_ORACLE_TRUE_DIV / RT_ORACLE_LIBM / RT_ORACLE_CARRY = 1 in the
 * environment at rt_create): the same restatement under the OTHER admissible readings of the
 * shader's text, so that the distance between them can be measured (tests/test_gpu_envelope.py,
 * profiles/r02_envelope.txt): true division in the slab test (:425-442 as written), the C
 * library's sinf/cosf/acosf/expf/logf in place of det_*, and the `doHitBackside` carry-over of
 * :522, 605-610 in the reference's own traversal order (initial value false [9.6]).
 * rto_first_hit_grazing marks the epsilon-grazing primary rays of SURVEY.md 9.7.
 * Decisions on undefined behaviour in the shader are those of SURVEY.md 9.6 and are marked
 * [9.6] where they apply.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/rt_hip.h"

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 neg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline float dot(v3 a, v3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
static inline v3 cross(v3 a, v3 b) {
    return V(fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x)));
}
static inline float length(v3 a) { return sqrtf(dot(a, a)); }
static inline v3 normalize(v3 a) { return scale(a, 1.0f / length(a)); }
/* o + t*d with one rounding per component */
static inline v3 madd(v3 o, float t, v3 d) { return V(fmaf(t, d.x, o.x), fmaf(t, d.y, o.y), fmaf(t, d.z, o.z)); }
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* ------------------------------------------------------------------ deterministic elementary functions
 * Cephes-style single precision kernels (public-domain coefficient sets), evaluated with
 * fmaf Horner steps exactly as written.  Domain notes: sin/cos |x| < ~1e4. */
static void det_sincos(float x, float* s_out, float* c_out) {
    float q = rintf(x * 0x1.45f306p-1f);             /* x * 2/pi, nearest integer (ties even) */
    float r = fmaf(q, -0x1.921fb6p+0f, x);           /* 3-term Cody-Waite reduction of pi/2   */
    r = fmaf(q, 0x1.777a5cp-25f, r);
    r = fmaf(q, 0x1.ee59dap-50f, r);
    float z = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = fmaf(r * z, ps, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
    int n = (int)q & 3;
    float ss = (n & 1) ? c : s;
    float cc = (n & 1) ? s : c;
    if (n == 2 || n == 3) ss = -ss;
    if (n == 1 || n == 2) cc = -cc;
    *s_out = ss;
    *c_out = cc;
}

static float det_asin_poly(float a, float z) { /* a + a*z*P(z), |a| <= 0.5ish */
    float p = fmaf(4.2163199048e-2f, z, 2.4181311049e-2f);
    p = fmaf(p, z, 4.5470025998e-2f);
    p = fmaf(p, z, 7.4953002686e-2f);
    p = fmaf(p, z, 1.6666752422e-1f);
    return fmaf(a * z, p, a);
}

static float det_acos(float x) {
    float ax = fabsf(x);
    if (ax > 0.5f) {
        float z = 0.5f * (1.0f - ax);
        float w = sqrtf(z);
        float r = 2.0f * det_asin_poly(w, z);
        return x > 0.0f ? r : 0x1.921fb6p+1f - r;
    }
    return 0x1.921fb6p+0f - det_asin_poly(x, x * x);
}

static float det_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97f) return 0.0f;
    float k = rintf(x * 0x1.715476p+0f);
    float r = fmaf(k, -0x1.62e43p-1f, x);
    r = fmaf(k, 0x1.05c61p-29f, r);
    float p = fmaf(1.9875691500e-4f, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float y = fmaf(p, r * r, r) + 1.0f;
    int ki = (int)k;
    int k1 = ki / 2, k2 = ki - k1;                   /* two normal powers of two */
    return (y * u2f((uint32_t)(k1 + 127) << 23)) * u2f((uint32_t)(k2 + 127) << 23);
}

static float det_log(float x) {
    if (x != x || x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e = 0;
    if (x < 0x1p-126f) { x *= 0x1p23f; e = -23; }
    uint32_t u = f2u(x);
    e += (int)(u >> 23) - 126;                       /* x = m * 2^e, m in [0.5,1) */
    float m = u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float p = fmaf(7.0376836292e-2f, m, -1.1514610310e-1f);
    p = fmaf(p, m, 1.1676998740e-1f);
    p = fmaf(p, m, -1.2420140846e-1f);
    p = fmaf(p, m, 1.4249322787e-1f);
    p = fmaf(p, m, -1.6668057665e-1f);
    p = fmaf(p, m, 2.0000714765e-1f);
    p = fmaf(p, m, -2.4999993993e-1f);
    p = fmaf(p, m, 3.3333331174e-1f);
    float fe = (float)e;
    float y = (m * z) * p;
    y = fmaf(-2.12194440e-4f, fe, y);
    y = fmaf(-0.5f, z, y);
    return fmaf(0.693359375f, fe, m + y);
}

/* exported for the math parity tests */
void rto_det_math(int which, const float* in, float* out, int n) {
    for (int i = 0; i < n; i++) {
        float s, c;
        switch (which) {
        case 0: det_sincos(in[i], &s, &c); out[i] = s; break;
        case 1: det_sincos(in[i], &s, &c); out[i] = c; break;
        case 2: out[i] = det_acos(in[i]); break;
        case 3: out[i] = det_exp(in[i]); break;
        case 4: out[i] = det_log(in[i]); break;
        default: out[i] = NAN;
        }
    }
}

/* ------------------------------------------------------------------ std140 inputs (AoS, as uploaded) */
typedef struct {            /* 80 bytes, :53-68 */
    float color[4];
    float specularColor[3]; float _p0;
    float emissionColor[3]; float emissionStrength;
    float specularStrength, smoothness, transmission, ior;
    float metallic; int32_t transparentFromBehind; int32_t _p1[2];
} Material;
typedef struct { float position[3]; float radius; Material material; } Sphere;          /* 96  */
typedef struct {            /* 112 bytes = 28 floats, :93-104 */
    float posA[3], _a, posB[3], _b, posC[3], _c;
    float normalA[3], _d, normalB[3], _e, normalC[3];
    int32_t meshIndex; int32_t triId; int32_t _f[3];
} Triangle;
typedef struct { int32_t index, node0Id, _p[2]; float bboxMin[4], bboxMax[4]; Material material; } Mesh; /* 128 */
typedef struct {            /* 48 bytes, :136-144 */
    float aabbMin[3], _p; float aabbMax[3];
    int32_t id, childId1, childId2, childObjOffset, childObjCount;
} GraphNode;

_Static_assert(sizeof(Material) == 80, "Material std140");
_Static_assert(sizeof(Sphere) == 96, "Sphere std140");
_Static_assert(sizeof(Triangle) == 112, "Triangle std140");
_Static_assert(sizeof(Mesh) == 128, "Mesh std140");
_Static_assert(sizeof(GraphNode) == 48, "GraphNode std140");

struct rt_ctx {
    Triangle* tris; Triangle* tris0; int n_tris;
    Mesh* meshes; int n_meshes;
    GraphNode* nodes; int n_nodes;
    int32_t* leaf_ids; int n_ids;
    Sphere* spheres; int n_spheres;
    Material atmosphere;
    int have_scene, have_cam;
    /* camera uniforms */
    float view_params[3], l2w[16], cam_pos[3], dof, aa;
    int w, h;
    float* accum_own; float* accum; uint64_t n_accum;
    rt_stats_t st;
    int true_div;
    int libm;                                        /* variant: libm sinf/cosf/acosf/expf/logf instead of det_* */
    int carry;                                       /* variant: doHitBackside carried across triangles (:522, 605-610) */
    uint32_t max_cycles;                             /* MAX_CYCLES uniform (:46, 931-935) */
    int sum_mode;                                    /* last rt_render used RT_ACCUM_SUM */
    char err[256];
};

#define FAIL(ctx, code, ...) do { snprintf((ctx)->err, sizeof((ctx)->err), __VA_ARGS__); return (code); } while (0)

const char* rt_backend_name(void) { return "oracle"; }
const char* rt_last_error(const rt_ctx* c) { return c ? c->err : "null context"; }

int rt_create(int device, rt_ctx** out) {
    (void)device;
    if (!out) return RT_E_ARG;
    rt_ctx* c = (rt_ctx*)calloc(1, sizeof(rt_ctx));
    if (!c) return RT_E_NOMEM;
    const char* e = getenv("RT_ORACLE_TRUE_DIV");
    c->true_div = (e && e[0] == '1');
    e = getenv("RT_ORACLE_LIBM");
    c->libm = (e && e[0] == '1');
    e = getenv("RT_ORACLE_CARRY");
    c->carry = (e && e[0] == '1');
    c->max_cycles = 0xffffffffu;
    *out = c;
    return RT_OK;
}

static void free_scene(rt_ctx* c) {
    free(c->tris); free(c->tris0); free(c->meshes); free(c->nodes); free(c->leaf_ids); free(c->spheres);
    c->tris = c->tris0 = NULL; c->meshes = NULL; c->nodes = NULL; c->leaf_ids = NULL; c->spheres = NULL;
    c->have_scene = 0;
}

int rt_destroy(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    free_scene(c);
    free(c->accum_own);
    free(c);
    return RT_OK;
}

static void* dup_mem(const void* p, size_t n) {
    void* q = malloc(n ? n : 1);
    if (q && n) memcpy(q, p, n);
    return q;
}

/* refit: every node's box = min/max over the current corners of its own triangles
 * (chunker.py:197-214).  Leaves hold their ids in leaf_ids; an interior node's set is the
 * union of its leaves, so a post-order union gives the same (exact) min/max. */
static void refit_node(rt_ctx* c, int id) {
    GraphNode* n = &c->nodes[id];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (n->childId1 < 0 && n->childId2 < 0) {
        for (int i = 0; i < n->childObjCount; i++) {
            const Triangle* t = &c->tris[c->leaf_ids[n->childObjOffset + i]];
            const float* p[3] = {t->posA, t->posB, t->posC};
            for (int k = 0; k < 3; k++)
                for (int a = 0; a < 3; a++) {
                    if (p[k][a] < lo[a]) lo[a] = p[k][a];
                    if (p[k][a] > hi[a]) hi[a] = p[k][a];
                }
        }
    } else {
        int ch[2] = {n->childId1, n->childId2};
        for (int j = 0; j < 2; j++) {
            if (ch[j] < 0) continue;
            refit_node(c, ch[j]);
            for (int a = 0; a < 3; a++) {
                if (c->nodes[ch[j]].aabbMin[a] < lo[a]) lo[a] = c->nodes[ch[j]].aabbMin[a];
                if (c->nodes[ch[j]].aabbMax[a] > hi[a]) hi[a] = c->nodes[ch[j]].aabbMax[a];
            }
        }
    }
    for (int a = 0; a < 3; a++) { n->aabbMin[a] = lo[a]; n->aabbMax[a] = hi[a]; }
}

static int node_depth(const rt_ctx* c, int id, int d) {
    if (d > 64) return d;
    const GraphNode* n = &c->nodes[id];
    int m = d;
    if (n->childId1 >= 0) { int x = node_depth(c, n->childId1, d + 1); if (x > m) m = x; }
    if (n->childId2 >= 0) { int x = node_depth(c, n->childId2, d + 1); if (x > m) m = x; }
    return m;
}

int rt_set_scene(rt_ctx* c, const float* tris28, int n_tris, const void* meshes128, int n_meshes,
                 const void* nodes48, int n_nodes, const int32_t* leaf_tri_ids, int n_ids,
                 const void* spheres96, int n_spheres, const void* atmosphere80) {
    if (!c) return RT_E_ARG;
    if (n_tris < 0 || n_meshes < 0 || n_nodes < 0 || n_ids < 0 || n_spheres < 0)
        FAIL(c, RT_E_ARG, "negative count");
    if ((n_tris && !tris28) || (n_meshes && !meshes128) || (n_nodes && !nodes48) || (n_ids && !leaf_tri_ids) ||
        (n_spheres && !spheres96) || !atmosphere80)
        FAIL(c, RT_E_ARG, "null buffer with non-zero count");
    if (n_meshes > 256 || n_spheres > 256) FAIL(c, RT_E_LIMIT, "more than 256 meshes/spheres (shader arrays, :15-17)");
    if (n_tris > (1 << 25) || n_nodes > (1 << 25)) FAIL(c, RT_E_LIMIT, "more than 2^25 triangles or nodes (the product's limit, mirrored)");
    free_scene(c);
    c->tris = (Triangle*)dup_mem(tris28, (size_t)n_tris * 112);
    c->tris0 = (Triangle*)dup_mem(tris28, (size_t)n_tris * 112);
    c->meshes = (Mesh*)dup_mem(meshes128, (size_t)n_meshes * 128);
    c->nodes = (GraphNode*)dup_mem(nodes48, (size_t)n_nodes * 48);
    c->leaf_ids = (int32_t*)dup_mem(leaf_tri_ids, (size_t)n_ids * 4);
    c->spheres = (Sphere*)dup_mem(spheres96, (size_t)n_spheres * 96);
    memcpy(&c->atmosphere, atmosphere80, 80);
    c->n_tris = n_tris; c->n_meshes = n_meshes; c->n_nodes = n_nodes; c->n_ids = n_ids; c->n_spheres = n_spheres;
    /* validation */
    for (int i = 0; i < n_ids; i++)
        if (c->leaf_ids[i] < 0 || c->leaf_ids[i] >= n_tris) FAIL(c, RT_E_ARG, "leaf_tri_ids[%d] out of range", i);
    for (int i = 0; i < n_tris; i++) {
        if (c->tris[i].meshIndex < 0 || c->tris[i].meshIndex >= n_meshes) FAIL(c, RT_E_ARG, "triangle %d: bad mesh index", i);
        if (c->tris[i].triId != i) FAIL(c, RT_E_ARG, "triangle %d: triId != index (scenes/utils.py:18-25)", i);
    }
    for (int i = 0; i < n_nodes; i++) {
        const GraphNode* n = &c->nodes[i];
        if (n->childId1 >= n_nodes || n->childId2 >= n_nodes) FAIL(c, RT_E_ARG, "node %d: child out of range", i);
        if ((n->childId1 >= 0 && n->childId1 <= i) || (n->childId2 >= 0 && n->childId2 <= i))
            FAIL(c, RT_E_ARG, "node %d: child id not greater than parent id", i);
        if (n->childId1 < 0 && n->childId2 < 0) {
            if (n->childObjCount < 0 || n->childObjOffset < 0 || (int64_t)n->childObjOffset + n->childObjCount > n_ids)
                FAIL(c, RT_E_ARG, "leaf node %d: triangle range out of bounds", i);
        }
    }
    for (int j = 0; j < n_meshes; j++) {
        int r = c->meshes[j].node0Id;
        if (r < 0 || r >= n_nodes) FAIL(c, RT_E_ARG, "mesh %d: bad root node", j);
        if (node_depth(c, r, 0) + 1 > RT_MAX_STACK) FAIL(c, RT_E_LIMIT, "mesh %d: BVH deeper than %d", j, RT_MAX_STACK - 1);
    }
    /* every triangle in at most one reachable leaf, and only in a tree of its own mesh (the
     * product's rules, mirrored; the reference's builder guarantees both, chunker.py:216-244) */
    {
        char* listed = (char*)calloc((size_t)n_tris + 1, 1);
        if (!listed) FAIL(c, RT_E_NOMEM, "out of memory");
        int bad = 0, bad_tri = -1, bad_mesh = -1;
        for (int j = 0; j < n_meshes && !bad; j++) {
            int stack[RT_MAX_STACK + 2], sp = 0;
            stack[sp++] = c->meshes[j].node0Id;
            while (sp > 0 && !bad) {
                const GraphNode* n = &c->nodes[stack[--sp]];
                if (n->childId1 >= 0 || n->childId2 >= 0) {
                    if (n->childId1 >= 0) stack[sp++] = n->childId1;
                    if (n->childId2 >= 0) stack[sp++] = n->childId2;
                    continue;
                }
                for (int i = 0; i < n->childObjCount && !bad; i++) {
                    int t = c->leaf_ids[n->childObjOffset + i];
                    if (c->tris[t].meshIndex != j) { bad = 1; bad_tri = t; bad_mesh = j; }
                    else if (listed[t]) { bad = 2; bad_tri = t; }
                    listed[t] = 1;
                }
            }
        }
        free(listed);
        if (bad == 1) FAIL(c, RT_E_ARG, "triangle %d in the BVH of mesh %d belongs to mesh %d", bad_tri, bad_mesh, c->tris[bad_tri].meshIndex);
        if (bad == 2) FAIL(c, RT_E_ARG, "triangle %d is listed in more than one leaf", bad_tri);
    }
    for (int j = 0; j < n_meshes; j++) refit_node(c, c->meshes[j].node0Id);
    c->have_scene = 1;
    return RT_OK;
}

int rt_set_materials(rt_ctx* c, const void* meshes128, int n_meshes, const void* atmosphere80) {
    if (!c) return RT_E_ARG;
    if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene");
    if (n_meshes != c->n_meshes || !meshes128) FAIL(c, RT_E_ARG, "mesh count mismatch");
    const Mesh* m = (const Mesh*)meshes128;
    for (int j = 0; j < n_meshes; j++) c->meshes[j].material = m[j].material;
    if (atmosphere80) memcpy(&c->atmosphere, atmosphere80, 80);
    return RT_OK;
}

/* world = [p,1] . T (row vector), k-ordered float32 sum without fma:
 * ((p.x*T0j + p.y*T1j) + p.z*T2j) + T3j  - the build's fixed meaning of the reference's
 * np.dot (numba_utils/classes.py:201-217); normals use the 3x3 block only. */
int rt_update_mesh_transform(rt_ctx* c, int mesh, const float T[16]) {
    if (!c || !T) return RT_E_ARG;
    if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene");
    if (mesh < 0 || mesh >= c->n_meshes) FAIL(c, RT_E_ARG, "mesh index out of range");
    for (int i = 0; i < c->n_tris; i++) {
        if (c->tris0[i].meshIndex != mesh) continue;
        const Triangle* s = &c->tris0[i];
        Triangle* d = &c->tris[i];
        const float* sp[3] = {s->posA, s->posB, s->posC};
        const float* sn[3] = {s->normalA, s->normalB, s->normalC};
        float* dp[3] = {d->posA, d->posB, d->posC};
        float* dn[3] = {d->normalA, d->normalB, d->normalC};
        for (int k = 0; k < 3; k++)
            for (int j = 0; j < 3; j++) {
                dp[k][j] = ((sp[k][0] * T[0 + j] + sp[k][1] * T[4 + j]) + sp[k][2] * T[8 + j]) + T[12 + j];
                dn[k][j] = (sn[k][0] * T[0 + j] + sn[k][1] * T[4 + j]) + sn[k][2] * T[8 + j];
            }
    }
    refit_node(c, c->meshes[mesh].node0Id);
    return RT_OK;
}

/* numba dtype rules of mat33_create_from_quaternion (numba_utils/functions.py:35-82):
 * float32 products and sums, float64 for 1/x and 2.0*x, cast to float32 at the end. */
static void pose_matrix(const float pos[3], const float q_in[4], float T[16]) {
    float q[4] = {q_in[0], q_in[1], q_in[2], q_in[3]};
    float nn = sqrtf(((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3]);
    if (!(fabs((double)nn - 1.0) <= 1e-8 + 1e-5 * 1.0)) /* np.isclose(norm, 1.0) */
        for (int i = 0; i < 4; i++) q[i] = q[i] / nn;
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float xx = x * x, yy = y * y, zz = z * z, ww = w * w;
    float xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    double inv = 1.0 / (double)(((xx + yy) + zz) + ww);
    memset(T, 0, 64);
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

int rt_update_mesh_pose(rt_ctx* c, int mesh, const float pos[3], const float quat[4]) {
    if (!c || !pos || !quat) return RT_E_ARG;
    float T[16];
    pose_matrix(pos, quat, T);
    return rt_update_mesh_transform(c, mesh, T);
}

int rt_read_tris28(rt_ctx* c, float* out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene");
    memcpy(out, c->tris, (size_t)c->n_tris * 112);
    return RT_OK;
}
int rt_read_nodes48(rt_ctx* c, void* out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene");
    memcpy(out, c->nodes, (size_t)c->n_nodes * 48);
    return RT_OK;
}

int rt_set_camera(rt_ctx* c, const float vp[3], const float m[16], const float pos[3], float dof, float aa) {
    if (!c || !vp || !m || !pos) return RT_E_ARG;
    memcpy(c->view_params, vp, 12); memcpy(c->l2w, m, 64); memcpy(c->cam_pos, pos, 12);
    c->dof = dof; c->aa = aa; c->have_cam = 1;
    return RT_OK;
}

int rt_resize(rt_ctx* c, int w, int h) {
    if (!c) return RT_E_ARG;
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768 || (int64_t)w * h > (1 << 28)) FAIL(c, RT_E_ARG, "bad image size %dx%d", w, h);
    float* p = (float*)calloc((size_t)w * h * 3, 4);
    if (!p) FAIL(c, RT_E_NOMEM, "accumulator allocation failed");
    if (c->accum == c->accum_own) c->accum = p;
    free(c->accum_own);
    c->accum_own = p; c->w = w; c->h = h; c->n_accum = 0;
    return RT_OK;
}
int rt_reset_accum(rt_ctx* c) {
    if (!c) return RT_E_ARG;
    if (!c->accum) FAIL(c, RT_E_STATE, "no image size");
    memset(c->accum, 0, (size_t)c->w * c->h * 12);
    c->n_accum = 0;
    return RT_OK;
}
int rt_bind_accum(rt_ctx* c, void* p) {
    if (!c) return RT_E_ARG;
    if (c->w <= 0 || c->h <= 0 || (!p && !c->accum_own)) FAIL(c, RT_E_STATE, "no image size");
    c->accum = p ? (float*)p : c->accum_own;
    c->n_accum = 0;
    return RT_OK;
}
int rt_set_accum_passes(rt_ctx* c, uint64_t n) { if (!c) return RT_E_ARG; c->n_accum = n; c->sum_mode = 1; return RT_OK; }

/* ------------------------------------------------------------------ the shader */
typedef struct {
    int didHit; float dst; v3 hitPoint, normal;
    Material material; int meshIndex, facetId, hitBackside;
    int sphere;                                      /* index of the sphere hit, else -1 (probe only) */
} HitInfo;
typedef struct {
    v3 origin, dir;
    /* what traceRay/calculateRayCollision read of prevHit: */
    int prevFacetId, prevMeshIndex; Material prevMaterial;
    float ior; int inSolid;
} Ray;
typedef struct { uint64_t rays, box, tri, hits; } Counters;

static inline float randomValue(uint32_t* s) {       /* :457-461 */
    uint32_t x = *s;
    x = x * ((x + 195439u) * (x + 124395u) * (x + 845921u));
    *s = x;
    return (float)x * 0x1p-32f;                      /* 4294967295.0 is 2^32 in float */
}

static inline v3 randomDirection(const rt_ctx* c, uint32_t* s) {      /* :478-492 */
    float u = randomValue(s);
    float v = randomValue(s);
    float theta = 6.2831855f * u;                    /* 2.0 * pi folded, pi = 3.14159265359 */
    float sp, cp, st, ct;
    if (c && c->libm) {                              /* variant: the C library's functions */
        float phi = acosf(2.0f * v - 1.0f);
        sp = sinf(phi); cp = cosf(phi); st = sinf(theta); ct = cosf(theta);
    } else {
        float phi = det_acos(2.0f * v - 1.0f);
        det_sincos(phi, &sp, &cp);
        det_sincos(theta, &st, &ct);
    }
    return V(sp * ct, sp * st, cp);
}

static void defaultHit(HitInfo* h) {                 /* :226-247, meshIndex = -1 [9.6] */
    memset(h, 0, sizeof(*h));
    h->dst = 1e20f;
    h->facetId = -1;
    h->meshIndex = -1;
    h->sphere = -1;
}

/* :421-453.  inv = 1/dir (correctly rounded) unless true_div. */
static inline int boxHit(const rt_ctx* c, v3 o, v3 d, v3 inv, const float* lo, const float* hi) {
    float tmin, tmax, tymin, tymax, tzmin, tzmax, t;
    if (c->true_div) {
        tmin = (lo[0] - o.x) / d.x; tmax = (hi[0] - o.x) / d.x;
        tymin = (lo[1] - o.y) / d.y; tymax = (hi[1] - o.y) / d.y;
        tzmin = (lo[2] - o.z) / d.z; tzmax = (hi[2] - o.z) / d.z;
    } else {
        tmin = (lo[0] - o.x) * inv.x; tmax = (hi[0] - o.x) * inv.x;
        tymin = (lo[1] - o.y) * inv.y; tymax = (hi[1] - o.y) * inv.y;
        tzmin = (lo[2] - o.z) * inv.z; tzmax = (hi[2] - o.z) * inv.z;
    }
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    return 1;
}

/* :305-409 */
static void rayTriangle(v3 orig, v3 dir, const Triangle* tri, int hitBehind, HitInfo* h) {
    defaultHit(h);
    v3 v0 = V(tri->posA[0], tri->posA[1], tri->posA[2]);
    v3 v1 = V(tri->posB[0], tri->posB[1], tri->posB[2]);
    v3 v2 = V(tri->posC[0], tri->posC[1], tri->posC[2]);
    v3 v0v1 = sub(v1, v0), v0v2 = sub(v2, v0);
    v3 N = cross(v0v1, v0v2);
    float nd = dot(N, dir);
    if (fabsf(nd) < 1e-6f) return;
    if (nd > 0.0f && !hitBehind) return;
    if (nd > 0.0f) h->hitBackside = 1;
    float d = -dot(N, v0);
    float t = -(dot(N, orig) + d) / nd;
    if (t < 0.0f) return;
    if (t < 1e-6f) return;
    v3 P = madd(orig, t, dir);
    float area = length(N) / 2.0f;
    v3 C = cross(sub(v1, v0), sub(P, v0));
    float w = length(C) / 2.0f / area;
    if (dot(N, C) < 0.0f) return;
    C = cross(sub(v2, v1), sub(P, v1));
    float u = length(C) / 2.0f / area;
    if (dot(N, C) < 0.0f) return;
    C = cross(sub(v0, v2), sub(P, v2));
    float v = 1.0f - u - w;
    if (dot(N, C) < 0.0f) return;
    /* normalA*u + normalB*v + normalC*w, left to right, no fma */
    h->normal = V(tri->normalA[0] * u + tri->normalB[0] * v + tri->normalC[0] * w,
                  tri->normalA[1] * u + tri->normalB[1] * v + tri->normalC[1] * w,
                  tri->normalA[2] * u + tri->normalB[2] * v + tri->normalC[2] * w);
    h->didHit = 1;
    h->dst = t;
    h->hitPoint = madd(orig, t, dir);               /* t*dir + origin */
    h->facetId = tri->triId;
    h->meshIndex = tri->meshIndex;
}

/* Audit of the product's distance-culling rules (rt_dev_trace.h: culled()), for tests and tools: the interval the
 * compare ladder ends with for a box the slab test passes, and a count of the queries whose WINNER lies in a leaf
 * box that those rules could skip: rule (b) exit < 5e-7, rule (a) entry > t_w (1 + 2^-10) + 2^-10 with the winner's
 * own distance as the best in hand (the boxes above the leaf nest around it and float subtract / multiply are
 * monotonic, so the leaf is the box of the path that the rules reject first).  rto_cull_audit() switches it on. */
static int g_audit = 0;
static unsigned long long g_audit_n[5];              /* queries with a triangle winner; rule (a), rule (b) cases for the winner's leaf box in the
                                                        reference's tree; the same for the winner's OWN bounding box (inside every box any structure
                                                        over the same triangles can put it in, so it covers the SAH structure's boxes too) */
static inline void boxInterval(v3 o, v3 inv, const float* lo, const float* hi, float* tmin_out, float* tmax_out) {
    float tmin = (lo[0] - o.x) * inv.x, tmax = (hi[0] - o.x) * inv.x, t;
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo[1] - o.y) * inv.y, tymax = (hi[1] - o.y) * inv.y;
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo[2] - o.z) * inv.z, tzmax = (hi[2] - o.z) * inv.z;
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    *tmin_out = tmin;
    *tmax_out = tmax;
}

/* :251-276 */
static void raySphere(v3 orig, v3 dir, const Sphere* s, HitInfo* h) {
    defaultHit(h);
    v3 c = V(s->position[0], s->position[1], s->position[2]);
    v3 oo = sub(orig, c);
    float a = dot(dir, dir);
    float b = 2.0f * dot(oo, dir);
    float cc = dot(oo, oo) - s->radius * s->radius;
    float disc = b * b - (4.0f * a) * cc;
    if (disc >= 0.0f) {
        float dst = (-b - sqrtf(disc)) / (2.0f * a);
        if (dst >= 0.0f) {
            h->didHit = 1;
            h->dst = dst;
            h->hitPoint = madd(orig, dst, dir);
            h->normal = normalize(sub(h->hitPoint, c));
        }
    }
}

/* :518-641 */
static void calculateRayCollision(const rt_ctx* c, const Ray* ray, HitInfo* closest, Counters* k) {
    defaultHit(closest);
    k->rays++;
    v3 inv = V(1.0f / ray->dir.x, 1.0f / ray->dir.y, 1.0f / ray->dir.z);
    HitInfo h;
    int doHitBackside = 0;                           /* :522 uninitialised; false [9.6] (carry variant only) */
    float win_lo = 0.0f, win_hi = 0.0f, own_lo = 0.0f, own_hi = 0.0f;   /* audit: intervals of the winner's leaf box and of its own box */
    for (int i = 0; i < c->n_spheres; i++) {
        raySphere(ray->origin, ray->dir, &c->spheres[i], &h);
        if (h.dst < 1e-6f) continue;
        if (h.didHit && h.dst < closest->dst) { *closest = h; closest->material = c->spheres[i].material; closest->sphere = i; }
    }
    for (int j = 0; j < c->n_meshes; j++) {
        int stack[RT_MAX_STACK];                    /* ids, not 48-byte nodes [9.6] */
        int sp = 0;
        stack[sp++] = c->meshes[j].node0Id;
        while (sp > 0) {
            const GraphNode* node = &c->nodes[stack[--sp]];
            int valid1 = node->childId1 >= 0, valid2 = node->childId2 >= 0;
            if (valid1) {
                const GraphNode* ch = &c->nodes[node->childId1];
                k->box++;
                if (boxHit(c, ray->origin, ray->dir, inv, ch->aabbMin, ch->aabbMax)) stack[sp++] = node->childId1;
            }
            if (valid2) {
                const GraphNode* ch = &c->nodes[node->childId2];
                k->box++;
                if (boxHit(c, ray->origin, ray->dir, inv, ch->aabbMin, ch->aabbMax)) stack[sp++] = node->childId2;
            }
            if (valid1 || valid2) continue;
            k->box++;
            if (!boxHit(c, ray->origin, ray->dir, inv, node->aabbMin, node->aabbMax)) continue;
            for (int i = 0; i < node->childObjCount; i++) { /* no 100-triangle cap [9.6] */
                const Triangle* tri = &c->tris[c->leaf_ids[node->childObjOffset + i]];
                if (tri->triId == ray->prevFacetId) continue;
                int hitBehind;
                if (c->carry) {                      /* variant: :605-615 as written, value carried across triangles */
                    if (ray->inSolid) { if (ray->prevMeshIndex == tri->meshIndex) doHitBackside = 1; }
                    else doHitBackside = !c->meshes[tri->meshIndex].material.transparentFromBehind;
                    hitBehind = doHitBackside;
                }
                else if (ray->inSolid) hitBehind = (ray->prevMeshIndex == tri->meshIndex); /* [9.6] no carry-over */
                else hitBehind = !c->meshes[tri->meshIndex].material.transparentFromBehind;
                k->tri++;
                rayTriangle(ray->origin, ray->dir, tri, hitBehind, &h);
                if (h.dst < 1e-6f) continue;
                if (h.didHit && h.dst < closest->dst) {
                    *closest = h; closest->material = c->meshes[tri->meshIndex].material;
                    if (g_audit) {
                        float lo[3], hi[3];
                        for (int a = 0; a < 3; a++) {
                            lo[a] = fminf(tri->posA[a], fminf(tri->posB[a], tri->posC[a]));
                            hi[a] = fmaxf(tri->posA[a], fmaxf(tri->posB[a], tri->posC[a]));
                        }
                        boxInterval(ray->origin, inv, node->aabbMin, node->aabbMax, &win_lo, &win_hi);
                        boxInterval(ray->origin, inv, lo, hi, &own_lo, &own_hi);
                    }
                }
            }
        }
    }
    if (g_audit && closest->didHit && closest->facetId >= 0) {
        const int rule_b = win_hi < 5e-7f;
        const float beyond = fmaf(closest->dst, 0x1p-10f, closest->dst) + 0x1p-10f;
        const int rule_a = win_lo > beyond, own_b = own_hi < 5e-7f, own_a = own_lo > beyond;
#pragma omp atomic
        g_audit_n[0]++;
        if (rule_a || rule_b || own_a || own_b) {
#pragma omp critical(rto_audit)
            {
                g_audit_n[1] += rule_a;
                g_audit_n[2] += rule_b;
                g_audit_n[3] += own_a;
                g_audit_n[4] += own_b;
                if (g_audit > 1 && (rule_a || rule_b))
                    fprintf(stderr, "[cull audit] rule %s: t_w %.9g leaf interval [%.9g, %.9g] facet %d o (%.9g %.9g %.9g) d (%.9g %.9g %.9g)\n",
                            rule_b ? "b" : "a", closest->dst, win_lo, win_hi, closest->facetId, ray->origin.x, ray->origin.y, ray->origin.z,
                            ray->dir.x, ray->dir.y, ray->dir.z);
            }
        }
    }
}

static inline v3 reflect(v3 I, v3 N) {               /* I - 2*dot(N,I)*N */
    float s = 2.0f * dot(N, I);
    return V(fmaf(-s, N.x, I.x), fmaf(-s, N.y, I.y), fmaf(-s, N.z, I.z));
}

/* :691-915 */
static v3 traceRay(const rt_ctx* c, Ray* ray, uint32_t* rng, int maxBounces, Counters* k) {
    v3 incoming = V(0, 0, 0), rayColor = V(1, 1, 1);
    HitInfo hit;
    for (int i = 0; i <= maxBounces; i++) {
        calculateRayCollision(c, ray, &hit, k);
        if (!hit.didHit) break;                      /* + getEnvironmentLight()*rayColor == +0, :279-282 */
        k->hits++;
        Material prevmat = ray->prevMaterial;
        const Material* m = &hit.material;
        if (m->color[3] < 0.99999f) {
            if (randomValue(rng) > m->color[3]) {
                if (i == maxBounces) break;
                ray->origin = hit.hitPoint;
                ray->prevFacetId = hit.facetId; ray->prevMeshIndex = hit.meshIndex; ray->prevMaterial = hit.material;
                continue;
            }
        }
        v3 emitted = scale(V(m->emissionColor[0], m->emissionColor[1], m->emissionColor[2]), m->emissionStrength);
        incoming = add(incoming, mul(emitted, rayColor));

        float nf = hit.hitBackside ? -1.0f : 1.0f;
        v3 nrm = scale(hit.normal, nf);              /* normalFlip * hitInfo.normal */
        float n2 = (ray->inSolid && hit.hitBackside) ? c->atmosphere.ior : m->ior;
        float n1 = ray->ior;
        /* specularDir = refract(...) at :762 is overwritten on every path that reads it -> omitted */
        v3 diffuseDir = normalize(add(neg(nrm), randomDirection(c, rng)));
        float costheta = clampf(dot(normalize(nrm), normalize(neg(ray->dir))), -1.0f, 1.0f);
        float r0 = (n1 - n2) / (n1 + n2);
        r0 = r0 * r0;                                /* pow(x,2) [9.6] */
        float om = 1.0f - costheta;
        float om5 = ((om * om) * (om * om)) * om;    /* pow(x,5) [9.6] */
        float schlick = r0 + (1.0f - r0) * om5;

        int isSpecular = randomValue(rng) < m->specularStrength;
        v3 specularDir = V(0, 0, 0);
        if (isSpecular) {
            specularDir = reflect(ray->dir, nrm);
            diffuseDir = neg(diffuseDir);
        } else if (randomValue(rng) > schlick) {
            isSpecular = 0;
            if (randomValue(rng) < m->transmission) {
                float a = c->libm ? -logf(prevmat.transmission) : -det_log(prevmat.transmission);
                float e = c->libm ? expf(-a * hit.dst) : det_exp(-a * hit.dst);
                v3 tc = scale(V(prevmat.color[0], prevmat.color[1], prevmat.color[2]), e);
                rayColor = mul(rayColor, scale(tc, prevmat.color[3]));
                if (hit.hitBackside) { ray->inSolid = 0; ray->ior = c->atmosphere.ior; }
                else { ray->inSolid = 1; ray->ior = m->ior; }
            } else {
                diffuseDir = neg(diffuseDir);
                if (hit.hitBackside) { ray->inSolid = 1; ray->ior = m->ior; }
            }
        } else {
            isSpecular = 1;
            specularDir = reflect(ray->dir, nrm);
            diffuseDir = neg(diffuseDir);
        }
        if (i == maxBounces) break;
        ray->origin = hit.hitPoint;
        if (isSpecular) {
            float a = m->smoothness;                 /* mix(x,y,a) = x*(1-a) + y*a */
            float b = 1.0f - a;
            ray->dir = V(diffuseDir.x * b + specularDir.x * a, diffuseDir.y * b + specularDir.y * a,
                         diffuseDir.z * b + specularDir.z * a);
            rayColor = mul(rayColor, V(m->specularColor[0], m->specularColor[1], m->specularColor[2]));
        } else {
            ray->dir = diffuseDir;
            rayColor = mul(rayColor, V(m->color[0], m->color[1], m->color[2]));
        }
        ray->prevFacetId = hit.facetId; ray->prevMeshIndex = hit.meshIndex; ray->prevMaterial = hit.material;
    }
    return incoming;
}

/* :928-989: pixel (px,py), py counted from the BOTTOM row */
static inline uint32_t pixelSeed(const rt_ctx* c, int px, int py, uint32_t pass) {
    return (uint32_t)py * (uint32_t)c->w + (uint32_t)px + pass * 747796405u;
}
static inline v3 viewPointOf(const rt_ctx* c, int px, int py) {
    /* fragPosition = NDC of the pixel centre (raytracer.vs.glsl:6-9) [9.6] */
    float xn = (2.0f * ((float)px + 0.5f)) / (float)c->w - 1.0f;
    float yn = (2.0f * ((float)py + 0.5f)) / (float)c->h - 1.0f;
    float lx = (xn * 0.5f) * c->view_params[0], ly = (yn * 0.5f) * c->view_params[1], lz = c->view_params[2];
    const float* M = c->l2w;                         /* transpose(G)*v == rows of the numpy matrix */
    float r[3];
    for (int i = 0; i < 3; i++) {
        float t = M[4 * i + 0] * lx;
        t = fmaf(M[4 * i + 1], ly, t);
        t = fmaf(M[4 * i + 2], lz, t);
        r[i] = t + M[4 * i + 3];
    }
    return V(r[0], r[1], r[2]);
}
static inline void newSampleRay(const rt_ctx* c, v3 viewPoint, uint32_t* rng, Ray* ray) {
    float r1 = randomValue(rng), r2 = randomValue(rng), r3 = randomValue(rng), r4 = randomValue(rng);
    float near = c->view_params[2];
    v3 dofO = V(((r1 * 2.0f - 1.0f) * c->dof) * near, ((r2 * 2.0f - 1.0f) * c->dof) * near,
                ((0.0f * 2.0f - 1.0f) * c->dof) * near);
    v3 aaO = V((r3 * 2.0f - 1.0f) * c->aa, (r4 * 2.0f - 1.0f) * c->aa, (0.0f * 2.0f - 1.0f) * c->aa);
    ray->dir = normalize(add(add(viewPoint, dofO), aaO));
    ray->origin = sub(V(c->cam_pos[0], c->cam_pos[1], c->cam_pos[2]), dofO);
    ray->prevFacetId = -1; ray->prevMeshIndex = -1;
    ray->prevMaterial = c->atmosphere;
    ray->ior = c->atmosphere.ior;
    ray->inSolid = 0;                                /* [9.6] */
}

static int ready(rt_ctx* c) {
    if (!c->have_scene) { snprintf(c->err, sizeof c->err, "no scene"); return 0; }
    if (!c->have_cam) { snprintf(c->err, sizeof c->err, "no camera"); return 0; }
    if (!c->accum || c->w <= 0 || c->h <= 0) { snprintf(c->err, sizeof c->err, "no image size"); return 0; }
    return 1;
}

int rt_render(rt_ctx* c, uint32_t first, uint32_t n_passes, uint32_t stride, int rpp, int maxBounces, int flags) {
    if (!c) return RT_E_ARG;
    if (!ready(c)) return RT_E_STATE;
    if (rpp < 1 || maxBounces < 0 || stride < 1) FAIL(c, RT_E_ARG, "bad sampling arguments");
    if (flags != RT_ACCUM_MEAN && flags != RT_ACCUM_SUM) FAIL(c, RT_E_ARG, "bad flags");
    if (flags == RT_ACCUM_MEAN && stride != 1) FAIL(c, RT_E_ARG, "running mean needs pass_stride 1");
    /* :931-935: frameNumber > MAX_CYCLES returns the previous frame */
    if (first > c->max_cycles) n_passes = 0;
    else { uint64_t m = ((uint64_t)c->max_cycles - first) / stride + 1; if (n_passes > m) n_passes = (uint32_t)m; }
    uint64_t rays = 0, box = 0, tri = 0, hits = 0;
    const int W = c->w, H = c->h;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, box, tri, hits)
    for (int py = 0; py < H; py++) {
        Counters k = {0, 0, 0, 0};
        for (int px = 0; px < W; px++) {
            v3 viewPoint = viewPointOf(c, px, py);
            float* acc = &c->accum[((size_t)py * W + px) * 3];
            v3 col = V(acc[0], acc[1], acc[2]);
            for (uint32_t p = 0; p < n_passes; p++) {
                uint32_t pass = first + p * stride;
                uint32_t rng = pixelSeed(c, px, py, pass);
                v3 total = V(0, 0, 0);
                for (int s = 0; s < rpp; s++) {
                    Ray ray;
                    newSampleRay(c, viewPoint, &rng, &ray);
                    total = add(total, traceRay(c, &ray, &rng, maxBounces, &k));
                }
                total = V(total.x / (float)rpp, total.y / (float)rpp, total.z / (float)rpp);
                if (flags == RT_ACCUM_SUM) col = add(col, total);
                else {
                    float wgt = 1.0f / (float)pass;      /* :1015-1021 */
                    float om = 1.0f - wgt;
                    col = V(col.x * om + total.x * wgt, col.y * om + total.y * wgt, col.z * om + total.z * wgt);
                }
            }
            acc[0] = col.x; acc[1] = col.y; acc[2] = col.z;
        }
        rays += k.rays; box += k.box; tri += k.tri; hits += k.hits;
    }
    c->n_accum += n_passes;
    c->sum_mode = (flags == RT_ACCUM_SUM);
    c->st.rays += rays; c->st.box_tests += box; c->st.tri_tests += tri; c->st.shaded_hits += hits;
    c->st.samples += (uint64_t)W * H * n_passes * (uint64_t)rpp;
    c->st.launches++;
    return RT_OK;
}

/* the product's allocation-free variant renders the same image; the oracle has nothing to allocate */
int rt_render_static(rt_ctx* c, uint32_t first, uint32_t n_passes, uint32_t stride, int rpp, int maxBounces, int flags) {
    return rt_render(c, first, n_passes, stride, rpp, maxBounces, flags);
}
/* the oracle applies poses when they are set: nothing is ever pending */
int rt_prepare(rt_ctx* c) { if (!c) return RT_E_ARG; if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene"); return RT_OK; }
int rt_set_max_cycles(rt_ctx* c, uint32_t m) { if (!c) return RT_E_ARG; c->max_cycles = m; return RT_OK; }

int rt_first_hit(rt_ctx* c, uint32_t pass, int32_t* ids, float* ts) {
    if (!c || !ids || !ts) return RT_E_ARG;
    if (!ready(c)) return RT_E_STATE;
    uint64_t rays = 0, box = 0, tri = 0;
    const int W = c->w, H = c->h;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : rays, box, tri)
    for (int py = 0; py < H; py++) {
        Counters k = {0, 0, 0, 0};
        for (int px = 0; px < W; px++) {
            uint32_t rng = pixelSeed(c, px, py, pass);
            Ray ray;
            HitInfo h;
            newSampleRay(c, viewPointOf(c, px, py), &rng, &ray);
            calculateRayCollision(c, &ray, &h, &k);
            int id = -1;
            if (h.didHit) id = h.sphere >= 0 ? -2 - h.sphere : h.facetId;
            ids[(size_t)py * W + px] = id;
            ts[(size_t)py * W + px] = h.dst;
        }
        rays += k.rays; box += k.box; tri += k.tri;
    }
    c->st.rays += rays; c->st.box_tests += box; c->st.tri_tests += tri;
    c->st.samples += (uint64_t)W * H;
    c->st.launches++;
    return RT_OK;
}

/* Variant switches (see the header comment).  Negative = leave as it is. */
int rto_set_variant(rt_ctx* c, int true_div, int libm, int carry) {
    if (!c) return RT_E_ARG;
    if (true_div >= 0) c->true_div = true_div != 0;
    if (libm >= 0) c->libm = libm != 0;
    if (carry >= 0) c->carry = carry != 0;
    return RT_OK;
}

/* the primary ray of (pixel, pass) is epsilon-grazing iff
 *   bit 0: another triangle (or sphere) is hit within |t2 - t1| <= 1e-5 * max(1, t1) of the winner,
 *   bit 1: one of the three edge values N.C of the winning triangle has |N.C| <= 1e-6 * |N|^2,
 *   bit 2: the winning triangle is met at |N.d| < 1e-5 * |N|.
 * A pixel whose ray MISSES is grazing (bit 3) iff some triangle in a leaf the ray reaches is missed
 * by an edge value that small (the neighbouring reading of the arithmetic could have hit it).
 * mask_out = W*H bytes; `only` (may be NULL) = W*H bytes, pixels with 0 there are skipped (mask 0). */
static int grazing_bits(const rt_ctx* c, const Ray* ray, const HitInfo* win) {
    int bits = 0;
    const float t1 = win->dst;
    if (win->didHit && win->sphere < 0) {
        const Triangle* tri = &c->tris[win->facetId];
        v3 v0 = V(tri->posA[0], tri->posA[1], tri->posA[2]), v1 = V(tri->posB[0], tri->posB[1], tri->posB[2]);
        v3 v2 = V(tri->posC[0], tri->posC[1], tri->posC[2]);
        v3 N = cross(sub(v1, v0), sub(v2, v0));
        float nn = dot(N, N), nd = dot(N, ray->dir);
        v3 P = madd(ray->origin, t1, ray->dir);
        float e0 = dot(N, cross(sub(v1, v0), sub(P, v0))), e1 = dot(N, cross(sub(v2, v1), sub(P, v1)));
        float e2 = dot(N, cross(sub(v0, v2), sub(P, v2)));
        if (fabsf(e0) <= 1e-6f * nn || fabsf(e1) <= 1e-6f * nn || fabsf(e2) <= 1e-6f * nn) bits |= 2;
        if (fabsf(nd) < 1e-5f * sqrtf(nn)) bits |= 4;
    }
    HitInfo h;
    const float tol = 1e-5f * fmaxf(1.0f, t1);
    for (int i = 0; i < c->n_spheres; i++) {
        if (win->didHit && win->sphere == i) continue;
        raySphere(ray->origin, ray->dir, &c->spheres[i], &h);
        if (h.didHit && !(h.dst < 1e-6f) && win->didHit && fabsf(h.dst - t1) <= tol) bits |= 1;
    }
    /* every triangle, whatever the boxes say: a near-tie hidden behind a box that one reading of
     * the slab arithmetic passes and another fails is exactly what the mask is for */
    for (int k = 0; k < c->n_tris; k++) {
        const Triangle* tri = &c->tris[k];
        if (win->didHit && win->sphere < 0 && tri->triId == win->facetId) continue;
        int hitBehind = !c->meshes[tri->meshIndex].material.transparentFromBehind;   /* primary rays: not in a solid */
        rayTriangle(ray->origin, ray->dir, tri, hitBehind, &h);
        if (win->didHit) {
            if (h.didHit && !(h.dst < 1e-6f) && fabsf(h.dst - t1) <= tol) bits |= 1;
        } else {
            /* a miss: would a slightly larger triangle have been hit? */
            v3 v0 = V(tri->posA[0], tri->posA[1], tri->posA[2]), v1 = V(tri->posB[0], tri->posB[1], tri->posB[2]);
            v3 v2 = V(tri->posC[0], tri->posC[1], tri->posC[2]);
            v3 N = cross(sub(v1, v0), sub(v2, v0));
            float nn = dot(N, N), nd = dot(N, ray->dir);
            if (fabsf(nd) < 1e-6f || (nd > 0.0f && !hitBehind)) continue;
            float t = -(dot(N, ray->origin) + (-dot(N, v0))) / nd;
            if (!(t >= 1e-6f)) continue;
            v3 P = madd(ray->origin, t, ray->dir);
            float e0 = dot(N, cross(sub(v1, v0), sub(P, v0))), e1 = dot(N, cross(sub(v2, v1), sub(P, v1)));
            float e2 = dot(N, cross(sub(v0, v2), sub(P, v2)));
            const float lim = -1e-6f * nn;
            if (e0 >= lim && e1 >= lim && e2 >= lim) bits |= 8;
        }
    }
    return bits;
}

int rto_first_hit_grazing(rt_ctx* c, uint32_t pass, uint8_t* mask_out, const uint8_t* only) {
    if (!c || !mask_out) return RT_E_ARG;
    if (!ready(c)) return RT_E_STATE;
    const int W = c->w, H = c->h;
#pragma omp parallel for schedule(dynamic, 1)
    for (int py = 0; py < H; py++) {
        Counters k = {0, 0, 0, 0};
        for (int px = 0; px < W; px++) {
            if (only && !only[(size_t)py * W + px]) { mask_out[(size_t)py * W + px] = 0; continue; }
            uint32_t rng = pixelSeed(c, px, py, pass);
            Ray ray;
            HitInfo h;
            newSampleRay(c, viewPointOf(c, px, py), &rng, &ray);
            calculateRayCollision(c, &ray, &h, &k);
            mask_out[(size_t)py * W + px] = (uint8_t)grazing_bits(c, &ray, &h);
        }
    }
    return RT_OK;
}

int rt_read_accum(rt_ctx* c, float* out, uint64_t* n) {
    if (!c || !out) return RT_E_ARG;
    if (!c->accum) FAIL(c, RT_E_STATE, "no image size");
    memcpy(out, c->accum, (size_t)c->w * c->h * 12);
    if (n) *n = c->n_accum;
    return RT_OK;
}
/* SUM accumulators are divided by the number of passes they hold (true division) */
int rt_read_mean(rt_ctx* c, float* out) {
    if (!c || !out) return RT_E_ARG;
    if (!c->accum) FAIL(c, RT_E_STATE, "no image size");
    size_t n = (size_t)c->w * c->h * 3;
    memcpy(out, c->accum, n * 4);
    if (c->sum_mode && c->n_accum > 0) {
        float d = (float)c->n_accum;
        for (size_t i = 0; i < n; i++) out[i] = out[i] / d;
    }
    return RT_OK;
}

int rt_enable_stats(rt_ctx* c, int e) { (void)e; return c ? RT_OK : RT_E_ARG; }
int rt_reset_stats(rt_ctx* c) { if (!c) return RT_E_ARG; memset(&c->st, 0, sizeof c->st); return RT_OK; }
int rt_get_stats(rt_ctx* c, rt_stats_t* o) { if (!c || !o) return RT_E_ARG; *o = c->st; return RT_OK; }
/* the oracle walks the uploaded trees with a plain stack: only the scene counts mean anything here */
int rt_get_info(rt_ctx* c, rt_info_t* o) {
    if (!c || !o) return RT_E_ARG;
    memset(o, 0, sizeof(*o));
    if (!c->have_scene) FAIL(c, RT_E_STATE, "no scene");
    o->n_slots = c->n_tris;
    o->n_records = c->n_nodes;
    return RT_OK;
}
int rt_sync(rt_ctx* c) { return c ? RT_OK : RT_E_ARG; }
/* the oracle is a single-process checker: there is nothing to combine with */
int rt_allreduce_sum(rt_ctx* c, void* comm, uint64_t total) { (void)comm; (void)total; if (!c) return RT_E_ARG; FAIL(c, RT_E_STATE, "the oracle has no collective"); }
int rt_set_cull(rt_ctx* c, int e) { (void)e; return c ? RT_OK : RT_E_ARG; } /* the oracle never culls */
int rt_set_accel(rt_ctx* c, int accel) { (void)accel; return c ? RT_OK : RT_E_ARG; }   /* one structure only: the reference's */
int rt_set_stream(rt_ctx* c, void* s) { (void)s; return c ? RT_OK : RT_E_ARG; }
void* rt_stream(rt_ctx* c) { (void)c; return NULL; }

/* mode 0 off, 1 count, 2 count and print each case; out (may be NULL) receives the five counters, which are then reset */
int rto_cull_audit(int mode, unsigned long long out[5]) {
    if (out) { memcpy(out, g_audit_n, sizeof g_audit_n); memset(g_audit_n, 0, sizeof g_audit_n); }
    g_audit = mode;
    return RT_OK;
}

/* single-function probes for the known-answer tests ------------------------------------ */
float rto_random_value(uint32_t* state) { return randomValue(state); }
void rto_random_direction(uint32_t* state, float out[3]) { v3 d = randomDirection(NULL, state); out[0] = d.x; out[1] = d.y; out[2] = d.z; }
int rto_box_hit(const float o[3], const float d[3], const float lo[3], const float hi[3], int true_div) {
    rt_ctx c; memset(&c, 0, sizeof c); c.true_div = true_div;
    v3 dd = V(d[0], d[1], d[2]);
    return boxHit(&c, V(o[0], o[1], o[2]), dd, V(1.0f / dd.x, 1.0f / dd.y, 1.0f / dd.z), lo, hi);
}
/* out = {didHit, dst, hitBackside, nx, ny, nz, px, py, pz} */
void rto_ray_triangle(const float o[3], const float d[3], const float tri28[28], int hitBehind, float out[9]) {
    HitInfo h;
    rayTriangle(V(o[0], o[1], o[2]), V(d[0], d[1], d[2]), (const Triangle*)tri28, hitBehind, &h);
    out[0] = (float)h.didHit; out[1] = h.dst; out[2] = (float)h.hitBackside;
    out[3] = h.normal.x; out[4] = h.normal.y; out[5] = h.normal.z;
    out[6] = h.hitPoint.x; out[7] = h.hitPoint.y; out[8] = h.hitPoint.z;
}
/* primary ray of (px,py,pass): out = {ox,oy,oz,dx,dy,dz}, and the rng state after the 4 draws */
int rto_primary_ray(rt_ctx* c, int px, int py, uint32_t pass, float out[6], uint32_t* state_out) {
    if (!c || !c->have_cam || c->w <= 0) return RT_E_STATE;
    uint32_t rng = pixelSeed(c, px, py, pass);
    Ray ray;
    newSampleRay(c, viewPointOf(c, px, py), &rng, &ray);
    out[0] = ray.origin.x; out[1] = ray.origin.y; out[2] = ray.origin.z;
    out[3] = ray.dir.x; out[4] = ray.dir.y; out[5] = ray.dir.z;
    if (state_out) *state_out = rng;
    return RT_OK;
}
/* radiance of ONE pass of one pixel (no accumulation) */
int rto_pixel_radiance(rt_ctx* c, int px, int py, uint32_t pass, int rpp, int maxBounces, float out[3]) {
    if (!c || !c->have_cam || !c->have_scene || c->w <= 0) return RT_E_STATE;
    Counters k = {0, 0, 0, 0};
    uint32_t rng = pixelSeed(c, px, py, pass);
    v3 vp = viewPointOf(c, px, py), total = V(0, 0, 0);
    for (int s = 0; s < rpp; s++) {
        Ray ray;
        newSampleRay(c, vp, &rng, &ray);
        total = add(total, traceRay(c, &ray, &rng, maxBounces, &k));
    }
    out[0] = total.x / (float)rpp; out[1] = total.y / (float)rpp; out[2] = total.z / (float)rpp;
    return RT_OK;
}
int rto_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * BVH construction / refit - restatement of scripts/scenes/chunker.py of the reference
 * (chunk_mesh_bvh :17-25; BVHParentNode.__init__/split/split_recursively :113-195: node ids in
 * creation order, centroid = mean of the three corners frozen at construction, longest axis
 * halved, "inside the lower half-box, inclusive" goes left, no split when one side is empty,
 * stop at <= 5 triangles or depth 20; refit :197-214 = min/max over ALL of a node's own
 * triangles; register_all_and_update_node_tri_ids :79-86; std140 image :247-257).
 * Plain recursive C, every node keeps its full triangle list like the reference does.
 * ------------------------------------------------------------------------------------------- */
typedef struct { float lo[3], pad, hi[3]; int32_t id, c1, c2, off, cnt; } ONode48;
typedef struct ONode { int *tris; int n, depth, left, right; float lo[3], hi[3]; } ONode;
typedef struct { const float* pos; ONode* nodes; int n_nodes, cap; } OBuild;

static void onode_box(const float* pos, const int* tris, int n, float lo[3], float hi[3]) {
    if (n == 0) { for (int c = 0; c < 3; c++) lo[c] = hi[c] = 0.0f; return; }
    for (int c = 0; c < 3; c++) { lo[c] = INFINITY; hi[c] = -INFINITY; }
    for (int i = 0; i < n; i++)
        for (int v = 0; v < 3; v++)
            for (int c = 0; c < 3; c++) {
                float x = pos[9 * (size_t)tris[i] + 3 * v + c];
                lo[c] = x < lo[c] ? x : lo[c];
                hi[c] = x > hi[c] ? x : hi[c];
            }
}
static int onode_new(OBuild* b, int* tris, int n, int depth) {
    if (b->n_nodes == b->cap) { b->cap *= 2; b->nodes = (ONode*)realloc(b->nodes, (size_t)b->cap * sizeof(ONode)); }
    ONode* nd = &b->nodes[b->n_nodes];
    nd->tris = tris; nd->n = n; nd->depth = depth; nd->left = nd->right = -1;
    onode_box(b->pos, tris, n, nd->lo, nd->hi);
    return b->n_nodes++;
}
static void onode_split_recursively(OBuild* b, int ni) {
    if (b->nodes[ni].n <= 5 || b->nodes[ni].depth == 20) return;
    ONode nd = b->nodes[ni];
    int axis = 0;
    for (int c = 1; c < 3; c++) if (fabsf(nd.lo[c] - nd.hi[c]) > fabsf(nd.lo[axis] - nd.hi[axis])) axis = c;
    float half[3] = {0.0f, 0.0f, 0.0f}, lh[3];
    half[axis] = 0.5f;
    for (int c = 0; c < 3; c++) lh[c] = nd.hi[c] - (nd.hi[c] - nd.lo[c]) * half[c];
    int* l = (int*)malloc((size_t)nd.n * sizeof(int));
    int* r = (int*)malloc((size_t)nd.n * sizeof(int));
    int nl = 0, nr = 0;
    for (int i = 0; i < nd.n; i++) {
        const float* p = b->pos + 9 * (size_t)nd.tris[i];
        int in = 1;
        for (int c = 0; c < 3; c++) {
            float cen = ((p[c] + p[3 + c]) + p[6 + c]) / 3.0f;     /* np.mean(vertices, axis=1), float32 */
            if (!(cen >= nd.lo[c] && cen <= lh[c])) in = 0;
        }
        if (in) l[nl++] = nd.tris[i]; else r[nr++] = nd.tris[i];
    }
    if (nl == 0 || nr == 0) { free(l); free(r); return; }
    int li = onode_new(b, l, nl, nd.depth + 1);
    int ri = onode_new(b, r, nr, nd.depth + 1);
    b->nodes[ni].left = li; b->nodes[ni].right = ri;
    onode_split_recursively(b, li);
    onode_split_recursively(b, ri);
}
int rt_bvh_build(const float* pos9, int n_tris, int32_t first_node_id, int32_t first_tri_id, int32_t first_leaf_offset,
                 void* nodes48_out, int nodes_cap, int32_t* leaf_tri_ids_out, int32_t* n_nodes_out) {
    if (!pos9 || n_tris <= 0 || !nodes48_out || !leaf_tri_ids_out || !n_nodes_out || nodes_cap <= 0) return RT_E_ARG;
    OBuild b = {pos9, (ONode*)malloc(64 * sizeof(ONode)), 0, 64};
    int* all = (int*)malloc((size_t)n_tris * sizeof(int));
    for (int i = 0; i < n_tris; i++) all[i] = i;
    onode_split_recursively(&b, onode_new(&b, all, n_tris, 0));
    int rc = RT_OK;
    if (b.n_nodes > nodes_cap) rc = RT_E_LIMIT;
    else {
        ONode48* out = (ONode48*)nodes48_out;
        int32_t off = first_leaf_offset;
        size_t w = 0;
        for (int i = 0; i < b.n_nodes; i++) {
            ONode* nd = &b.nodes[i];
            memset(&out[i], 0, sizeof(ONode48));
            for (int c = 0; c < 3; c++) { out[i].lo[c] = nd->lo[c]; out[i].hi[c] = nd->hi[c]; }
            out[i].id = first_node_id + i;
            out[i].c1 = nd->left < 0 ? -1 : first_node_id + nd->left;
            out[i].c2 = nd->right < 0 ? -1 : first_node_id + nd->right;
            out[i].cnt = nd->n;
            out[i].off = -1;
            if (nd->left < 0 && nd->right < 0) {
                out[i].off = off;
                for (int k = 0; k < nd->n; k++) leaf_tri_ids_out[w++] = first_tri_id + nd->tris[k];
                off += nd->n;
            }
        }
        *n_nodes_out = b.n_nodes;
    }
    for (int i = 0; i < b.n_nodes; i++) free(b.nodes[i].tris);
    free(b.nodes);
    return rc;
}
/* refit as the reference does it: every node over all triangles below it (here: gathered by
 * walking down to its leaves), not bottom-up */
static void orefit_gather(const float* pos9, const ONode48* nodes, int32_t id0, int i, int32_t first_tri_id, int32_t leaf0,
                          const int32_t* ids, float lo[3], float hi[3]) {
    const ONode48* nd = &nodes[i];
    if (nd->c1 < 0 && nd->c2 < 0) {
        for (int k = 0; k < nd->cnt; k++) {
            const float* p = pos9 + 9 * (size_t)(ids[nd->off - leaf0 + k] - first_tri_id);
            for (int q = 0; q < 9; q++) { int c = q % 3; lo[c] = p[q] < lo[c] ? p[q] : lo[c]; hi[c] = p[q] > hi[c] ? p[q] : hi[c]; }
        }
        return;
    }
    if (nd->c1 >= 0) orefit_gather(pos9, nodes, id0, nd->c1 - id0, first_tri_id, leaf0, ids, lo, hi);
    if (nd->c2 >= 0) orefit_gather(pos9, nodes, id0, nd->c2 - id0, first_tri_id, leaf0, ids, lo, hi);
}
int rt_bvh_refit(const float* pos9, int n_tris, int32_t first_tri_id, int32_t first_leaf_offset, void* nodes48, int n_nodes,
                 const int32_t* leaf_tri_ids) {
    if (!pos9 || n_tris <= 0 || !nodes48 || n_nodes <= 0 || !leaf_tri_ids) return RT_E_ARG;
    ONode48* nodes = (ONode48*)nodes48;
    for (int i = 0; i < n_nodes; i++) {
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        orefit_gather(pos9, nodes, nodes[0].id, i, first_tri_id, first_leaf_offset, leaf_tri_ids, lo, hi);
        for (int c = 0; c < 3; c++) { nodes[i].lo[c] = lo[c]; nodes[i].hi[c] = hi[c]; }
    }
    return RT_OK;
}
