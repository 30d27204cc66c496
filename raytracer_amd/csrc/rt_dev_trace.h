// rt_dev_trace.h - device data layout, closest-hit query and the bounce loop (gfx950).
//
// Restates, for one ray per lane of a 64-wide wavefront:
//   boundingBoxIntersect   shaders/raytracer.fs.glsl:421-453
//   rayTriangle            :305-409  (plane + three edge tests; the per-triangle constants
//                                     N = cross(B-A, C-A) and d = -dot(N,A) are precomputed
//                                     by the scene-prep kernel with the same expression tree)
//   raySphere              :251-276
//   calculateRayCollision  :518-641
//   traceRay               :691-915
//
// Data layout in HBM (built by rt_set_scene, refreshed by the prep kernels):
//   tri_geo  4 x float4 per triangle SLOT (64 B, one half cache line):
//              {A.xyz, N.x} {B.xyz, N.y} {C.xyz, N.z} {d, bits(rem), bits(triId), 0}
//            slots are in traversal order of the reference (mesh ascending, depth-first,
//            second child first, leaf order), so (a) every leaf and every subtree is one
//            contiguous range, (b) the slot index is the rank that decides ties exactly as
//            the reference's "first found wins" (strict <, :625).  rem = triangles left in
//            the leaf after this one, so a leaf is walked with no separate count fetch.
//   tri_nrm  3 x float4 per slot (corner normals), touched only for the winning triangle.
//   nodes    4 x float4 per INTERIOR node: both children's boxes and references
//              {lo1.xyz, bits(ref1)} {hi1.xyz, bits(ref2)} {lo2.xyz, 0} {hi2.xyz, 0}
//            ref >= 0: interior record index; ref < 0 and != RT_NONE: leaf, first slot = ~ref.
//   roots    2 x float4 per mesh: {lo.xyz, bits(ref)} {hi.xyz, 0} of the mesh's root node.
//   mats     the 80-byte std140 Material of every mesh, as uploaded.
#pragma once
#include "rt_dev_math.h"

#define RT_NONE ((int)0x80000000)
#define RT_BLOCK 256
#define RT_STACK 32

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
    const float4* tri_nrm;
    const float4* nodes;
    const float4* roots;
    const MatDev* mats;
    const SphereDev* spheres;
    int n_meshes;
    int n_spheres;
};

struct DevCamera {
    float view_params[3];
    float l2w[16];
    float cam_pos[3];
    float dof, aa;
    int width, height;
};

struct Counters { unsigned long long rays, box, tri, hits; };

#define RT_SPHERE_RANK (-(1 << 30))

struct Hit {
    float t;    // 1e20 = miss
    int rank;   // slot index of the triangle; RT_SPHERE_RANK + i for sphere i; INT_MAX = miss
};

// the reference's slab test, comparison for comparison (NaN compares false, no t >= 0 test)
__device__ __forceinline__ bool box_hit(v3 o, v3 inv, float4 lo, float4 hi) {
    float tmin = (lo.x - o.x) * inv.x, tmax = (hi.x - o.x) * inv.x, t;
    if (tmin > tmax) { t = tmin; tmin = tmax; tmax = t; }
    float tymin = (lo.y - o.y) * inv.y, tymax = (hi.y - o.y) * inv.y;
    if (tymin > tymax) { t = tymin; tymin = tymax; tymax = t; }
    if ((tmin > tymax) || (tymin > tmax)) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo.z - o.z) * inv.z, tzmax = (hi.z - o.z) * inv.z;
    if (tzmin > tzmax) { t = tzmin; tzmin = tzmax; tzmax = t; }
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    return true;
}

// Closest hit over spheres and all mesh BVHs.  prev_id = triId of the triangle the ray
// left (-1 none), prev_mesh / in_solid as in Ray.prevHit / Ray.inSolid.
template <bool STATS>
__device__ __forceinline__ Hit closest_hit(const DevScene& S, v3 o, v3 d, int prev_id, int prev_mesh, bool in_solid,
                                           int* stack /* LDS, this lane's column, stride RT_BLOCK */, Counters& cnt) {
    Hit best;
    best.t = 1e20f;
    best.rank = 0x7fffffff;
    if (STATS) cnt.rays++;
    const v3 inv = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);

    for (int i = 0; i < S.n_spheres; i++) {                     // :525-541
        const SphereDev& sp = S.spheres[i];
        v3 c = V3(sp.position[0], sp.position[1], sp.position[2]);
        v3 oo = o - c;
        float a = dot3(d, d);
        float b = 2.0f * dot3(oo, d);
        float cc = dot3(oo, oo) - sp.radius * sp.radius;
        float disc = b * b - (4.0f * a) * cc;
        if (disc >= 0.0f) {
            float dst = (-b - __builtin_sqrtf(disc)) / (2.0f * a);
            if (dst >= 0.0f && !(dst < 1e-6f) && dst < best.t) { best.t = dst; best.rank = RT_SPHERE_RANK + i; }
        }
    }

    int sp = 0;          // stack entries of this lane
    int j = 0;           // next mesh whose root has not been started
    int cur = RT_NONE;   // reference (interior >= 0 / leaf < 0) being visited
    bool hb = false;     // hitBehind for the mesh being walked
    // Stack entries carry no mesh tag: a mesh's subtree is finished (stack empty) before
    // the next root is started, so hb always belongs to `cur`.
    while (true) {
        if (cur == RT_NONE) {
            if (sp > 0) {
                cur = stack[(--sp) * RT_BLOCK];
            } else {
                if (j >= S.n_meshes) break;
                float4 r0 = S.roots[2 * j], r1 = S.roots[2 * j + 1];
                int ref = __float_as_int(r0.w);
                // hitBehind for triangles of mesh j (:605-615; in a solid: same mesh only [9.6])
                hb = in_solid ? (prev_mesh == j) : (S.mats[j].transparentFromBehind == 0);
                if (ref < 0 && ref != RT_NONE) {                  // root is a leaf: the reference tests its own box (:586)
                    if (STATS) cnt.box++;
                    if (!box_hit(o, inv, r0, r1)) ref = RT_NONE;
                }
                cur = ref;
                j++;
                continue;
            }
        }
        if (cur >= 0) {                                          // interior: test both children (:562-584)
            const float4* n = S.nodes + 4 * (size_t)cur;
            float4 n0 = n[0], n1 = n[1], n2 = n[2], n3 = n[3];
            int ref1 = __float_as_int(n0.w), ref2 = __float_as_int(n1.w);
            bool h1 = false, h2 = false;
            if (ref1 != RT_NONE) { if (STATS) cnt.box++; h1 = box_hit(o, inv, n0, n1); }
            if (ref2 != RT_NONE) { if (STATS) cnt.box++; h2 = box_hit(o, inv, n2, n3); }
            if (h1 && h2) { stack[(sp++) * RT_BLOCK] = ref1; cur = ref2; }
            else if (h2) cur = ref2;
            else if (h1) cur = ref1;
            else cur = RT_NONE;
            continue;
        }
        // leaf: walk its slots (:593-630)
        int k = ~cur;
        int rem;
        do {
            const float4* g = S.tri_geo + 4 * (size_t)k;
            float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
            rem = __float_as_int(g3.y);
            if (__float_as_int(g3.z) != prev_id) {
                if (STATS) cnt.tri++;
                v3 A = V3(g0.x, g0.y, g0.z), B = V3(g1.x, g1.y, g1.z), C = V3(g2.x, g2.y, g2.z);
                v3 N = V3(g0.w, g1.w, g2.w);
                float nd = dot3(N, d);
                bool ok = !(__builtin_fabsf(nd) < 1e-6f) && !(nd > 0.0f && !hb);
                float t = -(dot3(N, o) + g3.x) / nd;
                ok = ok && !(t < 0.0f) && !(t < 1e-6f);
                // closer than (or tied with a later-ranked) current best; NaN t never wins
                ok = ok && (t < best.t || (t == best.t && k < best.rank));
                if (ok) {
                    v3 P = madd3(o, t, d);
                    v3 c0 = cross3(B - A, P - A);
                    if (!(dot3(N, c0) < 0.0f)) {
                        v3 c1 = cross3(C - B, P - B);
                        if (!(dot3(N, c1) < 0.0f)) {
                            v3 c2 = cross3(A - C, P - C);
                            if (!(dot3(N, c2) < 0.0f)) { best.t = t; best.rank = k; }
                        }
                    }
                }
            }
            k++;
        } while (rem > 0);
        cur = RT_NONE;
    }
    return best;
}

struct PathState {
    v3 origin, dir;
    int prev_id, prev_mesh;      // Ray.prevHit.facetId / meshIndex
    float prev_color[4];         // Ray.prevHit.material.color
    float prev_transmission;     // Ray.prevHit.material.transmission
    float ior;
    bool in_solid;
};

// camera sample (raytracer.fs.glsl:972-987); four draws: DoF x,y then AA x,y
__device__ __forceinline__ void new_sample_ray(const DevCamera& cam, const MatDev& atmo, v3 view_point, uint32_t& rng, PathState& r) {
    float r1 = random_value(rng), r2 = random_value(rng), r3 = random_value(rng), r4 = random_value(rng);
    float nearp = cam.view_params[2];
    v3 dofo = V3(((r1 * 2.0f - 1.0f) * cam.dof) * nearp, ((r2 * 2.0f - 1.0f) * cam.dof) * nearp,
                 ((0.0f * 2.0f - 1.0f) * cam.dof) * nearp);
    v3 aao = V3((r3 * 2.0f - 1.0f) * cam.aa, (r4 * 2.0f - 1.0f) * cam.aa, (0.0f * 2.0f - 1.0f) * cam.aa);
    r.dir = normalize3((view_point + dofo) + aao);
    r.origin = V3(cam.cam_pos[0], cam.cam_pos[1], cam.cam_pos[2]) - dofo;
    r.prev_id = -1;
    r.prev_mesh = -1;
    r.prev_color[0] = atmo.color[0]; r.prev_color[1] = atmo.color[1];
    r.prev_color[2] = atmo.color[2]; r.prev_color[3] = atmo.color[3];
    r.prev_transmission = atmo.transmission;
    r.ior = atmo.ior;
    r.in_solid = false;
}

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

// One camera sample: the bounce loop of traceRay.  Returns incoming light.
template <bool STATS>
__device__ __forceinline__ v3 trace_path(const DevScene& S, const MatDev& atmo, PathState& ray, uint32_t& rng,
                                         int max_bounces, int* stack, Counters& cnt) {
    v3 incoming = V3(0.0f, 0.0f, 0.0f), color = V3(1.0f, 1.0f, 1.0f);
    for (int i = 0; i <= max_bounces; i++) {
        Hit h = closest_hit<STATS>(S, ray.origin, ray.dir, ray.prev_id, ray.prev_mesh, ray.in_solid, stack, cnt);
        if (h.rank == 0x7fffffff) break;                          // miss: environment light is vec3(0) (:279-282)
        if (STATS) cnt.hits++;
        // ---- rebuild the HitInfo of the winner -------------------------------------------------
        v3 hit_point = madd3(ray.origin, h.t, ray.dir);
        v3 normal;
        bool backside = false;
        int hit_id = -1, hit_mesh = -1;
        const MatDev* m;
        if (h.rank < 0) {                                         // sphere
            const SphereDev& sp = S.spheres[h.rank - RT_SPHERE_RANK];
            normal = normalize3(hit_point - V3(sp.position[0], sp.position[1], sp.position[2]));
            m = &sp.material;
        } else {
            const float4* g = S.tri_geo + 4 * (size_t)h.rank;
            float4 g0 = g[0], g1 = g[1], g2 = g[2], g3 = g[3];
            v3 A = V3(g0.x, g0.y, g0.z), B = V3(g1.x, g1.y, g1.z), C = V3(g2.x, g2.y, g2.z), N = V3(g0.w, g1.w, g2.w);
            backside = dot3(N, ray.dir) > 0.0f;
            float area = length3(N) / 2.0f;
            float w = length3(cross3(B - A, hit_point - A)) / 2.0f / area;
            float u = length3(cross3(C - B, hit_point - B)) / 2.0f / area;
            float v = 1.0f - u - w;
            const float4* nn = S.tri_nrm + 3 * (size_t)h.rank;
            float4 na = nn[0], nb = nn[1], nc = nn[2];
            normal = V3(na.x * u + nb.x * v + nc.x * w, na.y * u + nb.y * v + nc.y * w, na.z * u + nb.z * v + nc.z * w);
            hit_id = __float_as_int(g3.z);
            hit_mesh = __float_as_int(na.w);
            m = &S.mats[hit_mesh];
        }
        // ---- material response (:720-898) -------------------------------------------------------
        float alpha = m->color[3];
        if (alpha < 0.99999f) {
            if (random_value(rng) > alpha) {                      // pass straight through
                if (i == max_bounces) break;
                ray.origin = hit_point;
                ray.prev_id = hit_id; ray.prev_mesh = hit_mesh;
                ray.prev_color[0] = m->color[0]; ray.prev_color[1] = m->color[1];
                ray.prev_color[2] = m->color[2]; ray.prev_color[3] = alpha;
                ray.prev_transmission = m->transmission;
                continue;
            }
        }
        float es = m->emissionStrength;
        incoming = incoming + V3(m->emissionColor[0] * es, m->emissionColor[1] * es, m->emissionColor[2] * es) * color;

        float nf = backside ? -1.0f : 1.0f;
        v3 nrm = normal * nf;
        float n2 = (ray.in_solid && backside) ? atmo.ior : m->ior;
        float n1 = ray.ior;
        v3 diffuse = normalize3((-nrm) + random_direction(rng));
        float costheta = clampf(dot3(normalize3(nrm), normalize3(-ray.dir)), -1.0f, 1.0f);
        float r0 = (n1 - n2) / (n1 + n2);
        r0 = r0 * r0;
        float om = 1.0f - costheta;
        float om5 = ((om * om) * (om * om)) * om;
        float schlick = r0 + (1.0f - r0) * om5;

        bool specular = random_value(rng) < m->specularStrength;
        v3 spec_dir = V3(0.0f, 0.0f, 0.0f);
        if (specular) {
            spec_dir = reflect3(ray.dir, nrm);
            diffuse = -diffuse;
        } else if (random_value(rng) > schlick) {
            if (random_value(rng) < m->transmission) {            // transmission (:822-847)
                float a = -det_log(ray.prev_transmission);
                float e = det_exp(-a * h.t);
                v3 tc = V3(ray.prev_color[0], ray.prev_color[1], ray.prev_color[2]) * e;
                color = color * (tc * ray.prev_color[3]);
                if (backside) { ray.in_solid = false; ray.ior = atmo.ior; }
                else { ray.in_solid = true; ray.ior = m->ior; }
            } else {                                              // diffuse
                diffuse = -diffuse;
                if (backside) { ray.in_solid = true; ray.ior = m->ior; }
            }
        } else {
            specular = true;
            spec_dir = reflect3(ray.dir, nrm);
            diffuse = -diffuse;
        }
        if (i == max_bounces) break;
        ray.origin = hit_point;
        if (specular) {
            float a = m->smoothness, b = 1.0f - a;
            ray.dir = V3(diffuse.x * b + spec_dir.x * a, diffuse.y * b + spec_dir.y * a, diffuse.z * b + spec_dir.z * a);
            color = color * V3(m->specularColor[0], m->specularColor[1], m->specularColor[2]);
        } else {
            ray.dir = diffuse;
            color = color * V3(m->color[0], m->color[1], m->color[2]);
        }
        ray.prev_id = hit_id; ray.prev_mesh = hit_mesh;
        ray.prev_color[0] = m->color[0]; ray.prev_color[1] = m->color[1];
        ray.prev_color[2] = m->color[2]; ray.prev_color[3] = alpha;
        ray.prev_transmission = m->transmission;
    }
    return incoming;
}
