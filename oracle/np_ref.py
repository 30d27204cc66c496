"""np_ref.py - second, independent restatement of the shader path (scalar numpy float32).

TEST INFRASTRUCTURE ONLY (like everything under oracle/).  PARITY UNPINNED by the reference:
it has no tests or golden vectors; two independently written restatements
(this file and rt_oracle.c) agreeing bit-for-bit is the substitute the build uses.

Written from shaders/raytracer.fs.glsl directly (line numbers below), one numpy float32
operation per GLSL operation so that every intermediate is rounded to binary32, with the
arithmetic contract of DESIGN.md (explicit fma in dot/cross/o+t*d, a*(1/b) in the slab
test, the fixed polynomial sin/cos/acos/exp/log).  Pure-Python loops: small cases only.
It consumes the std140 buffers exactly as uploaded (export_scene_buffers).
"""
from __future__ import annotations

from fractions import Fraction

import numpy as np

F = np.float32
U32 = 0xFFFFFFFF


def fma(a, b, c) -> np.float32:
    """Correctly rounded float32 a*b+c (exact rational arithmetic, ties to even)."""
    a, b, c = F(a), F(b), F(c)
    if not (np.isfinite(a) and np.isfinite(b) and np.isfinite(c)):
        with np.errstate(all="ignore"):
            return F(np.float64(a) * np.float64(b) + np.float64(c))
    exact = Fraction(float(a)) * Fraction(float(b)) + Fraction(float(c))
    if exact == 0:
        with np.errstate(all="ignore"):
            return F(np.float64(a) * np.float64(b) + np.float64(c))   # keeps the sign of zero
    r = F(float(exact))
    if not np.isfinite(r):
        return r
    best, best_err = r, abs(exact - Fraction(float(r)))
    for cand in (np.nextafter(r, F(np.inf)), np.nextafter(r, F(-np.inf))):
        if not np.isfinite(cand):
            continue
        err = abs(exact - Fraction(float(cand)))
        if err < best_err or (err == best_err and (int(cand.view(np.uint32)) & 1) == 0 and (int(best.view(np.uint32)) & 1)):
            best, best_err = cand, err
    return best


def vec(x, y, z):
    return np.array([x, y, z], dtype=F)


def dot(a, b):
    return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))


def cross(a, b):
    return vec(fma(a[1], b[2], -(a[2] * b[1])), fma(a[2], b[0], -(a[0] * b[2])), fma(a[0], b[1], -(a[1] * b[0])))


def length(a):
    return np.sqrt(dot(a, a))


def normalize(a):
    with np.errstate(all="ignore"):
        return a * (F(1.0) / length(a))


def madd(o, t, d):
    return vec(fma(t, d[0], o[0]), fma(t, d[1], o[1]), fma(t, d[2], o[2]))


# ---- fixed elementary functions (same published coefficient sets as rt_oracle.c) ----------
def det_sincos(x):
    x = F(x)
    q = np.rint(x * F(float.fromhex("0x1.45f306p-1")))
    r = fma(q, F(float.fromhex("-0x1.921fb6p+0")), x)
    r = fma(q, F(float.fromhex("0x1.777a5cp-25")), r)
    r = fma(q, F(float.fromhex("0x1.ee59dap-50")), r)
    z = r * r
    ps = fma(fma(F(-1.9515295891e-4), z, F(8.3321608736e-3)), z, F(-1.6666654611e-1))
    s = fma(r * z, ps, r)
    pc = fma(fma(F(2.443315711809948e-5), z, F(-1.388731625493765e-3)), z, F(4.166664568298827e-2))
    c = fma(z * z, pc, fma(F(-0.5), z, F(1.0)))
    n = int(q) & 3
    ss, cc = (c, s) if n & 1 else (s, c)
    if n in (2, 3):
        ss = -ss
    if n in (1, 2):
        cc = -cc
    return F(ss), F(cc)


def _asin_poly(a, z):
    p = fma(F(4.2163199048e-2), z, F(2.4181311049e-2))
    p = fma(p, z, F(4.5470025998e-2))
    p = fma(p, z, F(7.4953002686e-2))
    p = fma(p, z, F(1.6666752422e-1))
    return fma(a * z, p, a)


def det_acos(x):
    x = F(x)
    ax = np.abs(x)
    if ax > F(0.5):
        z = F(0.5) * (F(1.0) - ax)
        w = np.sqrt(z)
        r = F(2.0) * _asin_poly(w, z)
        return r if x > 0 else F(float.fromhex("0x1.921fb6p+1")) - r
    return F(float.fromhex("0x1.921fb6p+0")) - _asin_poly(x, x * x)


def _pow2(k):
    return np.array([(k + 127) << 23], dtype=np.uint32).view(F)[0]


def det_exp(x):
    x = F(x)
    if x != x:
        return x
    if x > F(88.72283):
        return F(np.inf)
    if x < F(-103.97):
        return F(0.0)
    k = np.rint(x * F(float.fromhex("0x1.715476p+0")))
    r = fma(k, F(float.fromhex("-0x1.62e43p-1")), x)
    r = fma(k, F(float.fromhex("0x1.05c61p-29")), r)
    p = fma(F(1.9875691500e-4), r, F(1.3981999507e-3))
    for c in (8.3334519073e-3, 4.1665795894e-2, 1.6666665459e-1, 5.0000001201e-1):
        p = fma(p, r, F(c))
    y = fma(p, r * r, r) + F(1.0)
    ki = int(k)
    k1 = int(ki / 2)          # C division truncates toward zero
    k2 = ki - k1
    return (y * _pow2(k1)) * _pow2(k2)


def det_log(x):
    x = F(x)
    if x != x or x < 0:
        return F(np.nan)
    if x == 0:
        return F(-np.inf)
    if x == F(np.inf):
        return x
    e = 0
    if x < F(float.fromhex("0x1p-126")):
        x = x * F(float.fromhex("0x1p23"))
        e = -23
    u = int(np.array([x], dtype=F).view(np.uint32)[0])
    e += (u >> 23) - 126
    m = np.array([(u & 0x007FFFFF) | 0x3F000000], dtype=np.uint32).view(F)[0]
    if m < F(0.70710678118654752440):
        e -= 1
        m = (m + m) - F(1.0)
    else:
        m = m - F(1.0)
    z = m * m
    p = fma(F(7.0376836292e-2), m, F(-1.1514610310e-1))
    for c in (1.1676998740e-1, -1.2420140846e-1, 1.4249322787e-1, -1.6668057665e-1, 2.0000714765e-1,
              -2.4999993993e-1, 3.3333331174e-1):
        p = fma(p, m, F(c))
    fe = F(e)
    y = (m * z) * p
    y = fma(F(-2.12194440e-4), fe, y)
    y = fma(F(-0.5), z, y)
    return fma(F(0.693359375), fe, m + y)


# ---- RNG (:457-492) ---------------------------------------------------------------------------
class Rng:
    def __init__(self, state: int):
        self.s = state & U32

    def value(self) -> np.float32:
        s = self.s
        f = (((s + 195439) & U32) * ((s + 124395) & U32)) & U32
        f = (f * ((s + 845921) & U32)) & U32
        s = (s * f) & U32
        self.s = s
        return F(np.float64(s)) * F(2.0 ** -32)   # float(state) rounds to nearest even, then exact scaling

    def direction(self):
        u = self.value()
        v = self.value()
        theta = F(6.2831855) * u
        phi = det_acos(F(2.0) * v - F(1.0))
        sp, cp = det_sincos(phi)
        st, ct = det_sincos(theta)
        return vec(sp * ct, sp * st, cp)


# ---- intersection (:305-453) --------------------------------------------------------------------
def box_hit(o, inv, lo, hi) -> bool:
    with np.errstate(all="ignore"):
        tmin, tmax = (lo[0] - o[0]) * inv[0], (hi[0] - o[0]) * inv[0]
        if tmin > tmax:
            tmin, tmax = tmax, tmin
        tymin, tymax = (lo[1] - o[1]) * inv[1], (hi[1] - o[1]) * inv[1]
        if tymin > tymax:
            tymin, tymax = tymax, tymin
        if tmin > tymax or tymin > tmax:
            return False
        if tymin > tmin:
            tmin = tymin
        if tymax < tmax:
            tmax = tymax
        tzmin, tzmax = (lo[2] - o[2]) * inv[2], (hi[2] - o[2]) * inv[2]
        if tzmin > tzmax:
            tzmin, tzmax = tzmax, tzmin
        if tmin > tzmax or tzmin > tmax:
            return False
        return True


def ray_triangle(o, d, tri28, hit_behind):
    """-> None or dict(dst, hitBackside, normal, hitPoint, facetId, meshIndex)"""
    v0, v1, v2 = tri28[0:3], tri28[4:7], tri28[8:11]
    n_a, n_b, n_c = tri28[12:15], tri28[16:19], tri28[20:23]
    N = cross(v1 - v0, v2 - v0)
    nd = dot(N, d)
    if np.abs(nd) < F(1e-6):
        return None
    if nd > 0 and not hit_behind:
        return None
    dd = -dot(N, v0)
    with np.errstate(all="ignore"):
        t = -(dot(N, o) + dd) / nd
    if t < 0 or t < F(1e-6):
        return None
    P = madd(o, t, d)
    area = length(N) / F(2.0)
    C = cross(v1 - v0, P - v0)
    w = length(C) / F(2.0) / area
    if dot(N, C) < 0:
        return None
    C = cross(v2 - v1, P - v1)
    u = length(C) / F(2.0) / area
    if dot(N, C) < 0:
        return None
    C = cross(v0 - v2, P - v2)
    v = F(1.0) - u - w
    if dot(N, C) < 0:
        return None
    ids = tri28.view(np.int32)
    return {
        "dst": t, "hitBackside": bool(nd > 0), "normal": n_a * u + n_b * v + n_c * w, "hitPoint": madd(o, t, d),
        "facetId": int(ids[24]), "meshIndex": int(ids[23]),
    }


class Scene:
    """std140 buffers as exported by raytracer_amd.scripts.functions.export_scene_buffers."""

    def __init__(self, bufs: dict):
        self.tris = np.ascontiguousarray(bufs["tris28"], dtype=F)
        self.meshes = bufs["meshes128"]
        self.nodes = bufs["nodes48"]
        self.leaf = bufs["leaf_tri_ids"]
        self.atmo = bufs["atmosphere80"][0]


def closest_hit(sc: Scene, o, d, prev_facet, prev_mesh, in_solid):
    """calculateRayCollision (:518-641), no spheres; node boxes are taken from nodes48 as given."""
    best = None
    with np.errstate(all="ignore"):
        inv = vec(F(1.0) / d[0], F(1.0) / d[1], F(1.0) / d[2])
    for j in range(len(sc.meshes)):
        stack = [int(sc.meshes[j]["node0Id"])]
        while stack:
            node = sc.nodes[stack.pop()]
            c1, c2 = int(node["childId1"]), int(node["childId2"])
            if c1 >= 0 and box_hit(o, inv, sc.nodes[c1]["aabbMin"][:3], sc.nodes[c1]["aabbMax"]):
                stack.append(c1)
            if c2 >= 0 and box_hit(o, inv, sc.nodes[c2]["aabbMin"][:3], sc.nodes[c2]["aabbMax"]):
                stack.append(c2)
            if c1 >= 0 or c2 >= 0:
                continue
            if not box_hit(o, inv, node["aabbMin"][:3], node["aabbMax"]):
                continue
            off, cnt = int(node["childObjOffset"]), int(node["childObjCount"])
            for i in range(cnt):
                tri = sc.tris[sc.leaf[off + i]]
                ids = tri.view(np.int32)
                if int(ids[24]) == prev_facet:
                    continue
                mesh = int(ids[23])
                if in_solid:
                    hb = prev_mesh == mesh
                else:
                    hb = not bool(sc.meshes[mesh]["material"]["transparentFromBehind"])
                h = ray_triangle(o, d, tri, hb)
                if h is not None and (best is None or h["dst"] < best["dst"]):
                    best = h
                    best["material"] = sc.meshes[mesh]["material"]
    return best


def reflect(i, n):
    s = F(2.0) * dot(n, i)
    return vec(fma(-s, n[0], i[0]), fma(-s, n[1], i[1]), fma(-s, n[2], i[2]))


def trace_ray(sc: Scene, o, d, rng: Rng, max_bounces: int):
    """traceRay (:691-915) -> incoming light (3,) float32"""
    incoming = vec(0, 0, 0)
    color = vec(1, 1, 1)
    prev_facet, prev_mesh, prev_mat = -1, -1, sc.atmo
    ior = F(sc.atmo["ior"])
    in_solid = False
    for i in range(max_bounces + 1):
        h = closest_hit(sc, o, d, prev_facet, prev_mesh, in_solid)
        if h is None:
            break
        m = h["material"]
        alpha = F(m["color"][3])
        if alpha < F(0.99999) and rng.value() > alpha:
            if i == max_bounces:
                break
            o = h["hitPoint"]
            prev_facet, prev_mesh, prev_mat = h["facetId"], h["meshIndex"], m
            continue
        incoming = incoming + (m["emissionColor"] * F(m["emissionStrength"])) * color
        nf = F(-1.0) if h["hitBackside"] else F(1.0)
        nrm = h["normal"] * nf
        n2 = F(sc.atmo["ior"]) if (in_solid and h["hitBackside"]) else F(m["ior"])
        n1 = ior
        diffuse = normalize(-nrm + rng.direction())
        ct = dot(normalize(nrm), normalize(-d))
        ct = F(-1.0) if ct < F(-1.0) else (F(1.0) if ct > F(1.0) else ct)
        r0 = (n1 - n2) / (n1 + n2)
        r0 = r0 * r0
        om = F(1.0) - ct
        om5 = ((om * om) * (om * om)) * om
        schlick = r0 + (F(1.0) - r0) * om5
        specular = bool(rng.value() < F(m["specularStrength"]))
        spec_dir = vec(0, 0, 0)
        if specular:
            spec_dir = reflect(d, nrm)
            diffuse = -diffuse
        elif rng.value() > schlick:
            if rng.value() < F(m["transmission"]):
                a = -det_log(F(prev_mat["transmission"]))
                with np.errstate(all="ignore"):
                    e = det_exp(-a * h["dst"])
                tc = prev_mat["color"][:3] * e
                color = color * (tc * F(prev_mat["color"][3]))
                if h["hitBackside"]:
                    in_solid, ior = False, F(sc.atmo["ior"])
                else:
                    in_solid, ior = True, F(m["ior"])
            else:
                diffuse = -diffuse
                if h["hitBackside"]:
                    in_solid, ior = True, F(m["ior"])
        else:
            specular = True
            spec_dir = reflect(d, nrm)
            diffuse = -diffuse
        if i == max_bounces:
            break
        o = h["hitPoint"]
        if specular:
            a = F(m["smoothness"])
            b = F(1.0) - a
            d = diffuse * b + spec_dir * a
            color = color * m["specularColor"]
        else:
            d = diffuse
            color = color * m["color"][:3]
        prev_facet, prev_mesh, prev_mat = h["facetId"], h["meshIndex"], m
    return incoming


# ---- main() (:928-989) -----------------------------------------------------------------------------
def sample_ray(cam: dict, width: int, height: int, px: int, py: int, rng: Rng):
    """One camera sample (:972-987): four draws (DoF x,y then AA x,y) -> (origin, dir)."""
    xn = (F(2.0) * (F(px) + F(0.5))) / F(width) - F(1.0)
    yn = (F(2.0) * (F(py) + F(0.5))) / F(height) - F(1.0)
    vp = cam["view_params"]
    lx, ly, lz = (xn * F(0.5)) * vp[0], (yn * F(0.5)) * vp[1], vp[2]
    M = cam["local_to_world16"].reshape(4, 4)
    view = vec(*[fma(M[i, 2], lz, fma(M[i, 1], ly, M[i, 0] * lx)) + M[i, 3] for i in range(3)])
    r1, r2, r3, r4 = rng.value(), rng.value(), rng.value(), rng.value()
    dof, aa, near = F(cam["dof"]), F(cam["aa"]), vp[2]
    dofo = vec(((r1 * F(2.0) - F(1.0)) * dof) * near, ((r2 * F(2.0) - F(1.0)) * dof) * near,
               ((F(0.0) * F(2.0) - F(1.0)) * dof) * near)
    aao = vec((r3 * F(2.0) - F(1.0)) * aa, (r4 * F(2.0) - F(1.0)) * aa, (F(0.0) * F(2.0) - F(1.0)) * aa)
    return cam["cam_pos"] - dofo, normalize((view + dofo) + aao)


def pixel_seed(width: int, px: int, py: int, pass_index: int) -> int:
    """:956-957, py counted from the bottom row"""
    return (py * width + px + pass_index * 747796405) & U32


def primary_ray(cam: dict, width: int, height: int, px: int, py: int, pass_index: int):
    rng = Rng(pixel_seed(width, px, py, pass_index))
    o, d = sample_ray(cam, width, height, px, py, rng)
    return o, d, rng


def pixel_radiance(sc: Scene, cam: dict, width: int, height: int, px: int, py: int, pass_index: int,
                   rays_per_pixel: int, max_bounces: int):
    """radiance of ONE pass of one pixel; the RNG state runs on across the samples of the pass"""
    rng = Rng(pixel_seed(width, px, py, pass_index))
    total = vec(0, 0, 0)
    for _ in range(rays_per_pixel):
        o, d = sample_ray(cam, width, height, px, py, rng)
        total = total + trace_ray(sc, o, d, rng, max_bounces)
    return total / F(rays_per_pixel)
