"""Synthetic scenes that exercise the branches the Suzanne room never takes: spheres, glass
(transmission / inSolid / ior bookkeeping), alpha pass-through, specular mix, > 32 meshes
(top-level tree over the mesh roots), depth of field, empty space (misses)."""
import numpy as np

from raytracer_amd.scripts.classes import Material, Mesh, Scene, Sphere
from raytracer_amd.scripts.numba_utils import classes as nb
from raytracer_amd.scripts.scenes import chunker
from raytracer_amd.scripts.scenes.utils import reset_global_datas
from raytracer_amd.vecmath import Vector3, Vector4

F = np.float32


def box_triangles(lo, hi):
    """12 outward-facing triangles of an axis-aligned box, flat normals."""
    lo, hi = np.asarray(lo, F), np.asarray(hi, F)
    c = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])], dtype=F)
    quads = [(0, 1, 3, 2, (-1, 0, 0)), (4, 6, 7, 5, (1, 0, 0)), (0, 4, 5, 1, (0, -1, 0)),
             (2, 3, 7, 6, (0, 1, 0)), (0, 2, 6, 4, (0, 0, -1)), (1, 5, 7, 3, (0, 0, 1))]
    pos, nrm = [], []
    for a, b, cc, d, n in quads:
        for tri in ((a, b, cc), (a, cc, d)):
            p = c[list(tri)]
            if np.dot(np.cross(p[1] - p[0], p[2] - p[0]), n) < 0:
                p = p[[0, 2, 1]]
            pos.append(p)
            nrm.append(np.tile(np.asarray(n, F), (3, 1)))
    return np.asarray(pos, F), np.asarray(nrm, F)


def uv_sphere(center, radius, nu=10, nv=6):
    """Closed triangle mesh of a sphere with smooth (radial) normals."""
    pos, nrm = [], []
    def pt(i, j):
        th, ph = 2 * np.pi * i / nu, np.pi * j / nv
        return np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])
    for i in range(nu):
        for j in range(nv):
            q = [pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)]
            for tri in ((0, 2, 1), (0, 3, 2)):
                n = np.array([q[k] for k in tri])
                if np.linalg.norm(np.cross(n[1] - n[0], n[2] - n[0])) < 1e-9:
                    continue
                pos.append(n * radius + np.asarray(center))
                nrm.append(n)
    return np.asarray(pos, F), np.asarray(nrm, F)


def add_mesh(scene, pos, nrm, material, csys=None):
    m = Mesh(material=material)
    m.triangles = nb.TriangleSet(pos, nrm, m.id, len(nb.get_all_triangles_arr()))
    nb.add_to_all_triangles(m.triangles)
    if csys is not None:
        m.csys = csys
    scene.meshes.append(m)
    chunker.chunk_mesh_bvh(m)
    return m


def stress_scene(n_cubes=40, spheres=True, dof=0.02):
    reset_global_datas()
    rng = np.random.default_rng(7)
    sc = Scene()
    sc.atmosphere_material = Material(Vector4((0.9, 0.95, 1.0, 1.0), dtype="f4"), transmission=0.98, ior=1.0)
    white = Material(Vector4((0.8, 0.8, 0.8, 1.0), dtype="f4"), smoothness=0.3, specularStrength=0.3)
    light = Material(Vector4((0, 0, 0, 1.0), dtype="f4"), emissionColor=Vector3((1.0, 0.9, 0.8), dtype="f4"), emissionStrength=6.0)
    glass = Material(Vector4((0.9, 1.0, 0.9, 1.0), dtype="f4"), smoothness=1.0, specularStrength=0.05, transmission=0.9, ior=1.5,
                     specularColor=Vector3((1.0, 1.0, 0.9), dtype="f4"))
    veil = Material(Vector4((1.0, 0.3, 0.3, 0.5), dtype="f4"), transparent_from_behind=True)     # alpha pass-through
    mirror = Material(Vector4((1, 1, 1, 1.0), dtype="f4"), smoothness=0.95, specularStrength=1.0)
    # floor + ceiling light + back wall mirror
    add_mesh(sc, *box_triangles((-6, -1.2, -6), (6, -1, 6)), white)
    add_mesh(sc, *box_triangles((-2, 4, -2), (2, 4.2, 2)), light)
    add_mesh(sc, *box_triangles((-6, -1, 5.8), (6, 5, 6)), mirror)
    # glass ball (mesh) and a veil in front of the camera
    add_mesh(sc, *uv_sphere((0.0, 0.3, 0.0), 1.2, 14, 9), glass)
    add_mesh(sc, *box_triangles((-1.5, -1, -3.0), (1.5, 2, -2.95)), veil)
    # many small posed cubes: more than 32 meshes -> the top-level tree instead of the LDS root mask
    for i in range(n_cubes):
        from raytracer_amd.scripts.numba_utils.classes import Csys

        c = Csys()
        c.set_pos(rng.uniform((-5, -0.8, -4), (5, 3, 5)))
        c.rxg(float(rng.uniform(0, 360))).ryg(float(rng.uniform(0, 360)))
        mat = Material(Vector4((*rng.uniform(0.2, 1.0, 3), 1.0), dtype="f4"), smoothness=float(rng.uniform(0, 1)),
                       specularStrength=float(rng.uniform(0, 0.6)), transmission=float(rng.choice([0.0, 0.0, 0.7])), ior=1.3)
        h = rng.uniform(0.1, 0.35, 3)
        add_mesh(sc, *box_triangles(-h, h), mat, c)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    if spheres:
        sc.spheres.append(Sphere(Vector3((2.5, 0.0, 1.0)), 1.0, glass))
        sc.spheres.append(Sphere(Vector3((-2.5, -0.2, 0.5)), 0.8, Material(Vector4((0.2, 0.4, 1.0, 1.0), dtype="f4"))))
        sc.spheres.append(Sphere(Vector3((0.0, 3.0, 3.0)), 0.5, light))
    cam = sc.cam
    cam.csys.set_pos([0.0, 1.0, -9.0])
    cam.fov = 40
    cam.near_plane = 9.0
    cam.depth_of_field_strength = dof
    cam.antialias_strength = 0.002
    return sc
