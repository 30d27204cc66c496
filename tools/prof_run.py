#!/usr/bin/env python3
"""One k_trace launch of bench.py's workload (optionally fewer passes) for rocprofv3 runs.
usage: python tools/prof_run.py [passes] [width height] [bounces]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from raytracer_amd import ffi
from raytracer_amd.scripts.scenes import final_scene_numba as S

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
bounces = int(sys.argv[4]) if len(sys.argv) > 4 else 4
import os
lib = ffi.RtLib(os.environ.get('RT_LIB') or None)
which = os.environ.get('RT_SCENE', 'suzanne')          # suzanne | suzanne5 (991k triangles) | dragon (192 meshes)
if which == 'dragon':
    from raytracer_amd.scripts.scenes import dragon_scene as D
    sc = D.build()
elif which == 'dragon_merged':
    # experiment: the dragon scene's triangles as ONE mesh (one SAH tree over everything instead of a top-level tree over
    # 192 per-mesh trees).  Materials are then all the dragon's, so the image differs: step counts and time only.
    import numpy as np
    from raytracer_amd.scripts.classes import Mesh, Scene
    from raytracer_amd.scripts.scenes import chunker, dragon_scene as D
    from raytracer_amd.scripts.numba_utils import classes as _nb
    from raytracer_amd.scripts.scenes.utils import read_obj, reset_global_datas
    src = D.build()
    files, mat, atmo, pose, cam = src.mesh_files, src.meshes[0].material, src.atmosphere_material, src.meshes[0].csys, src.cam
    reset_global_datas()
    sc = Scene()
    sc.atmosphere_material = atmo
    mesh = Mesh(material=mat)
    parts = [read_obj(f"{D.DRAGON_DIR}/{f}") for f in files]
    mesh.triangles = _nb.TriangleSet(np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]), mesh.id, 0)
    _nb.add_to_all_triangles(mesh.triangles)
    mesh.csys = pose
    sc.meshes.append(mesh)
    chunker.chunk_mesh_bvh(mesh)
    chunker.BVHGraph.register_all_and_update_node_tri_ids()
    sc.cam = cam
elif which.startswith('suzanne') and which != 'suzanne':
    sc = S.build(subdivide=int(which[7:]), lib=lib)
else:
    sc = S.build()
sc.animate_frame(0)
ctx = lib.create(0)
ctx.set_accel(os.environ.get('RT_SAH', '0') == '1')
ctx.load_scene(sc)
ctx.load_camera(sc.cam)
ctx.resize(w, h)
import os
ctx.set_cull(os.environ.get('RT_CULL', '0') == '1')
if os.environ.get('RT_STATS') == '1':
    ctx.enable_stats(True)
ctx.render(1, passes, 1, 1, bounces, 0)
ctx.sync()
st = ctx.stats()
if os.environ.get("RT_STATS") == "1":
    r = st["rays"]; sd = st["sched"]
    print(f"rays={r} B/ray={st['box_tests']/r:.2f} T/ray={st['tri_tests']/r:.2f} H/ray={st['shaded_hits']/r:.3f}")
    print(f"node steps/ray={sd[1]/r:.2f} avg lanes={sd[1]/max(sd[0],1):.1f}; tri steps/ray={sd[3]/r:.2f} avg lanes={sd[3]/max(sd[2],1):.1f}; path steps/ray={sd[5]/r:.2f} avg lanes={sd[5]/max(sd[4],1):.1f}")
print(f"scene={which} tris={sc.count_triangles()} passes={passes} {w}x{h} bounces={bounces} kernel_ms={st['kernel_ms']:.3f}")
