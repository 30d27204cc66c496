"""Entry point with the reference's shape (run.py:1-10): pick a scene module, start the app.

    python run.py [--scene suzanne|dragon|dragon_animated] [--frames N] [--height H] [--passes P] [--bounces B] [--cull] [--sah]

Headless: renders N animation frames of the chosen scene module on GPU 0 and writes PNG + float32
dumps under renders/RayTracerDynamic/<timestamp>/ at the reference's cadence."""
import argparse

import raytracer_amd.scripts.application as application

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", choices=("suzanne", "dragon", "dragon_animated"), default="suzanne",
                    help="scene module: final_scene_numba, dragon_scene or dragon_scene_animated")
    ap.add_argument("--frames", type=int, default=1)
    ap.add_argument("--height", type=int, default=1080 // 4)
    ap.add_argument("--passes", type=int, default=None, help="override the scene's passes_per_frame")
    ap.add_argument("--bounces", type=int, default=None)
    ap.add_argument("--cull", action="store_true", help="distance culling (same image, fewer box tests)")
    ap.add_argument("--sah", action="store_true", help="traverse the library's SAH tree over the same triangles (same image)")
    args = ap.parse_args()

    from raytracer_amd.scripts.animate import FrameAnimation
    import importlib

    module = {"suzanne": "final_scene_numba", "dragon": "dragon_scene", "dragon_animated": "dragon_scene_animated"}[args.scene]
    scene_module = importlib.import_module("raytracer_amd.scripts.scenes." + module)

    application.set_scene(scene_module)
    application.HEADLESS = True
    if args.passes is not None or args.bounces is not None:
        def override(cam, i):
            if args.passes is not None:
                cam.passes_per_frame = args.passes
            if args.bounces is not None:
                cam.bounces_per_ray = args.bounces
        scene_module.scene.animations.append(FrameAnimation(scene_module.scene.cam, override))
    prog = None
    if args.cull or args.sah:
        app0 = application.Application.__new__(application.Application)
        prog = application.RayTracerDynamic(app=app0, file_vertex_shader="shaders/raytracer.vs.glsl",
                                            file_fragment_shader="shaders/raytracer.fs.glsl",
                                            width=args.height * 16 // 9, height=args.height, cull=args.cull, sah=args.sah)
        application.Application.__init__(app0, render_height=args.height, max_frames=args.frames, program=prog)
        app0.start()
        raise SystemExit(0)
    app = application.Application(render_height=args.height, max_frames=args.frames)
    app.start()
