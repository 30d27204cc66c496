#!/usr/bin/env python3
"""bench.py - Mrays/s of the rendering hot path.

Default workload = BASELINE.json configs[1]: Suzanne room, 1920x1080, 64 spp as 64 passes x
1 ray/pixel, 4 bounces, one MI355X.  `--workload subdiv4k` = configs[3] (Suzanne subdivided to
991 316 triangles, 3840x2160, 256 passes), `--workload dragon` = configs[4] (the 192-mesh dragon
scene, 1080p, 64 passes, 8 bounces); the driver's line stays on the default.

A "step" = one full render of that frame (one persistent k_trace launch over all passes, which
also folds them into the accumulator; scene and accumulator resident in HBM).  ray = one
closest-hit query (calculateRayCollision, shaders/raytracer.fs.glsl:518-641 of the reference),
counted exactly in-kernel by a separate counting launch of the same deterministic work.

N > 1: one rank per GPU.  Either launched by torch.distributed.run (the driver's way: RANK / WORLD_SIZE in the
environment) or plainly as `python bench.py --gpus N`, in which case this process starts the N ranks itself through
torch.distributed.run (before it imports torch or touches a GPU) and forwards rank 0's line and the exit code.  `--scaling`:
  weak   (default) every rank renders all PASSES of the frame with its own pass indices (rank r
         takes 1+r, 1+r+N, ...), i.e. PASSES*N spp in total, into a SUM accumulator;
  strong the frame's PASSES are split over the ranks (raytracer_amd.distributed.shard_passes):
         fixed total work, what "x-times faster at N GPUs" means (north_star: >= 7x at 8 GPUs on
         the 4K spp-sharded config = `--workload subdiv4k --scaling strong`);
both with one RCCL all-reduce of the W*H*3 float32 sum per step (the exchange the path really has,
SURVEY.md 8(e)).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

# the BASELINE.json configurations that fit one GPU: scene recipe, image, sampling, default steps, CPU sample
WORKLOADS = {
    "suzanne1080": dict(config="configs[1]", scene="suzanne", what="Suzanne room (1052 triangles, 9 meshes)",
                        width=1920, height=1080, passes=64, rpp=1, bounces=4, steps=20, cpu_passes=20, cpu_size=(1920, 1080)),
    "subdiv4k": dict(config="configs[3]", scene="suzanne5", what="Suzanne subdivided 5x (991 316 triangles) in the room",
                     width=3840, height=2160, passes=256, rpp=1, bounces=4, steps=2, cpu_passes=4, cpu_size=(1920, 1080)),
    "dragon": dict(config="configs[4]", scene="dragon", what="dragon scene (192 meshes, 12 658 triangles, glossy/emissive)",
                   width=1920, height=1080, passes=64, rpp=1, bounces=8, steps=20, cpu_passes=16, cpu_size=(1920, 1080)),
}
# RT_BENCH_REHEARSAL=1: a tiny frame, ONLY to walk the N > 1 code path with several ranks sharing one
# GPU (two full-size persistent kernels time-slice pathologically there); its line says so and is
# not a measurement
REHEARSAL = os.environ.get("RT_BENCH_REHEARSAL", "0") == "1"
# algorithmic work per unit (DESIGN.md "roofline accounting")
FLOP_BOX, FLOP_TRI, FLOP_HIT = 27, 47, 160
BYTE_BOX, BYTE_TRI, BYTE_HIT = 32, 36, 48
PEAK_FP32_TFLOPS = 157.3   # MI355X FP32 vector peak (= dense f32 MFMA peak), MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def build_scene(which: str, lib=None):
    from raytracer_amd.scripts.scenes import final_scene_numba as S

    if which == "dragon":
        from raytracer_amd.scripts.scenes import dragon_scene as D

        sc = D.build()
    elif which.startswith("suzanne") and which != "suzanne":
        sc = S.build(subdivide=int(which[7:]), lib=lib)          # BVH of the 991 316 triangles by rt_bvh_build
    else:
        sc = S.build()
    sc.animate_frame(0)
    return sc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: per workload, a few seconds of GPU time)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="suzanne1080")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-passes", type=int, default=None, help="passes of the CPU baseline sample (default per workload: about 10-20 s on 16 threads)")
    ap.add_argument("--strict", action="store_true", help="visit exactly the reference's box set (no distance culling)")
    ap.add_argument("--reference-structure", action="store_true", help="traverse the reference's own BVH instead of the SAH one")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the untimed legs on the reference's own BVH (profiling runs)")
    args = ap.parse_args()
    if os.environ.get("RT_BENCH_WATCHDOG"):            # seconds: dump every thread's stack and exit if the run is still going then
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["RT_BENCH_WATCHDOG"]), exit=True)
    wl = dict(WORKLOADS[args.workload])
    if REHEARSAL:
        wl.update(width=160, height=90, passes=4)
    WIDTH, HEIGHT, PASSES, RPP, BOUNCES = wl["width"], wl["height"], wl["passes"], wl["rpp"], wl["bounces"]
    if args.steps is None:
        args.steps = wl["steps"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))       # this process never imports torch nor loads librt_hip.so
    if world != args.gpus:
        args.gpus = world

    if os.environ.get("RT_BENCH_LAUNCH_CHECK") == "1":
        # CPU-only check of the launch plumbing (tests/test_distributed_cpu.py): rendezvous over gloo, one collective,
        # rank 0's line - and out, before anything needs a GPU.  Not a measurement.
        import torch
        import torch.distributed as dist

        dist.init_process_group("gloo")
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": dist.get_world_size(), "rank_sum": int(t.item()), "scaling": args.scaling,
                              "steps": args.steps, "warmup": args.warmup}), flush=True)
        dist.destroy_process_group()
        return

    # The CPU baseline leg runs FIRST (rank 0, N = 1 only), before anything touches the GPU: the GPU
    # legs then form one contiguous stretch at the end of the process, which is what a sampler of GPU
    # activity (rocm-smi every few seconds) needs to see them.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(wl, args.cpu_passes if args.cpu_passes is not None else wl["cpu_passes"])

    import torch
    import torch.distributed as dist

    from raytracer_amd import ffi
    from raytracer_amd.distributed import shard_passes

    # RT_BENCH_BACKEND=gloo + fewer GPUs than ranks is a rehearsal of the N > 1 code path on a
    # one-GPU box (ranks share the card); the real run is nccl (= RCCL), one GPU per rank
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"{world} ranks but {n_dev} GPUs")
    local_rank %= max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RT_BENCH_FORCE_DIST=1: take the N > 1 code path (process group, SUM accumulation, all-reduce, barriers) with
    # a single rank too: the only way to run the RCCL leg on a one-GPU box (tests/test_gpu_rccl_single_rank.py)
    dist_on = world > 1 or os.environ.get("RT_BENCH_FORCE_DIST", "0") == "1"
    if dist_on:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def barrier():
        if dist_on:
            dist.barrier()

    # ---- scene + context (inputs resident in HBM before the timed region) ----
    lib = ffi.RtLib()                      # hand-written HIP kernels; raises if missing
    scene = build_scene(wl["scene"], lib)
    ctx = lib.create(local_rank)
    # the opt-in SAH structure (same triangles, reference ranks for ties) + distance culling: the
    # fastest mode whose image is bit-identical to the reference structure's exact traversal on this
    # frame (asserted by tests/test_gpu_sah.py::test_sah_bench_frame_is_pixel_identical)
    ctx.set_accel(not args.reference_structure)
    ctx.load_scene(scene)
    ctx.load_camera(scene.cam)
    ctx.resize(WIDTH, HEIGHT)
    ctx.set_cull(not args.strict)          # conservative distance culling: same image bit for bit (tests), fewer box tests
    accum = torch.zeros((HEIGHT, WIDTH, 3), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)     # kernels, events and the all-reduce share one stream
    ctx.bind_accum(accum.data_ptr())

    sum_mode = dist_on
    if not sum_mode:
        first_pass, n_passes, stride, total_passes = 1, PASSES, 1, PASSES
    elif args.scaling == "weak":
        first_pass, n_passes, stride, total_passes = 1 + rank, PASSES, world, PASSES * world
    else:
        first_pass, n_passes, stride = shard_passes(PASSES, rank, world)
        total_passes = PASSES

    def render():
        ctx.reset_accum()
        if n_passes:
            ctx.render(first_pass, n_passes, stride, RPP, BOUNCES, ffi.RT_ACCUM_SUM if sum_mode else ffi.RT_ACCUM_MEAN)

    def step():
        if REHEARSAL and world > 1:
            # ranks that SHARE one GPU take turns: two persistent kernels preempting each other can take minutes
            # (context save/restore of a full chip per time slice); the code path is the same
            for r in range(world):
                if r == rank:
                    render()
                    torch.cuda.synchronize(dev)
                dist.barrier()
        else:
            render()
        if sum_mode:
            dist.all_reduce(accum)         # RCCL over xGMI; same stream as the render
            ctx.set_accum_passes(total_passes)

    # ---- counting launch: exact rays / box tests / triangle tests / hits of one step ----
    ctx.enable_stats(True)
    ctx.reset_stats()
    step()
    torch.cuda.synchronize(dev)
    counts = ctx.stats()
    ctx.enable_stats(False)
    cvec = torch.tensor([counts["rays"], counts["box_tests"], counts["tri_tests"], counts["shaded_hits"]],
                        dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(cvec)
    rays_step, box_step, tri_step, hit_step = (float(x) for x in cvec.tolist())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    ctx.reset_stats()

    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    # ---- dominant kernel: average launch duration by HIP events on the launch stream ----
    st = ctx.stats()
    info = ctx.info()
    launches = max(int(st["launches"]), 1)
    kern_ms = st["kernel_ms"] / launches
    per_step = max(1, launches // max(args.steps, 1))            # k_trace launches one step needs (1024 passes each at most)
    # per-launch algorithmic work of THIS rank (rank 0 reports its own kernel)
    flop_launch = (FLOP_BOX * counts["box_tests"] + FLOP_TRI * counts["tri_tests"] + FLOP_HIT * counts["shaded_hits"]) / per_step
    # bytes the traversal touches, as if every fetch went to memory ...
    byte_launch = (BYTE_BOX * counts["box_tests"] + BYTE_TRI * counts["tri_tests"] + BYTE_HIT * counts["shaded_hits"]) / per_step
    # ... and the streams that leave the L2.  ALGORITHMIC (passes looped in-kernel): the accumulator, 12 B read +
    # 12 B written per pixel per launch.  DESIGN OVERHEAD, not algorithmic: the radiance of every sample is parked in its wave's
    # staging slot and read back once by the fold (12 + 12 B per (pixel, pass); the 75 MB of slots do not stay in 8 x 4 MB of L2) -
    # the price of tracing the samples of a pixel on different lanes in any order while folding them in pass order.
    accum_launch = 24.0 * WIDTH * HEIGHT
    stage_launch = 24.0 * WIDTH * HEIGHT * n_passes / per_step
    hbm_launch = accum_launch + stage_launch
    achieved_tflops = flop_launch / (kern_ms * 1e-3) / 1e12 if kern_ms > 0 else 0.0
    achieved_gbs = accum_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
    cache_gbs = byte_launch / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0

    # HBM bytes per launch from the PMC passes of THIS command line (tools/prof_bench.sh; counters cannot be
    # read from inside the process): profiles/pmc_latest.json holds them per workload, with their provenance
    traffic, traffic_source = None, None
    pmc = ROOT / "profiles" / "pmc_latest.json"
    if pmc.exists() and not args.strict and not args.reference_structure and not REHEARSAL:
        try:
            rec = json.loads(pmc.read_text()).get(args.workload)
            if rec:
                traffic = rec.get("hbm_bytes_per_launch")
                traffic_source = {k: rec.get(k) for k in ("file", "commit", "command", "correction")}
        except Exception:
            traffic = None

    value = rays_step * args.steps / dt / 1e6

    # for the record: the same frame on the reference's own BVH, culled and exact (rt_set_cull(0));
    # not the headline (the headline mode renders the identical image)
    # These untimed legs run AFTER the timed region, each for about a second, so that the GPU stays busy for a few seconds at
    # the end of the process (a sampler of GPU activity with a 5 s period otherwise misses a 1 s run).
    other, band_only, audit = None, None, None
    if not dist_on and not args.strict and not args.reference_structure and not args.no_other_modes and not REHEARSAL:
        def timed2(min_s=1.0):
            step()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            n = 0
            while n < 2 or time.perf_counter() - t1 < min_s:
                step()
                torch.cuda.synchronize(dev)
                n += 1
            d = (time.perf_counter() - t1) / n
            return {"ms_per_step": round(d * 1e3, 3), "value": round(rays_step / d / 1e6, 2), "unit": "Mrays/s", "steps": n}

        # the fast mode with the weaker assumption: SAH structure + the distance band alone (rt_set_cull(ctx, RT_CULL_BAND))
        ctx.set_cull(ffi.RT_CULL_BAND)
        band_only = timed2()
        band_only["mode"] = "SAH structure + culling rule (a) only (distance band 2^-10; DESIGN.md 5), same rays, same image"
        # the culling audit of THIS frame, counted on the GPU by one counting launch of the exact mode (its winners are the
        # reference's): queries whose winner's own bounding box either rule could skip (rt_stats_t.cull_audit)
        ctx.set_cull(ffi.RT_CULL_OFF)
        ctx.enable_stats(True)
        ctx.reset_stats()
        step()
        torch.cuda.synchronize(dev)
        ca = ctx.stats()["cull_audit"]
        ctx.enable_stats(False)
        audit = {"queries_won_by_a_triangle": int(ca[0]), "flagged_rule_a_band": int(ca[1]), "flagged_rule_b_behind": int(ca[2]),
                 "note": "one exact-mode counting launch of this step on the GPU: queries whose winner's own box lies beyond the band (a) or is "
                         "left before t = 5e-7 (b); 0 = no culling traversal of any structure over these triangles can lose a winner on this frame"}
        if args.workload == "suzanne1080":
            ctx.set_accel(False)
            ctx.load_scene(scene)
            ctx.set_cull(True)
            other = {"structure": "the reference's BVH as uploaded (rt_set_accel(RT_ACCEL_REFERENCE)), same rays, same image",
                     "culled": timed2()}
            ctx.set_cull(False)
            other["exact_box_set"] = timed2()

    if rank == 0:
        deep = info["stack_entries_lds"] < info["stack_entries"]
        kernel = "k_trace<STATS=false,CULL=%s,FIRST_HIT=false,SAH=%s,DEEP=%s,ROCT=%s>" % (
            "0" if args.strict else "1", "false" if args.reference_structure else "true", "true" if deep else "false",
            "true" if info.get("root_table_variants") == 8 else "false")
        hbm_frac = (traffic / (kern_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if (traffic and kern_ms > 0) else None
        out = {
            "metric": "Mrays/sec, Suzanne 1080p (rays = closest-hit queries, counted in-kernel)" if args.workload == "suzanne1080"
                      else f"Mrays/sec, {args.workload} (rays = closest-hit queries, counted in-kernel)",
            "value": round(value, 2),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": f"synthetic: {wl['what']} from the reference's OBJ assets, camera = keyframe 0, RNG seeds = pass 1..{total_passes}",
            "config": {
                "workload": f"{wl['config']}: {wl['what']}, {WIDTH}x{HEIGHT}, {PASSES} spp ({PASSES} passes x {RPP} ray/px), "
                            f"{BOUNCES} bounces" + (", per GPU" if args.scaling == "weak" else ", whole job"),
                "structure": "reference BVH" if args.reference_structure else "SAH over the same triangles, reference tie ranks (exact by construction: DESIGN.md 5)",
                "traversal": "strict (reference box set)" if args.strict else
                             ("distance-culled, rules (a)+(b) (bit-identical to the oracle on this frame: tests/test_gpu_fullsize.py; "
                              "cull_audit below is this frame's evidence)"),
                "parallelism": "1 GPU, running mean" if not dist_on else
                               (f"spp-sharded x{world}, {args.scaling} scaling (rank r renders passes r+1, r+1+{world}, ...: {n_passes} on rank 0) "
                                f"+ RCCL all-reduce of the {WIDTH*HEIGHT*12/1e6:.1f} MB sum"),
                "rays_per_step": int(rays_step),
                "box_tests_per_ray": round(box_step / rays_step, 2),
                "tri_tests_per_ray": round(tri_step / rays_step, 2),
                "hits_per_ray": round(hit_step / rays_step, 3),
                "k_trace_launches_per_step": per_step,
                "node_stack_entries": [info["stack_entries_lds"], info["stack_entries"]],
            },
            "roofline": {
                "bound": "mfma",
                "engine": "f32 VALU issue (no MFMA instruction is issued: no GEMM shape on this path; the contract's 'mfma' stands for the "
                          "arithmetic roofline, priced at the f32 vector-FMA peak = the dense f32 MFMA peak)",
                "kernel": kernel,
                "achieved": round(achieved_tflops, 3),
                "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved_tflops / PEAK_FP32_TFLOPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "avg_launch_ms": round(kern_ms, 3),
                "flop_per_launch": int(flop_launch),
                "hbm": {"achieved": round(achieved_gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(achieved_gbs / PEAK_HBM_GBS, 4),
                        "measured_frac": None if hbm_frac is None else round(hbm_frac, 4),
                        "algorithmic_bytes_per_launch": int(accum_launch),
                        "design_overhead_bytes_per_launch": int(stage_launch),
                        "traffic_over_algorithmic": None if not traffic else round(traffic / accum_launch, 2),
                        "traffic_over_algorithmic_plus_overhead": None if not traffic else round(traffic / hbm_launch, 3),
                        "note": "algorithmic (SURVEY.md 8(d)) = the accumulator, 12 B read + 12 B written per pixel per launch; design overhead = "
                                "the waves' staging slots (12 B written + 12 B read per sample, 75 MB in all: they pass through the L2 to the "
                                "Infinity Cache) - the cost of folding passes in order inside the trace kernel instead of in a second kernel over "
                                "a per-sample buffer; achieved/frac price the algorithmic bytes, measured_frac = roofline.traffic / launch time / "
                                "peak; neither is the bound of this kernel"},
                "cache": {"node_triangle_bytes_per_launch": int(byte_launch), "rate_GBps": round(cache_gbs, 1),
                          "note": "32 B/box test + 36 B/triangle test + 48 B/hit (SURVEY.md 8d), served by L1/L2 (Infinity Cache for subdiv4k)"},
            },
            "cpu_baseline": cpu,
            "reference_structure": other,
            "sah_rule_a_only": band_only,
            "cull_audit": audit,
            "cull_audit_flagged": None if audit is None else audit["flagged_rule_a_band"] + audit["flagged_rule_b_behind"],
        }
        if REHEARSAL:
            out["rehearsal"] = "RT_BENCH_REHEARSAL=1: 160x90, 4 passes - not a measurement"
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist_on:
        dist.destroy_process_group()


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes through
    torch.distributed.run (one per GPU, rendezvous on 127.0.0.1) with this command line, let rank 0's JSON line flow
    through on stdout, and return the launcher's exit code.  The parent touches no GPU and nothing is re-exec'd."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")             # (torchrun sets it anyway, with a warning)
    return subprocess.run(cmd, env=env).returncode


def cpu_baseline(wl: dict, passes: int) -> dict:
    """The CPU oracle (a port: the reference has no CPU renderer) on a bounded sample of the same
    workload: same scene/camera/seeds, the first `passes` passes on all the threads OpenMP gives us,
    and pass 1 alone on one thread at a quarter of the pixels.  Runs in a child
    process so that nothing of it (OpenMP pool, 1 GB of scene copies) stays in the GPU process."""
    code = (
        "import json,sys,time,os\n"
        f"sys.path.insert(0, {str(ROOT)!r})\n"
        "import bench\n"
        "print(json.dumps(bench._cpu_baseline_child(json.loads(sys.argv[1]), int(sys.argv[2]))))\n"
    )
    r = subprocess.run([sys.executable, "-c", code, json.dumps(wl), str(passes)], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("cpu_baseline child failed:\n" + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def _cpu_baseline_child(wl: dict, passes: int) -> dict:
    from raytracer_amd import ffi

    lib = ffi.RtLib(ROOT / "oracle" / "librt_oracle.so")
    lib.dll.rto_set_threads.restype = int
    threads = int(lib.dll.rto_set_threads(int(os.environ.get("OMP_NUM_THREADS", "0") or 0)))
    scene = build_scene(wl["scene"], lib)
    ctx = lib.create(0)
    ctx.load_scene(scene)
    ctx.load_camera(scene.cam)
    W, H = wl["cpu_size"]
    ctx.resize(W, H)

    def timed(n_passes):
        ctx.reset_stats()
        ctx.reset_accum()
        t0 = time.perf_counter()
        ctx.render(1, n_passes, 1, wl["rpp"], wl["bounces"], ffi.RT_ACCUM_MEAN)
        dt = time.perf_counter() - t0
        return ctx.stats()["rays"], dt

    rays, dt = timed(passes)
    lib.dll.rto_set_threads(1)
    ctx.resize(W // 2, H // 2)                     # a quarter of the pixels keeps the scalar run to seconds
    rays1, dt1 = timed(1)
    lib.dll.rto_set_threads(threads)
    ctx.close()
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"oracle/rt_oracle.c (OpenMP), same scene/camera/seeds, {W}x{H}, passes 1..{passes} of {wl['passes']}, "
                  f"{wl['bounces']} bounces: {rays} rays in {dt:.2f} s",
        "single_thread": {"value": round(rays1 / dt1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
                          "sample": f"pass 1 at {W // 2}x{H // 2}: {rays1} rays in {dt1:.2f} s"},
    }


if __name__ == "__main__":
    main()
