#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and frames/s of the progressive path at 1080p, 1 spp/frame.

    python bench.py --gpus N --steps K --warmup W

A "step" is one progressive frame (ProgressiveRaytracingPipeline::render,
src/ProgressiveRaytracingPipeline.cpp:215-247) of BASELINE.json configs[1]: the
Sponza-class synthetic atrium (~262k triangles) at 1920x1080, one sample per pixel
per frame, reference default material / lights / options, everything resident in HBM.
For N > 1 (one process per GPU under torch.distributed.run) rank r renders frames
{f : f mod N == r} into an fp32 SUM buffer and ONE RCCL all-reduce of that buffer
closes the timed region (weak scaling: K frames per GPU).

Prints ONE JSON line on rank 0.  `value` counts every ray handed to traversal
(primary + secondary radiance + shadow); the reference's own figure,
width*height*fps/1e6 primary rays only (src/utils/DXSample.cpp:114), is reported
beside it as `primary_mrays_per_s`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
RAY_BYTES, NODE_BYTES, TRI_BYTES = 48, 32, 36   # 32 B ray in + 16 B hit out; node; triangle
# stage -> (rt_stats time fields, kernel, rt_pipeline_count_work stages whose rays the launch traces)
TRACE_STAGES = {"primary": (("ms_primary",), "k_primary", ("primary",)),
                "secondary": (("ms_trace_secondary",), "k_trace_secondary", ("secondary",)),
                "shadow": (("ms_trace_shadow0", "ms_trace_shadow1"), "k_trace_shadow", ("shadow0", "shadow1"))}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget (rank 0, N=1 only); 0 = skip")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def relaunch_distributed(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD
    (nothing here has touched the GPU yet) and exit with its code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(29500 + os.getpid() % 1000), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def usable_cores():
    """Host threads this process may really run on: the affinity mask, cut down by a cgroup CPU quota if one is set
    (containers on many-core hosts report every core in os.cpu_count() and then throttle)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                txt = f.read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, int(quota / int(g.read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(verts, tris, mat, env, pfc, W, H, budget_s):
    """Scalar CPU restatement (oracle/) timed on this host: bands of the same 1080p frame until the budget is spent."""
    import numpy as np
    from oracle import pyoracle as O
    cores = usable_cores()
    sc = O.Scene()
    sc.add_instance(sc.add_model(verts, tris))
    t0 = time.perf_counter()
    sc.build()
    build_s = time.perf_counter() - t0
    # the oracle splits a tile by rows, statically: a band needs a few rows per thread or most threads idle
    band = 8
    while band < 4 * cores and band * 2 <= H // 2:
        band *= 2
    acc = np.zeros((H, W, 4), np.float32)
    rays = 0
    bands = 0
    t0 = time.perf_counter()
    done = False
    for sweep in range(64):                       # many-core hosts finish a frame before the budget: keep sweeping
        for b in np.random.default_rng(sweep).permutation(H // band):
            _, st = sc.render(mat, pfc, W, H, accum=acc, env_faces=env, tile=(0, int(b) * band, W, int(b + 1) * band), nthreads=cores)
            rays += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
            bands += 1
            if time.perf_counter() - t0 >= budget_s:
                done = True
                break
        if done:
            break
    dt = time.perf_counter() - t0
    # the same scalar code on ONE thread (SURVEY 8(d) asks for both): a few bands, ~2 s
    t1 = time.perf_counter()
    rays1 = bands1 = 0
    for b in np.random.default_rng(99).permutation(H // 8):
        _, st = sc.render(mat, pfc, W, H, accum=acc, env_faces=env, tile=(0, int(b) * 8, W, int(b + 1) * 8), nthreads=1)
        rays1 += st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
        bands1 += 1
        if time.perf_counter() - t1 >= min(2.0, budget_s / 4):
            break
    dt1 = time.perf_counter() - t1
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d random %d-row bands of the same 1920x1080 frame (%d rays, all ray types) in %.1f s; BVH build %.2f s"
                      % (bands, band, rays, dt, build_s),
            "single_thread": {"value": rays1 / dt1 / 1e6, "unit": "Mrays/s", "cores": 1,
                              "sample": "%d bands (%d rays) in %.1f s" % (bands1, rays1, dt1)}}


def measured_traffic(kernel):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE and
    WRITE_SIZE need separate passes, so they cannot be taken inside this run): profiles/<round>/traffic.json,
    written by tools/traffic_from_pmc.py with the gfx950 FETCH_SIZE correction applied.  None if absent."""
    here = os.path.dirname(os.path.abspath(__file__))
    for rnd in sorted(os.listdir(os.path.join(here, "profiles")), reverse=True) if os.path.isdir(os.path.join(here, "profiles")) else []:
        path = os.path.join(here, "profiles", rnd, "traffic.json")
        if os.path.isfile(path):
            with open(path) as f:
                k = json.load(f).get("kernels", {}).get(kernel)
            if k:
                return k["bytes_per_launch"]
    return None


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks (tests/test_gpu_scale.py runs two ranks on the one GPU of a test box): every rank on one
    # device, and gloo instead of RCCL, which refuses two ranks per device
    if "DXR_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DXR_BENCH_DEVICE"])
    backend = os.environ.get("DXR_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and world == 1:
        sys.exit(relaunch_distributed(args))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    import numpy as np
    import torch
    import torch.distributed as dist
    from dxrexperiments_amd import capi, distributed as D, rtypes as T, scenes

    assert torch.cuda.is_available(), "bench.py needs a GPU: the product has no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # "nccl" is RCCL on ROCm: xGMI between the GPUs of the node
        else:
            dist.init_process_group(backend)

    W, H, K, Wu = args.width, args.height, args.steps, args.warmup
    verts, tris = scenes.sponza_class(seed=42)
    env = scenes.sky_cubemap(64)
    mat = T.default_material()

    ctx = capi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    model = capi.Model(ctx, verts, tris)
    scene = capi.Scene(ctx)
    scene.add_model(model)
    pipe = capi.Pipeline(ctx)
    pipe.set_scene(scene)
    pipe.add_material(mat)
    pipe.set_environment_cube(env)
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    pipe.bind_output(acc.data_ptr(), W, H)
    pipe.build_acceleration_structures()
    build_ms = scene.build_ms()              # first build of the process: includes loading the build kernels
    scene2 = capi.Scene(ctx)                 # the same build again, warm: the steady-state figure
    scene2.add_model(capi.Model(ctx, verts, tris))
    scene2.build()
    rebuild_ms = scene2.build_ms()
    del scene2

    # every rank generates the SAME global frame sequence and renders its share of it
    host = capi.ProgressiveHost(1234)
    total_frames = (Wu + K) * world
    host.options["maxIterations"] = max(1024, total_frames + 1)
    cam = scenes.sponza_camera()
    cam11 = capi.camera_array(cam["eye"], cam["at"], cam["up"], cam["fov"], W / H)
    pfcs = [host.update(cam11, 0.0, f + 1, W, H) for f in range(total_frames)]
    mine = D.shard_frames(rank, world, total_frames)
    pipe.set_accumulation_mode(T.ACCUM_SUM if world > 1 else T.ACCUM_RUNNING_MEAN)

    def step(i):
        pipe.update(pfcs[mine[i]])
        pipe.render()

    for i in range(Wu):
        step(i)
    if world > 1:                       # warm the collective too
        dist.all_reduce(torch.zeros_like(acc))
    torch.cuda.synchronize()
    if not args.no_roofline:
        pipe.enable_timing(K)
    pipe.reset_totals()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(Wu, Wu + K):
        step(i)
    if world > 1:
        mean, n_frames = D.reduce_accumulation(acc, Wu + K)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    tot = pipe.totals()
    rays_local = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"]
    red = torch.tensor([elapsed, float(rays_local), float(tot["rays_primary"])], dtype=torch.float64, device=dev)
    if world > 1:
        mx = red.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
    rays_all, primary_all = float(red[1].item()), float(red[2].item())

    out = None
    if rank == 0:
        out = {
            "metric": "Mrays/s (all traced rays: primary + secondary + shadow), 1080p 1 spp/frame progressive",
            "value": rays_all / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": K, "warmup": Wu, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: Sponza-class procedural atrium (seed 42, %d triangles), %dx%d, 1 spp/frame "
                                   "progressive accumulation, reference default material/lights/options" % (tris.shape[0], W, H),
                       "frames_per_gpu": K, "parallelism": "sample-sharded x%d, one RCCL all-reduce of the fp32 accumulation buffer" % world
                       if world > 1 else "single GPU", "accumulation": "sum+allreduce" if world > 1 else "running mean"},
            "primary_mrays_per_s": primary_all / elapsed / 1e6,
            "frames_per_s": K * world / elapsed,
            "rays_per_frame": rays_all / (K * world),
            "bvh_build_ms": build_ms, "bvh_rebuild_ms": rebuild_ms,
        }
        if not args.no_roofline:
            work = pipe.count_work()                       # canonical counters of the last frame's queues
            n_t = max(int(tot["frames"]), 1)
            stages = {}
            for name, (keys, kernel, parts) in TRACE_STAGES.items():
                w = {k: sum(work[p][k] for p in parts) for k in ("rays", "nodes", "tris")}
                b = RAY_BYTES * w["rays"] + NODE_BYTES * w["nodes"] + TRI_BYTES * w["tris"]
                ms = sum(tot[k] for k in keys) / n_t
                stages[name] = {"kernel": kernel, "avg_ms": ms, "rays": w["rays"], "algorithmic_bytes": b,
                                "nodes_per_ray": w["nodes"] / max(w["rays"], 1), "tris_per_ray": w["tris"] / max(w["rays"], 1),
                                "GBps": b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                                "Mrays_per_s": w["rays"] / (ms * 1e-3) / 1e6 if ms > 0 else 0.0}
            for key in ("ms_shade0", "ms_shade1", "ms_resolve", "ms_total"):
                stages[key] = tot[key] / n_t
            dom = max(TRACE_STAGES, key=lambda s: stages[s]["avg_ms"])
            out["roofline"] = {"bound": "hbm", "kernel": stages[dom]["kernel"], "stage": dom,
                               "achieved": stages[dom]["GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": stages[dom]["GBps"] / HBM_PEAK_GBS, "traffic": measured_traffic(stages[dom]["kernel"]),
                               "algorithmic_bytes_per_launch": stages[dom]["algorithmic_bytes"],
                               "avg_launch_ms": stages[dom]["avg_ms"], "launches_timed": n_t}
            tb = sum(stages[s]["algorithmic_bytes"] for s in TRACE_STAGES)
            tm = sum(stages[s]["avg_ms"] for s in TRACE_STAGES)
            out["roofline"]["all_traversal_frac"] = tb / (tm * 1e-3) / 1e9 / HBM_PEAK_GBS if tm > 0 else 0.0
            out["stages"] = stages
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(verts, tris, mat, env, pfcs[mine[Wu]], W, H, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
