#!/usr/bin/env python3
"""Headline benchmark: Mrays/s and frames/s of the progressive path at 1080p, 1 spp/frame.

    python bench.py --gpus N --steps K --warmup W

A "step" is one progressive frame (ProgressiveRaytracingPipeline::render,
src/ProgressiveRaytracingPipeline.cpp:215-247) of BASELINE.json configs[1]: the
Sponza-class synthetic atrium (~262k triangles) at 1920x1080, one sample per pixel
per frame, reference default material / lights / options, everything resident in HBM.
Every frame is issued as the reference's app loop issues it -- one rt_pipeline_update +
one rt_pipeline_render (src/DXRExperimentsApp.cpp:162-165, :194); the pipeline is in
deferred mode (rt_pipeline_set_deferred, the default of the C++ mirror), which renders
the recorded frames through shared sets of launches, bit for bit the same image.
`frame_by_frame` is the same run with deferred mode off.
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
# Ceilings of the vector-memory path for this engine's access shape (every lane its own pseudo-random 64-B line), calibrated with
# the counters of round 3 on tools/microbench/slab_fetch (profiles/r03/slab_131072_tcp.md + slab_kt.md, 8 MiB table):
#   L1 -> L2 read requests: one per 64-B line a lane gathers (TCP_TCC_READ_REQ / lane-line = 0.996); the chip sustains
#   120.6 G of them per second (mode 3: 411.26 M requests per 3409 us) = 7.7 TB/s of 64-B lines (round 2 measured 7.5);
#   L1 tag accesses: one per 16-B load per lane (TCP_TOTAL_CACHE_ACCESSES / lane-load = 1.000), ceiling one per clock per CU.
L2_REQUEST_PEAK_PER_S = 120.6e9
GATHER_PEAK_GBS = L2_REQUEST_PEAK_PER_S * 64 / 1e9
N_CU, N_SIMD, N_XCD = 256, 1024, 8
VALU_ISSUE_CYCLES = 2.0        # a wave64 VALU instruction occupies a SIMD-32 for 2 cycles (MI355X_MICROARCH.md, "Wave scheduling") -- the nominal figure
# ... which only v_add / v_sub / v_mul / v_mov and the simple integer forms reach.  Measured with six to eight waves per SIMD
# issuing independent instructions (tools/microbench/valu_rate under rocprofv3, cycles = GRBM_GUI_ACTIVE / 8:
# profiles/r03/valu_rate_pmc.md, valu_rate_pmc_w6.md): those take 2.4, v_fma_f32, v_min / v_max and their three-operand forms,
# v_cmp, v_cndmask, v_cvt_f32_ubyte, v_lshl_or take 4.2 - 4.5 (the packed f32 forms 4.7 for two operations) -- but the costs
# do not add up in a mixed stream (v_cvt + v_fma alternating: 2.8 each).  So the ceiling is measured on the mix itself: a
# synthetic stream with the class shares of the traversal kernels (add 9, mul 9, fma 16, cvt 12, int 16, min / max / compare /
# select 38 %: mode "traversal-kernel mix") issues one instruction per VALU_MIX_CYCLES per SIMD at the kernels' six waves per
# SIMD (2.66 at eight, 2.71 at five, 3.25 at three, 7.9 for one wave alone).  The live class counters (SQ_INSTS_VALU_*) are
# reported beside it so that the shares can be compared; the additive per-class prices (an upper bound on the cost) likewise.
VALU_MIX_CYCLES = 2.65
HBM_SET = 16                   # frames per set of launches of the 10 M-triangle workload (8: 12.3 ms per frame, 16: 12.0; 4 GB of queues per frame)
VALU_MIX_STREAM_SHARES = {"ADD_F32": 3 / 32, "MUL_F32": 3 / 32, "FMA_F32": 5 / 32, "CVT": 4 / 32, "INT32": 5 / 32, "TRANS_F32": 0.0, "other": 12 / 32}
VALU_CLASS_CYCLES = {"SQ_INSTS_VALU_ADD_F32": 2.0, "SQ_INSTS_VALU_MUL_F32": 2.0, "SQ_INSTS_VALU_FMA_F32": 4.0, "SQ_INSTS_VALU_CVT": 4.0,
                     "SQ_INSTS_VALU_INT32": 2.0, "SQ_INSTS_VALU_TRANS_F32": 8.0}
VALU_OTHER_CYCLES = 4.0
LINE_BYTES = 64
RAY_BYTES, NODE_BYTES, TRI_BYTES = 48, 32, 36   # 32 B ray in + 16 B hit out; node; triangle
# what the production kernels request per unit (DESIGN.md sections 3 / 4): one 64-B node per step that is not served by the
# LDS-resident top of the tree, one 48-B record per triangle test, the 96-B traversal prefix of an instance record per instance entry, and the
# ray in / result out of the stage
NODE_LINE_BYTES, TRIREC_BYTES, INSTANCE_BYTES = 64, 48, 96
STAGE_IO_BYTES = {"primary": 0 + 20, "secondary": 32 + 20, "shadow": 32 + 4}
# stage -> (rt_stats time fields, kernel, rt_pipeline_count_work stages whose rays the launch traces)
TRACE_STAGES = {"primary": (("ms_primary",), "k_primary", ("primary",)),
                "secondary": (("ms_trace_secondary",), "k_trace_secondary", ("secondary",)),
                "shadow": (("ms_trace_shadow0", "ms_trace_shadow1"), "k_trace_shadow", ("shadow0", "shadow1"))}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline sample budget (rank 0, N=1 only); 0 = skip")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-frame-by-frame", "--no-sample-batches", dest="no_sample_batches", action="store_true",
                    help="skip the frame_by_frame section (profiling passes: keeps the per-kernel averages to one kind of launch)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not re-run a few frames under rocprofv3 for roofline.traffic; use the "
                    "committed profiles/<round>/traffic.json only")
    ap.add_argument("--hbm-frames", type=int, default=16, help="frames of the HBM-bound 10 M-triangle 4K workload timed for roofline_hbm "
                    "(rank 0, N=1 only); 0 = skip")
    ap.add_argument("--batch", type=int, default=32, help="frames per set of launches (rt_pipeline_render_batch, at most 32: the sample-batch mode "
                    "of BASELINE configs[2]; the K timed frames are split evenly over ceil(K / batch) sets); 1 = one update() + render() "
                    "per frame, which the default run also measures and reports as `frame_by_frame`")
    ap.add_argument("--obj", default=None, help="render this Wavefront OBJ instead of the procedural atrium (e.g. the real Sponza); "
                    "config.workload then names the file and its triangle count")
    ap.add_argument("--scene", choices=("atrium", "stadium", "stadium2m"), default="atrium",
                    help="atrium: BASELINE configs[1], the headline (default).  stadium / stadium2m: the stress scene with real-asset triangle "
                         "statistics (scenes.stadium_class: huge quads under dense detail, slivers, log-normal areas; ~272 k / ~2.2 M triangles) -- "
                         "a comparison line, not the headline")
    ap.add_argument("--camera", type=float, nargs=6, default=None, metavar=("EX", "EY", "EZ", "AX", "AY", "AZ"),
                    help="eye and look-at point for --obj (default: a view from outside the mesh's bounding box towards its centre)")
    ap.add_argument("--fov", type=float, default=None, help="vertical field of view in radians for --obj (default pi/4, src/utils/Camera.h:141-144)")
    ap.add_argument("--workload", choices=("c2", "c5"), default="c2", help="c5: ONLY the 10 M-triangle workload (profiling passes)")
    ap.add_argument("--total-frames", type=int, default=None, metavar="T",
                    help="BASELINE configs[2] as written: T frames IN ALL (256 spp) sharded over the --gpus ranks, each rank's ceil(T / N) in deferred "
                         "sets, ONE all-reduce: that fixed-total run becomes the line's value (\"scaling\": \"strong\") and the K-frames-per-GPU run "
                         "moves to \"weak_scaling\".  Without the flag the line is the weak one and carries the 256-frame run as \"strong_scaling\"")
    ap.add_argument("--no-strong", action="store_true", help="skip the fixed-total pass")
    ap.add_argument("--partition", choices=("samples", "tiles"), default="samples",
                    help="samples (default, the headline): frames sharded over the GPUs, one all-reduce.  tiles: BASELINE configs[4], the "
                         "10 M-triangle 4K 4-bounce frame split into interleaved 16-row bands over the GPUs, one all-gather of the bands")
    args = ap.parse_args(argv)
    args.batch_given = any(a == "--batch" or a.startswith("--batch=") for a in (sys.argv[1:] if argv is None else argv))
    return args


def fixed_total_plan(rank, world, total_frames, batch):
    """Host logic of the fixed-total (strong-scaling) pass, BASELINE configs[2]: the frames of ONE global sequence rank `rank` renders
    ({f : f mod world == rank}: dxrexperiments_amd.distributed.shard_frames = rt_shard_frame_count), the sets of launches they go
    through and the frames per set (at most 32, sets of almost equal size)."""
    mine = list(range(rank, total_frames, world))
    per = max(1, min(batch, 32))
    sets = (len(mine) + per - 1) // per
    per_set = (len(mine) + sets - 1) // sets if sets else 1
    return mine, sets, per_set


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


def committed_profile(workload):
    """The newest committed rocprofv3 summary for `workload` ("c2" / "c5"): profiles/<round>/traffic.json, written by
    tools/traffic_from_pmc.py from PMC passes around this same bench command (FETCH_SIZE, WRITE_SIZE and the SQ / TCC
    counters need separate passes, so they cannot be taken inside this run).  {} if absent."""
    here = os.path.dirname(os.path.abspath(__file__))
    for rnd in sorted(os.listdir(os.path.join(here, "profiles")), reverse=True) if os.path.isdir(os.path.join(here, "profiles")) else []:
        path = os.path.join(here, "profiles", rnd, "traffic.json")
        if os.path.isfile(path):
            with open(path) as f:
                d = json.load(f)
            w = d.get("workloads", {}).get(workload)
            if w:
                return dict(w, commit=d.get("commit"), source=path[len(here) + 1:])
    return {}


LIVE_PMC_NOTES = []        # why a live pass was not used (the committed profile is the fallback; reported as roofline.traffic_fallback)
LIVE_PMC_PASSES = {"ea": ["TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"], "write": ["WRITE_SIZE"],
                   # round 3: what the SIMDs issue and where the waves' cycles go (SQ; GRBM_GUI_ACTIVE has slots of its own), and what
                   # the lanes ask of the L1 and the L1 of the L2 (TCP)
                   "sq": ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"],
                   "lanes": ["SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"],
                   "tcp": ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_LATENCY_sum"],
                   # the vector instructions by class (SQ_INSTS_VALU_MIX is SQ_INSTS_VALU of the same pass: shares are taken within one pass)
                   "mix": ["SQ_INSTS_VALU"] + sorted(VALU_CLASS_CYCLES)}


def live_traffic(workload, width, height, budget_s=110.0, passes=("ea", "write", "sq", "tcp", "lanes"), per_set=1):
    """Memory-side bytes per launch of the traversal kernels, measured IN THIS RUN: bench.py re-runs itself for a few frames
    as a child of `rocprofv3 --kernel-trace --pmc ...`, one pass for the L2's read requests by size (32 / 64 / 128 B: the
    calibrated byte count for this access shape, MI355X_MICROARCH.md HBM section) and one for WRITE_SIZE (they do not fit one
    pass).  Called before this process touches the GPU.  {kernel: {"bytes_per_launch", ...}} or {} if rocprofv3 is not
    usable here (the committed profiles/<round>/traffic.json is the fallback)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {}
    here = os.path.dirname(os.path.abspath(__file__))
    child = [os.path.join(here, "bench.py"), "--cpu-seconds", "0", "--no-roofline", "--no-live-pmc", "--no-frame-by-frame"]
    child += (["--steps", str(8 if per_set == 1 else 4 * per_set), "--warmup", str(2 if per_set == 1 else per_set), "--batch", str(per_set),
               "--hbm-frames", "0", "--width", str(width), "--height", str(height)] if workload == "c2"
              else ["--workload", "c5", "--hbm-frames", "2"] if per_set == 1 else ["--workload", "c5", "--hbm-frames", str(per_set), "--batch", str(per_set)])
    out, t0 = {}, time.perf_counter()
    env = dict(os.environ, TMPDIR="/tmp")
    for name in passes:
        counters = LIVE_PMC_PASSES[name]
        left = budget_s - (time.perf_counter() - t0)
        if left < 10.0:
            LIVE_PMC_NOTES.append("%s/%s: budget of %.0f s spent before the pass" % (workload, name, budget_s))
            break
        d = tempfile.mkdtemp(prefix="dxr_pmc_")
        try:
            # (the program itself directly behind `--`: the profiler's preloaded library initialises the GPU first)
            r = subprocess.run([exe, "--kernel-trace", "--pmc"] + counters + ["-d", d, "-o", "p", "--output-format", "csv", "--",
                                sys.executable, *child], cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=min(240.0, left))
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                LIVE_PMC_NOTES.append("%s/%s: rocprofv3 exit code %d, %d counter files" % (workload, name, r.returncode, len(files)))
                break
            acc = {}
            for f in files:
                for row in csv.DictReader(open(f)):
                    k = next((k for k in ("k_trace_secondary", "k_trace_shadow", "k_primary") if k in row["Kernel_Name"]), None)
                    if k:
                        a = acc.setdefault((k, row["Counter_Name"]), [0, 0.0])
                        a[0] += 1
                        a[1] += float(row["Counter_Value"])
            for (k, c), (n, v) in acc.items():
                out.setdefault(k, {})["SQ_INSTS_VALU_MIX" if name == "mix" and c == "SQ_INSTS_VALU" else c] = v / n
                out[k]["dispatches"] = n
            if name == "sq":                 # the same pass's launch durations: with GRBM_GUI_ACTIVE they give the clock under this load
                dur = {}
                for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        k = next((k for k in ("k_trace_secondary", "k_trace_shadow", "k_primary") if k in row["Kernel_Name"]), None)
                        if k:
                            a = dur.setdefault(k, [0, 0.0])
                            a[0] += 1
                            a[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                for k, (n, ns) in dur.items():
                    out.setdefault(k, {})["sq_pass_avg_us"] = ns / n / 1e3
        except (OSError, subprocess.SubprocessError, KeyError, ValueError) as e:
            LIVE_PMC_NOTES.append("%s/%s: %s" % (workload, name, type(e).__name__))
            break
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for k, e in out.items():
        if all(c in e for c in LIVE_PMC_PASSES["ea"]) and "WRITE_SIZE" in e:
            e["read_bytes_by_request_size"] = int(32 * e["TCC_EA0_RDREQ_32B_sum"] + 64 * e["TCC_EA0_RDREQ_64B_sum"] + 128 * e["TCC_EA0_RDREQ_128B_sum"])
            e["bytes_per_launch"] = int(e["read_bytes_by_request_size"] + e["WRITE_SIZE"] * 1024)      # rocprofv3 reports WRITE_SIZE in KiB
    return out         # (a pass that failed leaves its counters out: the committed profile fills in, roofline.traffic_fallback says why)


def walk_bytes(stage, w):
    """Bytes the lanes of a traversal stage request (per lane, before any sharing between lanes), from the production-walk
    tallies of rt_pipeline_count_walk."""
    return (NODE_LINE_BYTES * w["nodes_global"] + TRIREC_BYTES * w["tris"] + INSTANCE_BYTES * w["instance_entries"]
            + STAGE_IO_BYTES[stage] * w["rays"])


def gather_bytes(stage, w):
    """Bytes of DISTINCT 64-B lines a stage gathers: node lines de-duplicated over the lanes of every wave step, the lines
    the triangle records span, two lines per instance record, and the (coalesced) ray in / result out."""
    return LINE_BYTES * (w["lines"] + 2 * w["instance_entries"]) + STAGE_IO_BYTES[stage] * w["rays"]


def stage_table(pipe, tot, with_canonical=True, levels=1):
    """Per traversal stage: HIP-event time (average over the timed launches), rays, the production walk's requested bytes
    and -- as SURVEY 8(d)'s layout-independent contract figure -- the canonical-LBVH counters' algorithmic bytes."""
    walk = pipe.count_walk()
    work = pipe.count_work() if with_canonical else None
    n_t = max(int(tot["frames"]), 1)
    stages = {}
    for name, (keys, kernel, parts) in TRACE_STAGES.items():
        wk = {k: sum(walk[p][k] for p in parts) for k in ("rays", "nodes_global", "nodes_lds", "tris", "instance_entries", "lines", "node_lines",
                                                          "wave_node_steps", "wave_leaf_phases", "wave_tri_steps")}
        wk["longest_walk"] = max(walk[p]["longest_walk"] for p in parts)
        ms = sum(tot[k] for k in keys) / n_t
        rb, gb = walk_bytes(name, wk), gather_bytes(name, wk)
        st = {"kernel": kernel, "avg_ms": ms, "rays": wk["rays"], "walk": wk, "requested_bytes": rb, "gathered_bytes": gb,
              "requested_GBps": rb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0, "gathered_GBps": gb / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
              "nodes_global_per_ray": wk["nodes_global"] / max(wk["rays"], 1), "nodes_lds_per_ray": wk["nodes_lds"] / max(wk["rays"], 1),
              "tris_per_ray": wk["tris"] / max(wk["rays"], 1), "instance_entries_per_ray": wk["instance_entries"] / max(wk["rays"], 1),
              "lines_per_ray": wk["lines"] / max(wk["rays"], 1), "longest_walk_steps": wk["longest_walk"],
              "launches_per_frame": levels if name == "secondary" else 1,
              "Mrays_per_s": wk["rays"] / (ms * 1e-3) / 1e6 if ms > 0 else 0.0}
        if work is not None:
            w = {k: sum(work[p][k] for p in parts) for k in ("rays", "nodes", "tris")}
            b = RAY_BYTES * w["rays"] + NODE_BYTES * w["nodes"] + TRI_BYTES * w["tris"]
            st["canonical"] = {"algorithmic_bytes": b, "nodes_per_ray": w["nodes"] / max(w["rays"], 1), "tris_per_ray": w["tris"] / max(w["rays"], 1),
                               "GBps": b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0}
        stages[name] = st
    for key in ("ms_shade0", "ms_shade1", "ms_resolve", "ms_total"):
        stages[key] = tot[key] / n_t
    return stages, n_t


def lane_utilisation(walk_stage):
    """Live lanes per wave instruction in the two halves of the walk, from the counting re-walk's tallies (rt_pipeline_count_walk:
    the same refill / straggler-exit logic as the timed kernel, static instead of pooled chunk hand-out)."""
    w = walk_stage
    node = (w["nodes_global"] + w["nodes_lds"]) / max(64 * w["wave_node_steps"], 1)
    tri = w["tris"] / max(64 * w["wave_tri_steps"], 1)
    return {"node_steps": node, "triangle_steps": tri, "source": "rt_pipeline_count_walk: lanes live / (64 x wave steps) of ONE frame walked by itself",
            "wave_node_steps": w["wave_node_steps"], "wave_triangle_steps": w["wave_tri_steps"], "wave_leaf_phases": w["wave_leaf_phases"]}


def headline_roofline(args, stages, dom, n_t, n_sets, S, live, W, H, ms_per_step, scene_bytes):
    """The contract's roofline of the dominant kernel (task 1): bound "hbm"; `traffic` = HBM-side bytes per launch
    from the PMC passes (read requests by size + WRITE_SIZE, MI355X_MICROARCH.md's HBM section); `achieved` = traffic / this
    run's HIP-event duration of the same launch; `peak` = 8 TB/s; frac <= 1 by construction.  Beside it: the production walk's
    own minimum bytes (64 B x distinct node lines per wave step + 48 B x triangle records + ray in / result out) and
    traffic / minimum as the wasted-traffic ratio; the vector-issue reading with BOTH denominators (the nominal 2 cycles per
    wave64 instruction and the measured 2.65 of this instruction mix); the active-lane fraction; where the waves' cycles go."""
    d = stages[dom]
    prof_name = "c2" if S == 1 else ("c2s" if S <= 24 and committed_profile("c2s") else "c2b")
    # (the committed counters are those of the default workload: they stand in for live ones only when this run is that workload)
    default_workload = (W, H) == (1920, 1080) and not args.obj and not args.camera and args.scene == "atrium"
    prof = committed_profile(prof_name).get("kernels", {}).get(d["kernel"], {}) if default_workload else {}
    lv = live.get("c2", {}).get(d["kernel"], {})
    # a launch covers a set of frames: stage times are per frame (stage_table), hardware counters per launch
    fpl = float(S) if lv else float(prof.get("frames_per_launch", 1))     # frames per launch in the counter passes
    frames_per_launch = n_t / n_sets                                       # ... in this run's timed region
    launch_ms = d["avg_ms"] * frames_per_launch                            # this run's average launch of the stage (HIP events)
    pm = dict(prof)
    pm.update({k: v for k, v in lv.items() if not isinstance(v, str)})
    traffic_pass = lv.get("bytes_per_launch", prof.get("bytes_per_launch"))            # bytes per launch of the counter pass
    traffic = traffic_pass / fpl * frames_per_launch if traffic_pass else None        # ... per launch of THIS run
    # the production walk's minimum for the launch: what ONE frame's walk must fetch x the frames of the launch
    wk = d["walk"]
    min_frame = NODE_LINE_BYTES * wk["node_lines"] + TRIREC_BYTES * wk["tris"] + INSTANCE_BYTES * wk["instance_entries"] + STAGE_IO_BYTES[dom] * wk["rays"]
    algorithmic = min_frame * frames_per_launch
    rl = {"bound": "hbm", "kernel": d["kernel"], "stage": dom, "unit": "GB/s", "peak": HBM_PEAK_GBS,
          "achieved": traffic / (launch_ms * 1e-3) / 1e9 if traffic else None,
          "frac": traffic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
          "traffic": traffic,
          "algorithmic_bytes": algorithmic, "algorithmic_GBps": algorithmic / (launch_ms * 1e-3) / 1e9,
          "algorithmic_frac": algorithmic / (launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
          "wasted_traffic_ratio": traffic / algorithmic if traffic and algorithmic else None,
          # what perfect caches would leave: the traversal arrays read ONCE per launch + the launch's rays in and results out
          "compulsory_bytes": scene_bytes + STAGE_IO_BYTES[dom] * wk["rays"] * frames_per_launch,
          "traffic_over_compulsory": traffic / (scene_bytes + STAGE_IO_BYTES[dom] * wk["rays"] * frames_per_launch) if traffic else None,
          "avg_launch_ms": launch_ms, "launches_timed": n_sets, "frames_per_launch": frames_per_launch, "avg_ms_per_frame": d["avg_ms"],
          "frames_per_launch_in_counter_passes": fpl,
          "traffic_source": ("this run: %d launches under rocprofv3 --pmc (read requests by size, WRITE_SIZE)" % lv["dispatches"])
                            if "bytes_per_launch" in lv else "committed profile %s" % committed_profile(prof_name).get("source"),
          "traffic_fallback": LIVE_PMC_NOTES or None,
          "algorithmic_definition": "per frame: 64 B x node lines (distinct per wave step, the LDS-resident top not counted) + 48 B x triangle "
                                    "records + 96 B x instance entries + %d B ray in / result out per ray, counted by rt_pipeline_count_walk on the "
                                    "production tree; x frames per launch.  wasted_traffic_ratio = traffic / algorithmic_bytes (below 1: the L1s and L2s "
                                    "serve part of it); traffic_over_compulsory = traffic / (the %d B of nodes and triangle records once + the rays' "
                                    "in / out bytes): how often the scene crosses the fabric per launch" % (STAGE_IO_BYTES[dom], scene_bytes),
          "definition": "frac = achieved / peak; achieved = traffic / avg_launch_ms; traffic = HBM-side bytes of one launch of this kernel = "
                        "32 x TCC_EA0_RDREQ_32B + 64 x TCC_EA0_RDREQ_64B + 128 x TCC_EA0_RDREQ_128B + 1024 x WRITE_SIZE (rocprofv3 --pmc, separate "
                        "passes), scaled from the counter passes' frames per launch to this run's; avg_launch_ms = HIP events around the "
                        "stage on the context's stream, all timed launches; peak = 8 TB/s (MI355X_MICROARCH.md)"}
    rl["lane_utilisation"] = lane_utilisation(wk)
    have_sq = all(k in pm for k in ("SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE")) and (pm.get("sq_pass_avg_us") or pm.get("avg_us"))
    if have_sq:
        clk_hz = pm["GRBM_GUI_ACTIVE"] / N_XCD / ((pm.get("sq_pass_avg_us") or pm["avg_us"]) * 1e-6)
        insts_per_s = pm["SQ_INSTS_VALU"] / fpl / (d["avg_ms"] * 1e-3)          # (counters per launch -> per frame; stage time per frame)
        vi = {"unit": "G wave64 VALU instructions/s", "achieved": insts_per_s / 1e9,
              "peak_nominal": N_SIMD * clk_hz / VALU_ISSUE_CYCLES / 1e9, "frac_nominal": insts_per_s / (N_SIMD * clk_hz / VALU_ISSUE_CYCLES),
              "peak_measured_mix": N_SIMD * clk_hz / VALU_MIX_CYCLES / 1e9, "frac_measured_mix": insts_per_s / (N_SIMD * clk_hz / VALU_MIX_CYCLES),
              "clock_GHz_under_load": clk_hz / 1e9,
              "denominators": "nominal: one wave64 VALU instruction per 2 cycles per SIMD (MI355X_MICROARCH.md); measured_mix: %.2f cycles, "
                              "a synthetic stream with this kernel's instruction-class shares at six waves per SIMD "
                              "(tools/microbench/valu_rate, profiles/r03/valu_rate_pmc_w6.md) -- a ceiling of the builder's own measuring, "
                              "quoted beside the nominal one, never instead of it" % VALU_MIX_CYCLES}
        # useful work: issue fraction x live lanes (node steps dominate the instruction count)
        lu = rl["lane_utilisation"]
        steps_w = 150.0 * lu["wave_node_steps"] + 120.0 * lu["wave_triangle_steps"]
        if steps_w > 0:
            mean_lanes = (150.0 * lu["wave_node_steps"] * lu["node_steps"] + 120.0 * lu["wave_triangle_steps"] * lu["triangle_steps"]) / steps_w
            vi["mean_live_lane_fraction"] = mean_lanes
            vi["frac_nominal_x_live_lanes"] = vi["frac_nominal"] * mean_lanes
        if "SQ_THREAD_CYCLES_VALU" in lv and lv.get("SQ_ACTIVE_INST_VALU"):          # (live: the "lanes" pass came last, both counters are its own)
            rl["lane_utilisation"]["pmc_thread_cycles_over_64x_active_inst_valu"] = lv["SQ_THREAD_CYCLES_VALU"] / (64.0 * lv["SQ_ACTIVE_INST_VALU"])
        elif pm.get("active_lane_fraction"):
            rl["lane_utilisation"]["pmc_thread_cycles_over_64x_active_inst_valu"] = pm["active_lane_fraction"]
        rl["valu_issue"] = vi
        wc = pm["SQ_WAVE_CYCLES"]
        rl["wave_cycles"] = {"parked_on_waitcnt": pm.get("SQ_WAIT_ANY", 0.0) / wc, "issue_stalled": pm.get("SQ_WAIT_INST_ANY", 0.0) / wc,
                             "issuing": pm.get("SQ_ACTIVE_INST_ANY", 0.0) / wc}
        rl["counters_source"] = ("this run (rocprofv3 --pmc child passes)" if any(c in lv for c in LIVE_PMC_PASSES["sq"]) else
                                 "committed profile %s" % committed_profile(prof_name).get("source"))
    mp = {"lines_counted_by_count_walk": d["gathered_bytes"] // LINE_BYTES, "gathered_GBps_count_walk": d["gathered_GBps"]}
    if "TCP_TCC_READ_REQ_sum" in pm:
        req = pm["TCP_TCC_READ_REQ_sum"]
        mp.update({"l2_read_requests_per_launch": req, "l2_request_rate_frac": req / fpl / (d["avg_ms"] * 1e-3) / L2_REQUEST_PEAK_PER_S,
                   "l2_request_GBps": req / fpl * LINE_BYTES / (d["avg_ms"] * 1e-3) / 1e9, "l2_request_peak_GBps": GATHER_PEAK_GBS,
                   "l1_served_share_of_counted_lines": 1.0 - req / fpl / max(d["gathered_bytes"] / LINE_BYTES, 1.0)})
        if "TCP_TCC_READ_REQ_LATENCY_sum" in pm and req:
            mp["l1_to_l2_read_latency_cycles"] = pm["TCP_TCC_READ_REQ_LATENCY_sum"] / req
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in pm and have_sq:
        mp["l1_tag_accesses_per_launch"] = pm["TCP_TOTAL_CACHE_ACCESSES_sum"]
        mp["l1_tag_rate_frac"] = pm["TCP_TOTAL_CACHE_ACCESSES_sum"] / fpl / (d["avg_ms"] * 1e-3) / (N_CU * clk_hz)
    rl["memory_path"] = mp
    # SURVEY 8(d)'s unit as written (canonical binary-LBVH counters x 32 / 36 / 48 B): kept as a label -- it prices a tree the timed kernel
    # does not walk and comes out above the HBM peak (What's weak 2)
    if "canonical" in d:
        # ... and as one number under the name the contract uses: SURVEY 8(d)'s bytes per ray x rays / the kernel's time / the HBM peak
        rl["survey_8d_frac"] = d["canonical"]["GBps"] / HBM_PEAK_GBS
        rl["survey_8d_frac_note"] = ("above 1 not because work is skipped (whole frames are bit-exact against the oracle) but because SURVEY 8(d) prices the walk of the "
                                     "canonical binary LBVH (32 B x nodes + 36 B x triangles by the oracle's counters) while the timed kernel walks the production tree "
                                     "(PLOC, four-wide 64-B nodes, top levels in LDS: a third of the node visits); `frac` is the fraction of the HBM peak the kernel's own "
                                     "fabric traffic reaches")
        rl["survey_8d_unit_as_written"] = {"bytes_per_frame": d["canonical"]["algorithmic_bytes"], "GBps": d["canonical"]["GBps"],
                                           "over_hbm_peak": d["canonical"]["GBps"] / HBM_PEAK_GBS}
    rl["pmc"] = {k: pm.get(k) for k in ("TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_WAIT_ANY",
                                        "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE",
                                        "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_LATENCY_sum") if k in pm}
    tb = sum(stages[s_]["gathered_bytes"] for s_ in TRACE_STAGES)
    rl["all_stages_gathered_GBps_over_step"] = tb / (ms_per_step * 1e-3) / 1e9
    return rl


def hbm_workload(ctx, capi, T, scenes, frames, warm, batch=1, one_frame_walk=True):
    """The one configuration whose traversal working set (~0.8 GB of slabs + triangle records) does not fit the 256 MB
    Infinity Cache: BASELINE configs[4], the 10 M-triangle mesh at 3840x2160 with 4 radiance bounces (SURVEY 7 "Roofline
    honesty").  Rendered on the same context after the headline measurement; returns the stage table of `frames` frames."""
    import numpy as np
    W, H = 3840, 2160
    t0 = time.perf_counter()
    v, tri = scenes.displaced_grid(2236, seed=7)
    gen_s = time.perf_counter() - t0
    model = capi.Model(ctx, v, tri)
    scene = capi.Scene(ctx)
    scene.add_model(model)
    pipe = capi.Pipeline(ctx)
    pipe.set_scene(scene)
    mat = T.default_material()
    mat["type"] = 2
    mat["reflectivity"] = 0.6
    mat["roughness"] = 0.3
    pipe.add_material(mat)
    pipe.set_depth_limits(4, 2)
    pipe.set_environment_cube(scenes.sky_cubemap(32))
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(3)
    host.options["maxIterations"] = 1 << 20
    cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
    if batch <= 1:
        for f in range(warm):
            pipe.update(host.update(cam, 0.0, f + 1, W, H))
            pipe.render()
    else:                   # (the first batch sizes the queues for `batch` frames: not part of the timed region)
        pipe.render_batch([host.update(cam, 0.0, f + 1, W, H) for f in range(max(warm, batch))])
        warm = max(warm, batch)
    ctx.synchronize()
    pipe.enable_timing(frames)
    pipe.reset_totals()
    t0 = time.perf_counter()
    if batch <= 1:
        for f in range(warm, warm + frames):
            pipe.update(host.update(cam, 0.0, f + 1, W, H))
            pipe.render()
    else:
        for f0 in range(warm, warm + frames, batch):
            pipe.render_batch([host.update(cam, 0.0, f + 1, W, H) for f in range(f0, min(f0 + batch, warm + frames))])
    ctx.synchronize()
    dt = time.perf_counter() - t0
    tot = pipe.totals()
    if batch > 1 and one_frame_walk:           # (the walk counters of the stage table are those of ONE frame: the next one, by itself, after the timed region;
                                               #  not under the profiler: every launch of a counter pass covers a whole set)
        pipe.update(host.update(cam, 0.0, warm + frames + 1, W, H))
        pipe.render()
    stages, n_t = stage_table(pipe, tot, with_canonical=False, levels=4)
    rays = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"] - tot["rays_shadow_skipped"]      # rays actually traversed
    return {"workload": "BASELINE configs[4] on one GPU: displaced-grid mesh (seed 7, %d triangles), %dx%d, 4 radiance bounces, "
                        "1 spp/frame" % (tri.shape[0], W, H),
            "frames": frames, "frames_per_launch_set": min(batch, frames) if batch > 1 else 1, "ms_per_frame": dt / frames * 1e3, "Mrays_per_s": rays / dt / 1e6, "rays_per_frame": rays / frames,
            "bvh_build_ms": scene.build_ms(), "generate_s": gen_s, "stages": stages, "launches_timed": n_t}



def headline_workload(args, capi, scenes, np):
    """(verts, tris, description, camera dict) of the headline run: the procedural Sponza-class atrium, or the mesh of --obj
    (the product's own OBJ reader, no device needed) seen from --camera or from outside its bounding box."""
    if args.obj:
        verts, tris = capi.obj_read(args.obj)
        workload = "user OBJ %s (%d triangles, %d vertices)" % (os.path.basename(args.obj), tris.shape[0], verts.shape[0])
        lo, hi = verts["position"].min(axis=0), verts["position"].max(axis=0)
        centre, ext = 0.5 * (lo + hi), float(np.linalg.norm(hi - lo))
        eye, at = (args.camera[:3], args.camera[3:]) if args.camera else (centre + np.array([0.35, 0.2, 1.0]) * ext, centre)
        cam = dict(eye=tuple(float(x) for x in eye), at=tuple(float(x) for x in at), up=(0.0, 1.0, 0.0),
                   fov=args.fov if args.fov else float(np.float32(np.pi / 4)))
    elif getattr(args, "scene", "atrium") != "atrium":
        verts, tris = scenes.stadium_class(seed=5, scale=8.0 if args.scene == "stadium2m" else 1.0)
        workload = "stress scene with real-asset triangle statistics (scenes.stadium_class, seed 5, %d triangles; NOT the headline workload)" % tris.shape[0]
        cam = scenes.stadium_camera()
    else:
        verts, tris = scenes.sponza_class(seed=42)
        workload = "BASELINE configs[1]: Sponza-class procedural atrium (seed 42, %d triangles)" % tris.shape[0]
        cam = scenes.sponza_camera()
    return verts, tris, workload, cam


def rank_report(torch, dist, ctx, rank, world, local_rank, dev, backend, elapsed, collective_ms):
    """What makes an N > 1 line self-verifying (task 8): the world size the process group really has, the backend
    and RCCL version, every rank's device ordinal and PCI bus id (two ranks on one GPU show up as a repeated id), the
    HIP-event time of the collective per rank, and the spread of the per-rank elapsed times.  Gathered with one SUM
    all-reduce of a table in which every rank fills its own row (works over RCCL and gloo alike)."""
    bus = ctx.pci_bus_id()
    try:
        dom, b, rest = bus.split(":")
        d_, fn = rest.split(".")
        nums = [int(dom, 16), int(b, 16), int(d_, 16), int(fn, 16)]
    except ValueError:
        nums = [0, 0, 0, 0]
    table = torch.zeros((world, 8), dtype=torch.float64, device=dev)
    table[rank] = torch.tensor([float(local_rank)] + [float(x) for x in nums] + [elapsed, collective_ms, float(os.getpid())], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(table, op=dist.ReduceOp.SUM)
    rows = table.cpu().tolist()
    ids = ["%04x:%02x:%02x.%x" % (int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in rows]
    try:
        rccl = ".".join(str(x) for x in torch.cuda.nccl.version()) if backend == "nccl" else None
    except Exception:                                            # (no RCCL in this torch build)
        rccl = None
    return {"ranks_seen": dist.get_world_size() if world > 1 else 1, "backend": (dist.get_backend() if world > 1 else "none"),
            "rccl_version": rccl, "device_ordinals": [int(r[0]) for r in rows], "pci_bus_ids": ids,
            "distinct_devices": len(set(ids)), "collective_ms": [r[6] for r in rows],
            "elapsed_s_min": min(r[5] for r in rows), "elapsed_s_max": max(r[5] for r in rows), "pids": [int(r[7]) for r in rows]}

def tiles_main(args, rank, world, local_rank, dev, capi, D, T, scenes, torch, dist):
    """BASELINE configs[4]: the 10 M-triangle mesh at 3840x2160 with 4 radiance bounces, tile-partitioned:
    every rank renders its interleaved 16-row bands of EVERY frame (pixels are seeded by their global index, so the bands
    equal the same rows of the whole frame bit for bit) and ONE all-gather of the disjoint bands closes the timed region.
    Strong scaling: the frame is fixed, the rows per GPU shrink as N grows."""
    W, H, K, Wu, band = 3840, 2160, args.steps, args.warmup, 16
    v, tri = scenes.displaced_grid(2236, seed=7)
    ctx = capi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
    scene = capi.Scene(ctx)
    scene.add_model(capi.Model(ctx, v, tri))
    pipe = capi.Pipeline(ctx)
    pipe.set_scene(scene)
    mat = T.default_material()
    mat["type"] = 2
    mat["reflectivity"] = 0.6
    mat["roughness"] = 0.3
    pipe.add_material(mat)
    pipe.set_depth_limits(4, 2)
    pipe.set_environment_cube(scenes.sky_cubemap(32))
    acc = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
    pipe.bind_output(acc.data_ptr(), W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(3)
    host.options["maxIterations"] = 1 << 20
    cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
    pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(Wu + K)]

    S = max(1, min(args.batch if args.batch_given else 32, 32))       # frames per set of launches (1: one set per frame, round 3 form)

    def steps(lo, hi):
        """frames lo..hi-1: this rank's bands of S frames at a time through shared sets of launches (rt_pipeline_render_bands_batch)"""
        if S == 1:
            for i in range(lo, hi):
                pipe.update(pfcs[i])
                pipe.render_bands(band, rank, world)          # all of this rank's bands of ONE frame in one set of launches
        else:
            for i in range(lo, hi, S):
                pipe.render_bands_batch(band, rank, world, pfcs[i:min(i + S, hi)])

    if S > 1:                                         # the work memory of a full set of this rank's bands: outside the timed region
        pipe.reserve_batch(min(S, K), rows=sum(1 for r in range(H) if (r // band) % world == rank))
    steps(0, Wu)
    if world > 1:
        D.gather_tiles(acc.clone(), band)             # warm the collective
        dist.barrier()
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))       # (first use before the timed region, as in main())
    ev[0].record(); ev[1].record()
    torch.cuda.synchronize()
    ev[0].elapsed_time(ev[1])
    pipe.reset_totals()
    t0 = time.perf_counter()
    steps(Wu, Wu + K)
    ev[0].record()
    if world > 1:
        D.gather_tiles(acc, band)
    ev[1].record()
    torch.cuda.synchronize()
    mine_elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    report = rank_report(torch, dist, ctx, rank, world, local_rank, dev, os.environ.get("DXR_BENCH_BACKEND", "nccl"), mine_elapsed, ev[0].elapsed_time(ev[1]))
    tot = pipe.totals()
    red = torch.tensor([elapsed, float(tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"] - tot["rays_shadow_skipped"]),
                        float(tot["rays_primary"])], dtype=torch.float64, device=dev)
    if world > 1:
        mx = red.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
    if rank == 0:
        slots, floats = capi.tile_gather_layout(W, H, band, world)
        print(json.dumps({
            "metric": "Mrays/s (all traced rays), 10 M triangles 3840x2160 4-bounce progressive, tile-partitioned",
            "value": float(red[1].item()) / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": K, "warmup": Wu,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: displaced-grid mesh (seed 7, %d triangles), %dx%d, 4 radiance bounces, 1 spp/frame, "
                                   "interleaved %d-row bands per GPU" % (tri.shape[0], W, H, band),
                       "parallelism": "tile-partitioned x%d, one all-gather of %d band slots per rank (%.1f MB sent per rank)"
                                      % (world, slots, floats * 4 / 1e6) if world > 1 else "single GPU",
                       "frames": K, "frames_per_launch_set": S,
                       "entry_point": "rt_pipeline_render_bands_batch" if S > 1 else "rt_pipeline_update + rt_pipeline_render_bands"},
            "frames_per_s": K / elapsed, "primary_mrays_per_s": float(red[2].item()) / elapsed / 1e6, "bvh_build_ms": scene.build_ms(),
            "ranks": report}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


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
    # memory-side traffic for the two rooflines, measured by child runs under rocprofv3 BEFORE this process touches the GPU
    live = {}
    if world == 1 and not args.no_roofline and not args.no_live_pmc and args.partition == "samples" and args.workload == "c2":
        t_live = time.perf_counter()
        n_sets_ = (args.steps + max(1, min(args.batch, 32)) - 1) // max(1, min(args.batch, 32))
        live["c2"] = live_traffic("c2", args.width, args.height, budget_s=100.0, per_set=(args.steps + n_sets_ - 1) // n_sets_)
        if args.hbm_frames > 0:           # (the 10 M-triangle child runs take ~30 s each: the two traffic passes only)
            live["c5"] = live_traffic("c5", args.width, args.height, budget_s=max(20.0, 190.0 - (time.perf_counter() - t_live)), passes=("ea", "write"),
                                      per_set=HBM_SET if args.batch > 1 else 1)

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

    if args.partition == "tiles":
        return tiles_main(args, rank, world, local_rank, dev, capi, D, T, scenes, torch, dist)

    if args.workload == "c5":            # profiling passes: only the HBM-bound workload, one JSON line of its own
        assert world == 1, "--workload c5 is a single-GPU profiling mode"
        ctx = capi.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
        h = hbm_workload(ctx, capi, T, scenes, max(args.hbm_frames, 1), 2, batch=args.batch if args.batch_given else 1, one_frame_walk=not args.no_roofline)
        print(json.dumps({"metric": "Mrays/s, 10 M triangles 4K 4-bounce (roofline workload)", "value": h["Mrays_per_s"], "unit": "Mrays/s",
                          "n_gpus": 1, "roofline_hbm": h}))
        return

    W, H, K, Wu = args.width, args.height, args.steps, args.warmup
    verts, tris, workload, cam = headline_workload(args, capi, scenes, np)
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
    cam11 = capi.camera_array(cam["eye"], cam["at"], cam["up"], cam["fov"], W / H)
    pfcs = [host.update(cam11, 0.0, f + 1, W, H) for f in range(total_frames)]
    mine = D.shard_frames(rank, world, total_frames)
    pipe.set_accumulation_mode(T.ACCUM_SUM if world > 1 else T.ACCUM_RUNNING_MEAN)

    # frames per set of launches: the K timed frames go through ceil(K / batch) sets of (almost) equal size
    n_sets = (K + max(1, min(args.batch, 32)) - 1) // max(1, min(args.batch, 32))
    S = (K + n_sets - 1) // n_sets

    def step(i):
        pipe.update(pfcs[mine[i]])
        pipe.render()

    host_ms = {}                # the host's side of the last steps() call: time inside the update() + render() calls, the slowest one, the flush

    def steps(lo, hi, per_set=None):
        """frames lo..hi-1 of this rank, one update() + render() each: rendered at once (per_set 1), or recorded by the deferred
        pipeline and rendered in sets of per_set frames; the last, possibly partial, set is flushed before this returns"""
        per_set = S if per_set is None else per_set
        pipe.set_deferred(per_set if per_set > 1 else 0)
        ta = time.perf_counter()
        slowest = 0.0
        for i in range(lo, hi):
            t1 = time.perf_counter()
            step(i)
            slowest = max(slowest, time.perf_counter() - t1)
        tb = time.perf_counter()
        pipe.flush()
        host_ms.update(calls=(tb - ta) * 1e3, slowest_call=slowest * 1e3, flush=(time.perf_counter() - tb) * 1e3)
        if os.environ.get("DXR_BENCH_TRACE"):          # where the host's time goes (diagnostic)
            tc = time.perf_counter()
            torch.cuda.synchronize()
            sys.stderr.write("[bench] frames %d..%d: update+render calls %.2f ms, flush %.2f ms, sync %.2f ms\n"
                             % (lo, hi, (tb - ta) * 1e3, (tc - tb) * 1e3, (time.perf_counter() - tc) * 1e3))

    def fixed_total(T_all):
        """BASELINE configs[2] as it is written -- "256 spp accumulated, sample batches sharded across 8 MI355X with RCCL accum-buffer
        reduce": T_all frames IN ALL.  Rank r renders the frames {f : f mod N == r} of ONE global sequence through update() + render()
        in deferred sets of up to 32, then ONE all-reduce of the SUM image closes the timed region: strong scaling (the work is fixed,
        a rank's share shrinks as N grows), the collective inside the time.  Every rank calls this (it is a collective)."""
        host_s = capi.ProgressiveHost(4321)
        host_s.options["maxIterations"] = max(1024, T_all + 1)
        pf = [host_s.update(cam11, 0.0, f + 1, W, H) for f in range(T_all)]
        mine_s, sets_s, S_s = fixed_total_plan(rank, world, T_all, args.batch)
        assert mine_s == D.shard_frames(rank, world, T_all)
        pipe.set_deferred(0)
        pipe.clear_output()
        pipe.set_accumulation_mode(T.ACCUM_SUM if world > 1 else T.ACCUM_RUNNING_MEAN)
        if S_s > 1:
            pipe.reserve_batch(S_s)
        pipe.reset_totals()
        evs = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0s = time.perf_counter()
        pipe.set_deferred(S_s if S_s > 1 else 0)
        for f in mine_s:
            pipe.update(pf[f])
            pipe.render()
        pipe.flush()
        evs[0].record()
        total = len(mine_s)
        if world > 1:
            _, total = D.reduce_accumulation(acc, len(mine_s))
        evs[1].record()
        torch.cuda.synchronize()
        mine_s_elapsed = time.perf_counter() - t0s
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0s
        pipe.set_deferred(0)
        ts = pipe.totals()
        r_ = torch.tensor([el, float(ts["rays_primary"] + ts["rays_secondary"] + ts["rays_shadow"] - ts["rays_shadow_skipped"]), float(len(mine_s)),
                           float(evs[0].elapsed_time(evs[1])), mine_s_elapsed], dtype=torch.float64, device=dev)
        if world > 1:
            mx_ = r_.clone()
            mn_ = r_.clone()
            dist.all_reduce(mx_, op=dist.ReduceOp.MAX)
            dist.all_reduce(mn_, op=dist.ReduceOp.MIN)
            dist.all_reduce(r_, op=dist.ReduceOp.SUM)
            el = float(mx_[0].item())
            coll, frames_max, rank_s = float(mx_[3].item()), int(mx_[2].item()), (float(mn_[4].item()), float(mx_[4].item()))
        else:
            coll, frames_max, rank_s = float(r_[3].item()), len(mine_s), (mine_s_elapsed, mine_s_elapsed)
        assert total == T_all and int(round(float(r_[2].item()))) == T_all, "the ranks' shards do not add up to the total"
        return {"metric": "Mrays/s (all traced rays), 1080p progressive, %d frames IN ALL sharded over the ranks + one all-reduce" % T_all,
                "value": float(r_[1].item()) / el / 1e6, "unit": "Mrays/s", "scaling": "strong", "n_gpus": world,
                "total_frames": T_all, "steps": T_all, "ms_per_step": el / T_all * 1e3, "elapsed_s": el, "frames_per_s": T_all / el,
                "frames_per_rank_max": frames_max, "frames_per_launch_set": S_s, "launch_sets_per_rank": sets_s,
                "collective_ms_max": coll, "collective_bytes": int(acc.numel() * 4) if world > 1 else 0,
                "rank_elapsed_s_min_max": list(rank_s),
                "accumulation": "sum + one all-reduce, mean = sum / %d" % T_all if world > 1 else "running mean",
                "note": "BASELINE configs[2] as written; the time includes the collective.  At N = 1 this is the headline's own work in sets of %d "
                        "(compare ms_per_step with repeat_of_timed_steps: both run late in the process)" % S_s}

    if S > 1:
        pipe.reserve_batch(S)           # the work memory of a set of S frames: sized outside the timed region, like the output
    if os.environ.get("DXR_BENCH_PREROLL"):       # experiment (profiles/r03/shadow_cache_coarse_level.txt): frames on ANOTHER pipeline first
        other = capi.Pipeline(ctx)
        other.set_scene(scene); other.add_material(mat); other.set_environment_cube(env)
        other.create_output(W, H)
        n_pre = int(os.environ["DXR_BENCH_PREROLL"])
        for f0 in range(0, n_pre, 20):
            other.render_batch([host.update(cam11, 0.0, 10000 + f, W, H) for f in range(f0, min(f0 + 20, n_pre))])
        torch.cuda.synchronize()
        other.close()
    steps(0, Wu)
    if world > 1:                       # warm the collective too
        dist.all_reduce(torch.zeros_like(acc))
    # ... and everything else the timed region calls for the first time in this process: on a box whose page cache is cold a first
    # call into a library can cost tens of milliseconds of disk reads (one driver-form run of round 4 read 7.7 ms per step with
    # normal per-kernel times; profiles/r04/preroll.txt)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))     # the device's own clock around the K steps
    ev[0].record(); ev[1].record(); span[0].record(); span[1].record()
    torch.cuda.synchronize()
    ev[0].elapsed_time(ev[1]); span[0].elapsed_time(span[1])
    if not args.no_roofline:
        pipe.enable_timing(K)
    pipe.reset_totals()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    span[0].record()
    steps(Wu, Wu + K)
    span[1].record()
    timed_host_ms = dict(host_ms)
    ev[0].record()
    if world > 1:
        mean, n_frames = D.reduce_accumulation(acc, Wu + K)
    ev[1].record()
    t_sync = time.perf_counter()
    torch.cuda.synchronize()
    mine_elapsed = time.perf_counter() - t0
    timed_host_ms["wait_for_the_device"] = (time.perf_counter() - t_sync) * 1e3
    timed_host_ms["device_span_of_the_steps"] = span[0].elapsed_time(span[1])       # first to last command of the K steps, by the device's events
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    report = rank_report(torch, dist, ctx, rank, world, local_rank, dev, backend, mine_elapsed, ev[0].elapsed_time(ev[1]))

    tot = pipe.totals()
    # `value` counts rays that were actually TRAVERSED: every ray the reference's shaders trace.  (rt_pipeline_set_skip_unlit_shadow_rays,
    # off here as by default, would leave out the shadow rays of lights with N.L == 0 and count them in rays_shadow_skipped.)
    rays_local = tot["rays_primary"] + tot["rays_secondary"] + tot["rays_shadow"] - tot["rays_shadow_skipped"]
    red = torch.tensor([elapsed, float(rays_local), float(tot["rays_primary"]), float(tot["rays_shadow_skipped"])], dtype=torch.float64, device=dev)
    if world > 1:
        mx = red.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(red, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0].item())
    rays_all, primary_all, skipped_all = float(red[1].item()), float(red[2].item()), float(red[3].item())

    out = None
    if rank == 0:
        out = {
            # (the mode is part of the metric's name; `value_frame_by_frame` below is the same frames, one set of launches each)
            "metric": "Mrays/s (all traced rays: primary + secondary + shadow), 1080p 1 spp/frame progressive" +
                      (", rendered in deferred sets of %d frames" % S if S > 1 else ", one set of launches per frame"),
            "value": rays_all / elapsed / 1e6, "unit": "Mrays/s",
            "n_gpus": world, "steps": K, "warmup": Wu, "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s, %dx%d, 1 spp/frame progressive accumulation, reference default material/lights/options" % (workload, W, H),
                       "camera": {"eye": list(cam["eye"]), "at": list(cam["at"]), "vfov": cam["fov"]},
                       "frames_per_gpu": K, "parallelism": "sample-sharded x%d, one RCCL all-reduce of the fp32 accumulation buffer" % world
                       if world > 1 else "single GPU", "accumulation": "sum+allreduce" if world > 1 else "running mean",
                       "frames_per_launch_set": S, "launch_sets": n_sets,
                       # (rt_pipeline_set_shadow_cache, automatic: the occluder a shadow ray's light-space cell met last is tested first;
                       #  every ray is still answered exactly and counted)
                       "shadow_cache_cells_per_side": pipe.shadow_cache(),
                       "entry_point": "rt_pipeline_update + rt_pipeline_render per frame" + (" (deferred mode: rt_pipeline_set_deferred(%d) renders the "
                                      "recorded frames through shared sets of launches, the same image bit for bit)" % S if S > 1 else ""),
                       "queue_memory_bytes_per_frame": pipe.queue_memory()[0] / max(S, 1),
                       # (the traversal kernels' global stack rows behind the LDS rows, never touched by this scene's rays.  Round 5: rows for the
                       #  resident threads of the persistent launches only -- one allocation per context whatever the set's size; the
                       #  one-tile-per-wave primary launch keeps none (k_primary_retry).  Rounds 1 - 4: 224 MB per frame of a set.)
                       "global_stack_rows_bytes_per_frame": ctx.stack_memory() / max(S, 1),
                       "global_stack_rows_bytes_per_context": ctx.stack_memory()},
            "primary_mrays_per_s": primary_all / elapsed / 1e6,
            "frames_per_s": K * world / elapsed,
            "rays_per_frame": rays_all / (K * world),
            # every ray the reference's shaders trace for these frames (the unlit shadow rays included) per second of this run
            "reference_ray_budget_mrays_per_s": (rays_all + skipped_all) / elapsed / 1e6,
            "shadow_rays_not_traversed_per_frame": skipped_all / (K * world),
            "bvh_build_ms": build_ms, "bvh_rebuild_ms": rebuild_ms,
            # the host's side of the timed region: time inside the K update() + render() calls (deferred mode: the last one enqueues the
            # set), the slowest single call, the explicit flush, then the wait for the device.  Kernel time is in `stages`.
            "timed_region_host_ms": timed_host_ms,
            "ranks": report,
        }
        if not args.no_roofline:
            if S > 1:
                pipe.set_deferred(0)
                step(Wu + K - 1)        # (the walk counters below are those of ONE frame: the last one again, by itself, after the timed region)
            stages, n_t = stage_table(pipe, tot)
            dom = max(TRACE_STAGES, key=lambda s: stages[s]["avg_ms"])
            n_nodes, n_recs = scene.wide_counts(0)        # the one instance's BLAS: the traversal arrays of the headline scene
            out["roofline"] = headline_roofline(args, stages, dom, n_t, n_sets, S, live, W, H, out["ms_per_step"], n_nodes * NODE_LINE_BYTES + n_recs * TRIREC_BYTES)
            out["stages"] = stages
        if world == 1 and not args.no_sample_batches and S > 1:
            # the same K frames again, one update() + render() per frame as the reference's app loop issues them
            # (DXRExperimentsApp::OnUpdate / OnRender): the same image bit for bit, every persistent traversal launch and its tail
            # paid per frame instead of per set
            pipe.clear_output()
            steps(0, Wu, per_set=1)
            ctx.synchronize()
            pipe.enable_timing(K)
            pipe.reset_totals()
            tb0 = time.perf_counter()
            steps(Wu, Wu + K, per_set=1)
            ctx.synchronize()
            tb = time.perf_counter() - tb0
            totb = pipe.totals()
            raysb = totb["rays_primary"] + totb["rays_secondary"] + totb["rays_shadow"] - totb["rays_shadow_skipped"]
            fb_stages = {name: sum(totb[k] for k in keys) / max(int(totb["frames"]), 1) for name, (keys, _, _) in TRACE_STAGES.items()}
            # top level, beside `value`: what an application that presents every frame gets (the reference's loop, src/DXRExperimentsApp.cpp:162-165,194)
            out["value_frame_by_frame"] = raysb / tb / 1e6
            out["ms_per_step_frame_by_frame"] = tb / K * 1e3
            out["frame_by_frame"] = {"frames": K, "ms_per_frame": tb / K * 1e3, "Mrays_per_s": raysb / tb / 1e6, "frames_per_s": K / tb,
                                     "stage_ms": fb_stages, "value_over_frame_by_frame": (tb / K) / (elapsed / K),
                                     "note": "the same update() + render() calls with deferred mode off (rt_pipeline_set_deferred(0)): every frame is rendered "
                                             "by its own set of launches before render() returns -- what an application that presents every frame gets; "
                                             "same frames, same bits; `value` above renders them %d per set of launches" % S}
        if world == 1 and not args.no_sample_batches:
            # the timed steps ONCE MORE, now that the process has rendered for a while: how much of `value` is the state the chip and
            # the shadow cache are in when the K steps start right after W = 5 warm-up frames (say the spread in the line)
            pipe.clear_output()
            steps(0, Wu)
            ctx.synchronize()
            pipe.reset_totals()
            tr0 = time.perf_counter()
            steps(Wu, Wu + K)
            ctx.synchronize()
            tr = time.perf_counter() - tr0
            totr = pipe.totals()
            raysr = totr["rays_primary"] + totr["rays_secondary"] + totr["rays_shadow"] - totr["rays_shadow_skipped"]
            out["repeat_of_timed_steps"] = {"ms_per_step": tr / K * 1e3, "Mrays_per_s": raysr / tr / 1e6, "value_over_repeat": (rays_all / elapsed) / (raysr / tr),
                                            "note": "the same W warm-up and K timed steps again at the end of this run (same calls, same frames, same bits). "
                                                    "`value` is the FIRST pass; measured over rounds 3 - 4 it reads 3 - 7 % below this one (chip power state "
                                                    "after the idle set-up phase, a shadow cache still filling: profiles/r04/warmup_sensitivity.txt, preroll.txt) "
                                                    "and varies +-1.5 % from box to box"}
    # the fixed-total form of configs[2] (every rank: it ends in a collective); at N = 1 after the extras above, which replay the timed frames
    strong = None
    if not args.no_strong:
        strong = fixed_total(args.total_frames if args.total_frames else 256)
    if rank == 0:
        if strong is not None:
            if args.total_frames:           # --total-frames: the strong line IS the line, the K-frames-per-GPU run rides along
                weak = {k: out[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "scaling", "frames_per_s")}
                out["weak_scaling"] = weak
                for k in ("metric", "value", "steps", "ms_per_step", "scaling", "frames_per_s"):
                    out[k] = strong[k]
                out["config"]["total_frames"] = strong["total_frames"]
                out["config"]["parallelism"] = ("sample-sharded x%d: %d frames in all, ceil(T / N) per rank in deferred sets, one RCCL all-reduce of the fp32 "
                                                "accumulation buffer inside the timed region" % (world, strong["total_frames"])) if world > 1 else "single GPU"
                out["strong_scaling"] = {k: v for k, v in strong.items() if k not in ("metric", "value", "unit", "steps", "ms_per_step", "scaling", "frames_per_s")}
            else:
                out["strong_scaling"] = strong
        if world == 1 and args.hbm_frames > 0 and not args.no_roofline:
            del pipe, scene, model
            hb = HBM_SET if S > 1 else 1                       # like the headline: sets of frames
            h = hbm_workload(ctx, capi, T, scenes, args.hbm_frames, 2, batch=hb)
            dom = max(TRACE_STAGES, key=lambda s: h["stages"][s]["avg_ms"])
            d = h["stages"][dom]
            prof5 = committed_profile("c5b" if hb > 1 else "c5")
            k5 = prof5.get("kernels", {}).get(d["kernel"], {})
            lv5 = live.get("c5", {}).get(d["kernel"], {})
            fpl5 = float(hb) if lv5 else float(k5.get("frames_per_launch", 1))         # frames per launch in the counter passes
            traffic = lv5.get("bytes_per_launch", k5.get("bytes_per_launch"))
            traffic = traffic / fpl5 if traffic else traffic                            # ... per frame, like the stage time
            prof_us = k5.get("avg_us")
            live_ms = d["avg_ms"] / d["launches_per_frame"]          # per frame; the stage time covers one launch per radiance level
            h["roofline"] = {"bound": "hbm", "kernel": d["kernel"], "stage": dom, "launches_per_frame": d["launches_per_frame"],
                             # memory-side bytes per launch (PMC passes of THIS workload, profiles/<round>/traffic.json: read requests
                             # by size + WRITE_SIZE) / this run's HIP-event duration of the same launches; the profiling session's own
                             # duration and rate alongside.  The ~0.8 GB of nodes + triangle records do not fit the 256 MB Infinity Cache.
                             "achieved": traffic / (live_ms * 1e-3) / 1e9 if traffic else None,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": traffic / (live_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None,
                             "traffic": traffic, "avg_launch_ms": live_ms * h["frames_per_launch_set"], "avg_ms_per_frame": live_ms,
                             "frames_per_launch": h["frames_per_launch_set"], "frames_timed": h["launches_timed"],
                             "traffic_source": ("this run: %d launches under rocprofv3 --pmc (read requests by size, WRITE_SIZE)" % lv5["dispatches"])
                                               if "bytes_per_launch" in lv5 else "committed profile",
                             "traffic_committed_profile": k5.get("bytes_per_launch"),
                             "profiled_avg_us": prof_us, "profiled_GBps": k5.get("GBps"),
                             "gathered_bytes_per_frame": d["gathered_bytes"], "gathered_GBps": d["gathered_GBps"],
                             "requested_bytes_per_frame": d["requested_bytes"],
                             "pmc": {k: k5.get(k) for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum",
                                                            "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
                                                            "SQ_INSTS_VALU") if k in k5},
                             "pmc_source": prof5.get("source"), "pmc_commit": prof5.get("commit")}
            out["roofline_hbm"] = h
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(verts, tris, mat, env, pfcs[mine[Wu]], W, H, args.cpu_seconds)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
