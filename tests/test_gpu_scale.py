"""BASELINE configs 4 and 5 at their full sizes (parity-test cases, not bench lines):
C5  synthetic 10 M-triangle mesh, 3840x2160, 4 bounces -> BVH index-exact, hits bit-exact on a ray sample, a window of the 4K frame bit-exact
C4  4096 instances of two meshes, one of them the reference's susanne.obj (TLAS over many BLAS instances), 3840x2160,
    realtime pipeline + denoiser -> both AOVs and the composite of the whole 4K frame bit-exact against the oracle."""
import time

import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import ANY, CULL, Pair, assert_hits_equal, cam_array, nodes_equal, random_rays

pytestmark = pytest.mark.gpu


def test_c5_ten_million_triangles(gpu, oracle, capi):
    t0 = time.time()
    v, tri = scenes.displaced_grid(2236, seed=7)                 # 9,999,392 triangles
    assert tri.shape[0] > 9_900_000
    t1 = time.time()
    p = Pair(oracle, capi, gpu, [(v, tri)], [(0, None)])
    t2 = time.time()
    gn, gk, gp, gd = p.g.bvh(0)
    on, ok, op, od = p.o.bvh(0)
    assert np.array_equal(gk, ok) and nodes_equal(gn, on) and np.array_equal(gp, op) and gd == od
    del gn, gk, gp, on, ok, op
    O, D = random_rays(200000, 21, [-20, -12, -20], [20, 4, 20])
    for flags in (0, CULL):
        assert_hits_equal(p.g.trace(O, D, flags=flags), p.o.trace(O, D, flags=flags, mode=1, nthreads=8), "10M fast vs oracle flags=%d" % flags)
    assert_hits_equal(p.g.trace(O, D, flags=ANY), p.o.trace(O, D, flags=ANY, mode=1, nthreads=8), "10M any-hit", closest=False)
    gc = p.g.trace(O[:20000], D[:20000], flags=0, canonical=True)
    ob = p.o.trace(O[:20000], D[:20000], flags=0, mode=1, nthreads=8)
    assert np.array_equal(gc["nodes"], ob["nodes"]) and np.array_equal(gc["tris"], ob["tris"])
    # 4K progressive frames as a 4-bounce path trace: glossy surface, specular chains up to radiance depth 4
    W, H = 3840, 2160
    pipe = capi.Pipeline(gpu)
    pipe.set_scene(p.g)
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
    cam = cam_array(dict(eye=(0.0, 6.0, 19.0), at=(0.0, -4.0, 0.0), up=(0, 1, 0), fov=0.8), W / H)
    pipe.enable_timing(2)
    for f in range(2):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        pipe.update(pfc)
        pipe.render()
    img = pipe.read_output()
    st = pipe.stats()
    assert np.isfinite(img).all() and img[..., 3].min() == 1.0
    assert st["rays_primary"] == W * H and st["primary_hits"] > W * H // 4
    assert st["secondary_hits"] > st["primary_hits"]
    # a 96x32 window of a single 4K frame against the oracle, bit for bit
    tile = (1900, 1300, 1996, 1332)
    pfc["cameraParams"]["accumCount"] = 0
    pipe.clear_output()
    pipe.update(pfc)
    pipe.render()
    one = pipe.read_output()
    ref, _ = p.o.render(mat, pfc, W, H, env_faces=scenes.sky_cubemap(32), tile=tile, max_radiance_depth=4, max_shadow_depth=2, nthreads=8)
    assert np.array_equal(one[tile[1]:tile[3], tile[0]:tile[2]], ref[tile[1]:tile[3], tile[0]:tile[2]])
    # (round 4) ... and the WHOLE 3840 x 2160 frame, four bounces -- about 55 M rays through the oracle -- with its ray counts.
    # Round 5: on every box, whatever cores it grants (16 on the driver's boxes: ~10 s; fewer cores take longer, they do not
    # shrink the comparison to the window above without saying so)
    import os
    cores = len(os.sched_getaffinity(0))
    quarter = (0, 0, W, H)
    ref, ost = p.o.render(mat, pfc, W, H, env_faces=scenes.sky_cubemap(32), max_radiance_depth=4, max_shadow_depth=2, nthreads=cores)
    assert np.array_equal(one, ref), "%d pixels of the 4K frame differ" % int((one != ref).any(axis=2).sum())
    pipe.clear_output()
    pipe.render(tile=quarter)
    gst = pipe.stats()
    for key in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
        assert gst[key] == ost[key], (key, gst[key], ost[key])
    rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
    print("\nC5: generate %.1fs, build both %.1fs (GPU BVH build %.1f ms), 4K frame %.2f ms = %.0f Mrays/s" % (
        t1 - t0, t2 - t1, p.g.build_ms(), st["ms_total"], rays / st["ms_total"] / 1e3))


def _multi_mesh_fbx(path, verts, tris, parts, shift):
    """Writes (verts, tris) as a binary FBX of `parts` Geometry nodes (tests/fbx_tools.py's writer), the triangles dealt to them
    in runs, every mesh but the first under a Model with an Lcl Translation of k * shift: what the reference's importer
    concatenates and pre-transforms into ONE RtModel (libs/DXRFramework/RtModel.cpp:26, :36-56)."""
    import fbx_tools as F
    meshes = []
    per = (tris.shape[0] + parts - 1) // parts
    for k in range(parts):
        t = tris[k * per:(k + 1) * per]
        m = dict(positions=verts["position"].astype(np.float64), polygons=[list(map(int, x)) for x in t],
                 normals=verts["normal"][t.reshape(-1)].astype(np.float64), mapping="ByPolygonVertex")
        if k:
            m["translation"] = tuple(float(k) * np.asarray(shift, np.float64))
        meshes.append(m)
    F.write(path, meshes, version=7500 if parts % 2 else 7400, compress=bool(parts % 2))
    return F.ingest(path)


def test_c4_many_instances_realtime_denoise_4k(gpu, oracle, capi, tmp_path):
    """BASELINE config 4 at its stated size and in its stated form ("FBX multi-mesh scene"): a TLAS over 4096 instances of TWO
    distinct BLASes plus a ground plane, every BLAS ingested by the product's FBX reader through rt_model_create_from_file
    (round 5) -- the reference's own assets/models/susanne.obj mesh written as a TWO-mesh binary FBX, a second mesh as a THREE-mesh
    one (32-bit records, uncompressed arrays), and the reference's own ground.fbx (the committed re-emission) -- at 3840x2160
    through RealtimeRaytracingPipeline (src/RealtimeRaytracingPipeline.cpp:201-235) and DenoiseCompositor
    (src/DenoiseCompositor.cpp:109-148).  Both AOVs and the denoised composite of the WHOLE 4K frame are compared with the
    oracle, bit for bit."""
    import os
    from util import GOLDEN
    sus = oracle.obj_load(os.path.join(GOLDEN, "susanne.obj"))
    assert sus[1].shape[0] == 968
    blob = scenes.blob_mesh(level=3)                              # 1280 triangles
    fa, fb, fg = str(tmp_path / "susanne_two_meshes.fbx"), str(tmp_path / "blob_three_meshes.fbx"), os.path.join(GOLDEN, "ground.fbx")
    ia = _multi_mesh_fbx(fa, sus[0], sus[1], 2, (0.0, 0.35, 0.0))
    ib = _multi_mesh_fbx(fb, blob[0], blob[1], 3, (0.15, 0.0, 0.1))
    xf = scenes.instance_grid(64, spacing=3.0)                    # 4096 instances
    inst = [(k % 2, xf[k]) for k in range(xf.shape[0])]
    ground = np.array([0.3, 0, 0, 0, 0, 0.3, 0, -2.5, 0, 0, 0.3, 0], np.float32)      # the 400 x 400 plane scaled under the grid
    inst.append((2, ground))
    p = Pair(oracle, capi, gpu, [fa, fb, fg], inst)
    # what the product's FBX reader handed over is what the independent parser reads from the same files
    for gm, (iv, ii) in zip(p.gmodels, (ia, ib, (None, None))):
        gv, gi = gm.geometry()
        if iv is not None:
            assert np.array_equal(np.concatenate([gv["position"], gv["normal"]], 1), iv) and np.array_equal(gi, ii)
    assert p.gmodels[0].geometry()[1].shape[0] == 968 and p.gmodels[1].geometry()[1].shape[0] == 1280 and p.gmodels[2].geometry()[1].shape[0] == 800
    gn, gk, gp, gd = p.g.bvh(-1)
    on, ok, op, od = p.o.bvh(-1)
    assert np.array_equal(gk, ok) and nodes_equal(gn, on) and np.array_equal(gp, op) and gd == od
    for which, model in ((0, 0), (1, 1), (4096, 2)):              # all three BLASes (instance 0 uses model 0, instance 1 model 1, the last the plane)
        gn, gk, gp, gd = p.g.bvh(which)
        on, ok, op, od = p.o.bvh(model)
        assert np.array_equal(gk, ok) and nodes_equal(gn, on) and np.array_equal(gp, op) and gd == od
    O, D = random_rays(200000, 22, [-100, -3, -100], [100, 3, 100])
    for flags in (0, CULL):
        assert_hits_equal(p.g.trace(O, D, flags=flags), p.o.trace(O, D, flags=flags, mode=1, nthreads=8), "instanced fast vs oracle flags=%d" % flags)
    assert_hits_equal(p.g.trace(O, D, flags=ANY), p.o.trace(O, D, flags=ANY, mode=1, nthreads=8), "instanced any-hit", closest=False)
    W, H = 3840, 2160
    pipe = capi.Pipeline(gpu, capi.PIPELINE_REALTIME)
    pipe.set_scene(p.g)
    r = np.random.default_rng(5)
    mats = []
    for k in range(len(inst)):
        m = T.default_material()
        m["albedo"][:3] = r.uniform(0.1, 0.9, 3)
        m["type"] = k % 3
        pipe.add_material(m)
        mats.append(m)
    env = scenes.sky_cubemap(32)
    pipe.set_environment_cube(env)
    pipe.create_output(W, H)
    pipe.build_acceleration_structures()
    host = capi.ProgressiveHost(4)
    cam = cam_array(dict(eye=(0.0, 30.0, 110.0), at=(0.0, 0.0, 0.0), up=(0, 1, 0), fov=0.9), W / H)
    pipe.enable_timing(2)
    dn = capi.Denoiser(gpu)
    dn.create_output(W, H)
    for f in range(2):
        pfc = host.update_realtime(cam, 0.0, f + 1, W, H)
        pipe.update(pfc)
        pipe.render()
        dn.dispatch(pipe.output_device_ptr(0), pipe.output_device_ptr(1))
    direct, indirect, out = pipe.read_output(0), pipe.read_output(1), dn.read_output()
    st = pipe.stats()
    assert np.isfinite(out).all() and 0 < st["primary_hits"] < W * H
    od, oi, ost = p.o.render_realtime(np.stack(mats), pfc, W, H, env_faces=env, nthreads=16)
    assert all(st[k] == ost[k] for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"))
    assert np.array_equal(direct, od), "direct-lighting AOV differs on %d of %d pixels" % (int((direct != od).any(axis=2).sum()), W * H)
    assert np.array_equal(indirect, oi), "indirect-specular AOV differs on %d pixels" % int((indirect != oi).any(axis=2).sum())
    prm = np.frombuffer(dn.params[0].tobytes(), oracle.DENOISE_PARAMS)[0]
    _, ocomp = oracle.denoise(od, oi, prm, nthreads=16)
    assert np.array_equal(out, ocomp), "denoised composite differs on %d pixels" % int((out != ocomp).any(axis=2).sum())
    rays = st["rays_primary"] + st["rays_secondary"] + st["rays_shadow"]
    print("\nC4: 4096 instances of 2 BLASes, 4K realtime frame %.2f ms = %.0f Mrays/s, denoise %.3f ms" % (
        st["ms_total"], rays / st["ms_total"] / 1e3, dn.last_ms()))


def test_bench_two_ranks_on_one_gpu():
    """bench.py's N > 1 path end to end (rendezvous, frame sharding, SUM accumulation, all-reduce, max-over-ranks
    timing, rank-0 JSON) with two ranks sharing this box's single GPU: gloo stands in for RCCL, which refuses
    two ranks per device.  The 8-GPU run itself is the driver's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DXR_BENCH_BACKEND="gloo", DXR_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29871", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--width", "480", "--height", "270", "--cpu-seconds", "0"]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["frames_per_gpu"] == 4
    # the N > 1 line says what really ran: both ranks seen, their backend, devices and PCI ids (ONE device here: that is the
    # point of this test and exactly what the line must expose), a measured collective per rank, the elapsed spread
    rk = d["ranks"]
    assert rk["ranks_seen"] == 2 and rk["backend"] == "gloo" and rk["device_ordinals"] == [0, 0]
    assert len(rk["pci_bus_ids"]) == 2 and rk["pci_bus_ids"][0] == rk["pci_bus_ids"][1] and rk["distinct_devices"] == 1
    assert len(set(rk["pids"])) == 2 and all(ms > 0.0 for ms in rk["collective_ms"])
    assert 0.0 < rk["elapsed_s_min"] <= rk["elapsed_s_max"]
    # (round 5) ... and carries BASELINE configs[2] as written beside the weak line: 256 frames IN ALL over the two ranks, one all-reduce
    sg = d["strong_scaling"]
    assert sg["scaling"] == "strong" and sg["n_gpus"] == 2 and sg["total_frames"] == 256 and sg["frames_per_rank_max"] == 128
    assert sg["frames_per_launch_set"] == 32 and sg["launch_sets_per_rank"] == 4 and sg["value"] > 0 and sg["collective_ms_max"] > 0
    assert sg["collective_bytes"] == 480 * 270 * 16
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--width", "480", "--height", "270",
                          "--cpu-seconds", "0", "--no-live-pmc", "--hbm-frames", "0"], cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # weak scaling: two ranks trace twice the frames, i.e. about twice the rays per step
    assert abs(d["value"] * d["ms_per_step"] / (d1["value"] * d1["ms_per_step"]) - 2.0) < 0.05
    # the single-GPU line carries the contract's objects: the reference's per-frame calls as the entry point (rendered in sets by the
    # deferred pipeline), the same calls with deferred mode off beside it, and a roofline whose fraction is achieved / peak of the same unit
    assert d1["config"]["frames_per_launch_set"] == 4 and d1["config"]["launch_sets"] == 1
    assert d1["config"]["entry_point"].startswith("rt_pipeline_update + rt_pipeline_render per frame") and "rt_pipeline_set_deferred(4)" in d1["config"]["entry_point"]
    assert d1["config"]["queue_memory_bytes_per_frame"] > 0 and d1["config"]["global_stack_rows_bytes_per_frame"] >= 0
    fb = d1["frame_by_frame"]
    assert fb["frames"] == 4 and fb["ms_per_frame"] > 0 and fb["Mrays_per_s"] > 0
    # the host's side of the timed region and the same steps once more at the end of the run (the line states its own spread)
    hm = d1["timed_region_host_ms"]
    assert all(hm[k] >= 0.0 for k in ("calls", "slowest_call", "flush", "wait_for_the_device", "device_span_of_the_steps"))
    assert hm["device_span_of_the_steps"] <= d1["ms_per_step"] * d1["steps"] * 1.05 + 0.5
    rp = d1["repeat_of_timed_steps"]
    assert rp["ms_per_step"] > 0 and 0.2 < rp["value_over_repeat"] < 5.0
    rl = d1["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "frames_per_launch", "memory_path", "algorithmic_bytes",
              "compulsory_bytes", "lane_utilisation"):
        assert k in rl, k
    assert rl["bound"] == "hbm" and rl["unit"] == "GB/s" and rl["peak"] == 8000.0
    assert 0.2 < rl["lane_utilisation"]["node_steps"] <= 1.0 and 0.1 < rl["lane_utilisation"]["triangle_steps"] <= 1.0
    assert rl["frac"] is None          # (no live counters asked for and not the default workload: the committed profile does not apply)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in d1 and key in d, key
    assert json.loads(json.dumps(d1)) == d1          # (no NaN / Infinity in the line)
    s1 = d1["strong_scaling"]
    assert s1["scaling"] == "strong" and s1["n_gpus"] == 1 and s1["total_frames"] == 256 and s1["collective_bytes"] == 0
    # the fixed-total form as the line itself: --total-frames T
    st = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "1", "--width", "480", "--height", "270", "--total-frames", "48",
                         "--cpu-seconds", "0", "--no-live-pmc", "--hbm-frames", "0", "--no-roofline", "--no-frame-by-frame"], cwd=root, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, text=True, timeout=600)
    assert st.returncode == 0, st.stderr[-3000:]
    d2 = json.loads([l for l in st.stdout.splitlines() if l.startswith("{")][-1])
    assert d2["scaling"] == "strong" and d2["steps"] == 48 and d2["config"]["total_frames"] == 48 and d2["weak_scaling"]["scaling"] == "weak" and d2["weak_scaling"]["steps"] == 4
    assert abs(d2["value"] * d2["ms_per_step"] * 48 / (d2["weak_scaling"]["value"] * d2["weak_scaling"]["ms_per_step"] * 4) - 12.0) < 0.5      # rays of 48 frames : rays of 4


def test_tile_partition_two_ranks_on_one_gpu():
    """BASELINE config 5's multi-GPU scheme end to end: interleaved row bands per rank, one all-reduce as the gather,
    bit-identical to the whole frame (tests/scripts/tile_ranks.py; gloo + one device stand in for RCCL + 8 GPUs)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DXR_BENCH_BACKEND="gloo", DXR_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29873", os.path.join(root, "tests", "scripts", "tile_ranks.py")]
    r = subprocess.run(cmd, env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "tiles ok: 2 ranks" in r.stdout, r.stdout[-3000:]
