"""The wavefront progressive pipeline on the GPU == the oracle's recursive restatement.

Bar: every pixel of the fp32 accumulation buffer equal (the engine's arithmetic is
FMA-free with engine-defined transcendentals, so equality is exact, which is far
inside north_star's 1e-5 RMS).  Full-size (1080p) cases use size-independent
properties: tiles == whole frame, sharded sums == single run, reruns identical."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import CORNELL_OBJ, GOLDEN, cam_array, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def make_gpu_pipeline(capi, ctx, models, instances, mats, W, H, env=None, env_const=(0.5, 0.5, 0.5)):
    sc = capi.Scene(ctx)
    gm = [capi.Model(ctx, v, i) for v, i in models]
    for mi, x in instances:
        sc.add_model(gm[mi], x)
    p = capi.Pipeline(ctx)
    p.set_scene(sc)
    for m in mats:
        p.add_material(m)
    if env is not None:
        p.set_environment_cube(env)
    else:
        p.set_environment_constant(env_const)
    p.create_output(W, H)
    p.build_acceleration_structures()
    return p


def make_oracle_scene(oracle, models, instances):
    sc = oracle.Scene()
    for v, i in models:
        sc.add_model(v, i)
    for mi, x in instances:
        sc.add_instance(mi, x)
    sc.build()
    return sc


def rms(a, b):
    return float(np.sqrt(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)))


def test_cornell_golden_frames(gpu, capi):
    """Config C1 against the committed oracle fixture (no oracle needed at run time)."""
    g = np.load(GOLDEN + "/cornell64_golden.npz")
    m = capi.Model(gpu, path=CORNELL_OBJ)
    sc = capi.Scene(gpu)
    sc.add_model(m)
    p = capi.Pipeline(gpu)
    assert p.name == "Progressive Ray Tracing Pipeline"
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.set_environment_constant((0.5, 0.5, 0.5))
    p.create_output(64, 64)
    p.build_acceleration_structures()
    for f in range(4):
        p.update(g["pfc"][f])
        p.render()
        img = p.read_output()
        assert rms(img, g["images"][f]) <= 1e-5
        assert np.array_equal(img, g["images"][f]), "frame %d differs from the golden image" % f
        if f == 0:
            t, prim, inst = p.primary_hits(64 * 64)
            assert np.array_equal(prim, g["prim"]) and np.array_equal(inst, g["inst"]) and np.array_equal(t, g["t"])
            st = p.stats()
            assert st["rays_primary"] == 64 * 64 and st["primary_hits"] == int((g["inst"] != T.RT_NO_HIT).sum())


OPTION_CASES = [
    {},
    {"cosineHemisphereSampling": 0},
    {"debug": 2},
    {"noIndirectDiffuse": 1},
    {"showAmbientOcclusionOnly": 1},
    {"showAmbientOcclusionOnly": 1, "cosineHemisphereSampling": 0},
    {"showDirectLightingOnly": 1},
    {"showIndirectDiffuseOnly": 1},
    {"showIndirectSpecularOnly": 1},
    {"showFresnelTerm": 1},
    {"showGBufferAlbedoOnly": 1},
    {"environmentStrength": 2.5},
]


@pytest.mark.parametrize("opts", OPTION_CASES, ids=lambda o: "-".join("%s=%s" % kv for kv in o.items()) or "defaults")
def test_cornell_options_vs_oracle(gpu, capi, oracle, opts):
    W, H = 96, 80
    v, i = oracle.obj_load(CORNELL_OBJ)
    models, inst = [(v, i)], [(0, None)]
    env = scenes.sky_cubemap(16)
    mat = T.default_material()
    p = make_gpu_pipeline(capi, gpu, models, inst, [mat], W, H, env=env)
    osc = make_oracle_scene(oracle, models, inst)
    host = capi.ProgressiveHost(7)
    for k, val in opts.items():
        host.options[k] = val
    cam = cam_array(scenes.cornell_camera(), W / H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        p.update(pfc)
        p.render()
        acc, ost = osc.render(mat, pfc, W, H, accum=acc, env_faces=env, nthreads=8)
        img = p.read_output()
        assert np.array_equal(img, acc), "opts %s frame %d: %d pixels differ, rms %g" % (
            opts, f, int((img != acc).any(axis=2).sum()), rms(img, acc))
        gst = p.stats()
        for key in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
            assert gst[key] == ost[key], key


@pytest.mark.parametrize("mtype,refl", [(0, 0.7), (2, 0.4), (1, 0.0)])
def test_material_types_and_depth_limits(gpu, capi, oracle, mtype, refl):
    W, H = 64, 48
    v, i = oracle.obj_load(CORNELL_OBJ)
    mat = T.default_material()
    mat["type"] = mtype
    mat["reflectivity"] = refl
    mat["emissive"] = (0.1, 0.2, 0.3, 0.5)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    host = capi.ProgressiveHost(3)
    cam = cam_array(scenes.cornell_camera(), W / H)
    # the reference compiles (1, 2) in (RaytracingCommon.hlsli:11-12); deeper specular chains are BASELINE config 5
    for (mr, ms) in ((1, 2), (0, 2), (1, 1), (1, 0), (2, 2), (4, 2), (3, 4), (4, 5), (2, 1)):
        p.set_depth_limits(mr, ms)
        p.clear_output()
        pfc = host.update(cam, 0.0, 5, W, H)
        pfc["cameraParams"]["accumCount"] = 0
        p.update(pfc)
        p.render()
        acc, ost = osc.render(mat, pfc, W, H, max_radiance_depth=mr, max_shadow_depth=ms, nthreads=8)
        assert np.array_equal(p.read_output(), acc), "depth limits (%d,%d)" % (mr, ms)
        gst = p.stats()
        for key in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
            assert gst[key] == ost[key], (key, mr, ms)
    with pytest.raises(capi.RtError):
        p.set_depth_limits(5, 2)


def test_instanced_scene_materials_and_misses(gpu, capi, oracle):
    W, H = 120, 68
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(200, seed=2, extent=1.5, size=0.5)
    xf = random_xforms(24, seed=5, spread=6.0)
    inst = [(k % 2, xf[k]) for k in range(24)] + [(0, None)]
    mats = []
    r = np.random.default_rng(1)
    for k in range(len(inst)):
        m = T.default_material()
        m["albedo"][:3] = r.uniform(0.1, 0.9, 3)
        m["roughness"] = r.uniform(0.2, 0.9)
        m["type"] = k % 3
        mats.append(m)
    env = scenes.sky_cubemap(8)
    p = make_gpu_pipeline(capi, gpu, [blob, soup], inst, mats, W, H, env=env)
    osc = make_oracle_scene(oracle, [blob, soup], inst)
    host = capi.ProgressiveHost(11)
    cam = cam_array(dict(eye=(0, 3, 16), at=(0, 0, 0), up=(0, 1, 0), fov=0.8), W / H)
    acc = np.zeros((H, W, 4), np.float32)
    omats = np.stack(mats)
    for f in range(2):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        p.update(pfc)
        p.render()
        acc, ost = osc.render(omats, pfc, W, H, accum=acc, env_faces=env, nthreads=8)
        assert np.array_equal(p.read_output(), acc)
    assert 0 < ost["primary_hits"] < W * H      # both hit and miss pixels were exercised
    # the same scene as a 4-bounce path trace (BASELINE config 5): mirrors between instances, two-level traversal
    for m in mats:
        m["type"] = 2
        m["reflectivity"] = 0.8
        m["roughness"] = 0.05
    for k, m in enumerate(mats):
        p.set_material(k, m)
    p.set_depth_limits(4, 3)
    p.clear_output()
    pfc = host.update(cam, 0.0, 9, W, H)
    pfc["cameraParams"]["accumCount"] = 0
    p.update(pfc)
    p.render()
    acc, ost = osc.render(np.stack(mats), pfc, W, H, env_faces=env, max_radiance_depth=4, max_shadow_depth=3, nthreads=8)
    assert np.array_equal(p.read_output(), acc)
    gst = p.stats()
    for key in ("rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
        assert gst[key] == ost[key], key
    assert ost["secondary_hits"] > ost["primary_hits"]      # chains really go past depth 1


def test_checkpoint_resume_is_bit_exact(gpu, capi, oracle, tmp_path):
    """8 frames straight == 4 frames, checkpoint, a NEW pipeline + host, load, 4 more frames."""
    W, H = 96, 64
    v, i = oracle.obj_load(CORNELL_OBJ)
    cam = cam_array(scenes.cornell_camera(), W / H)
    straight = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H)
    h0 = capi.ProgressiveHost(21)
    for f in range(8):
        straight.update(h0.update(cam, 0.0, f + 1, W, H))
        straight.render()
    first = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H)
    h1 = capi.ProgressiveHost(21)
    for f in range(4):
        first.update(h1.update(cam, 0.0, f + 1, W, H))
        first.render()
    path = str(tmp_path / "accum.ckpt")
    first.save_checkpoint(path, h1)
    second = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H)
    h2 = capi.ProgressiveHost(999)
    second.load_checkpoint(path, h2)
    assert np.array_equal(second.read_output(), first.read_output())
    for f in range(4, 8):
        second.update(h2.update(cam, 0.0, f + 1, W, H))
        second.render()
    assert np.array_equal(second.read_output(), straight.read_output())
    other = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W + 8, H)
    with pytest.raises(capi.RtError):
        other.load_checkpoint(path, h2)              # size mismatch
    with pytest.raises(capi.RtError):
        second.load_checkpoint(str(tmp_path / "missing.ckpt"), h2)
    img = np.random.default_rng(3).random((H, W, 4), dtype=np.float32)
    second.write_output(img)
    assert np.array_equal(second.read_output(), img)


def test_max_iterations_early_out_and_formats(gpu, capi, oracle):
    W, H = 32, 32
    v, i = oracle.obj_load(CORNELL_OBJ)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H)
    host = capi.ProgressiveHost(1)
    host.options["maxIterations"] = 2
    cam = cam_array(scenes.cornell_camera(), 1.0)
    imgs = []
    for f in range(4):
        p.update(host.update(cam, 0.0, f + 1, W, H))
        p.render()
        imgs.append(p.read_output())
    assert not np.array_equal(imgs[0], imgs[1])
    assert np.array_equal(imgs[1], imgs[2]) and np.array_equal(imgs[2], imgs[3])   # accumCount >= maxIterations: untouched
    p16 = capi.Pipeline(gpu)
    sc = capi.Scene(gpu)
    sc.add_model(capi.Model(gpu, v, i))
    p16.set_scene(sc)
    p16.add_material(T.default_material())
    p16.create_output(W, H, T.FORMAT_R16G16B16A16_FLOAT)
    p16.build_acceleration_structures()
    host2 = capi.ProgressiveHost(1)
    p16.update(host2.update(cam, 0.0, 1, W, H))
    p16.render()
    assert np.array_equal(p16.read_output(), imgs[0].astype(np.float16))


@pytest.mark.parametrize("rounding", (T.ROUND_NEAREST_EVEN, T.ROUND_TOWARD_ZERO))
@pytest.mark.parametrize("deferred", (0, 4))
def test_rgba16f_accumulation_storage(gpu, capi, oracle, rounding, deferred):
    """The reference's accumulation STORAGE (src/DXRExperimentsApp.cpp:28 -> src/ProgressiveRaytracingPipeline.cpp:127-131; read-modify-write at
    assets/shaders/ProgressiveRaytracing.hlsl:36-38): an RGBA16F texture, so every frame's running mean is rounded to fp16 before the next frame reads
    it.  rt_pipeline_set_accumulation_storage(RT_FORMAT_R16G16B16A16_FLOAT) does that; 8 Cornell frames equal the oracle's twin bit for bit, frame by
    frame and in deferred sets, and the fp16 read-out is then exact (every stored value IS an fp16 number)."""
    W, H = 64, 64
    v, i = oracle.obj_load(CORNELL_OBJ)
    mat = T.default_material()
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H)
    ref32 = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H)
    p.set_accumulation_storage(T.FORMAT_R16G16B16A16_FLOAT, rounding)
    p.set_deferred(deferred)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    host = capi.ProgressiveHost(5)
    cam = cam_array(scenes.cornell_camera(), 1.0)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(8):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        for q in (p, ref32):
            q.update(pfc)
            q.render()
        acc, _ = osc.render(mat, pfc, W, H, accum=acc, accum_f16=1 if rounding == T.ROUND_NEAREST_EVEN else 2, nthreads=4)
    img = p.read_output()
    assert np.array_equal(img, acc), "%d pixels differ" % int((img != acc).any(axis=2).sum())
    assert np.array_equal(img.astype(np.float16).astype(np.float32), img)              # every stored value is an fp16 number
    img32 = ref32.read_output()
    assert not np.array_equal(img, img32)
    # what the storage format costs: RMS against the fp32 accumulation over 8 frames (half an fp16 ulp per frame at most; stated in DESIGN.md section 2)
    assert rms(img[..., :3], img32[..., :3]) < (4e-4 if rounding == T.ROUND_NEAREST_EVEN else 2e-3)
    # back to fp32 storage: the next frame's mean is no longer rounded
    p.set_accumulation_storage(T.FORMAT_R32G32B32A32_FLOAT)
    pfc = host.update(cam, 0.0, 9, W, H)
    p.update(pfc)
    p.render()
    acc, _ = osc.render(mat, pfc, W, H, accum=acc, nthreads=4)
    assert np.array_equal(p.read_output(), acc)
    with pytest.raises(capi.RtError):
        p.set_accumulation_storage(1234)


def test_error_paths(gpu, capi):
    p = capi.Pipeline(gpu)
    with pytest.raises(capi.RtError):
        p.build_acceleration_structures()          # no scene
    with pytest.raises(capi.RtError):
        capi.Model(gpu, path="/nonexistent/file.obj")
    sc = capi.Scene(gpu)
    with pytest.raises(capi.RtError):
        sc.build()                                 # no instances
    with pytest.raises(capi.RtError):
        sc.trace(np.zeros((1, 4), np.float32), np.zeros((1, 4), np.float32))   # not built
    with pytest.raises(capi.RtError):
        capi.Context(9999)


@pytest.fixture(scope="module")
def sponza_pipeline(gpu, capi):
    v, i = scenes.sponza_class()
    W, H = 1920, 1080
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(64))
    return p, (v, i), W, H


def test_sponza_reduced_vs_oracle(gpu, capi, oracle):
    """Config C2's scene at 192x108 (oracle finishes in seconds): exact image equality."""
    W, H = 192, 108
    v, i = scenes.sponza_class()
    env = scenes.sky_cubemap(64)
    mat = T.default_material()
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H, env=env)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    host = capi.ProgressiveHost(1234)
    cam = cam_array(scenes.sponza_camera(), W / H)
    acc = np.zeros((H, W, 4), np.float32)
    for f in range(2):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        p.update(pfc)
        p.render()
        acc, ost = osc.render(mat, pfc, W, H, accum=acc, env_faces=env, nthreads=8)
        img = p.read_output()
        assert rms(img, acc) <= 1e-5
        assert np.array_equal(img, acc), "%d pixels differ" % int((img != acc).any(axis=2).sum())
    gst = p.stats()
    for key in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
        assert gst[key] == ost[key], key


def test_sponza_1080p_tiles_equal_whole_frame(sponza_pipeline, capi):
    """BASELINE config C2 at full size: rendering the frame as 4x3 tiles, or twice, gives the same bits."""
    p, _, W, H = sponza_pipeline
    host = capi.ProgressiveHost(1234)
    cam = cam_array(scenes.sponza_camera(), W / H)
    pfc = host.update(cam, 0.0, 1, W, H)
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    p.clear_output(); p.update(pfc); p.render()
    whole = p.read_output()
    st = p.stats()
    assert st["rays_primary"] == W * H
    assert st["rays_shadow"] == 2 * st["primary_hits"] + 2 * st["secondary_hits"]
    assert st["rays_secondary"] == 2 * st["primary_hits"]
    p.clear_output(); p.render()
    assert np.array_equal(p.read_output(), whole), "re-render differs"
    p.clear_output()
    for ty in range(3):
        for tx in range(4):
            p.render(tile=(tx * 480, ty * 360, (tx + 1) * 480, (ty + 1) * 360))
    assert np.array_equal(p.read_output(), whole), "tiled render differs from the whole frame"
    assert np.isfinite(whole).all() and whole[..., 3].min() == 1.0 and whole[..., :3].min() >= 0.0


def test_sponza_1080p_frame_digest(sponza_pipeline, capi):
    """Drift fixture: the sha256 of BASELINE config C2's first two accumulated 1080p frames (host seed 1234, the
    bench camera) is committed in tests/golden/reference_assets.json.  The oracle comparison of this scene runs at 192x108;
    at full size the image is only property-checked, so a change of any bit of it between rounds would otherwise go
    unnoticed.  (The exactness rule makes the image independent of tree, node width and traversal order: the digest of
    round 3's four-wide and eight-wide builds is the same.)  DXR_RECORD_DIGEST=1 prints the digest instead of checking it."""
    import hashlib
    import json
    import os
    p, _, W, H = sponza_pipeline
    host = capi.ProgressiveHost(1234)
    cam = cam_array(scenes.sponza_camera(), W / H)
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    p.clear_output()
    for f in range(2):
        p.update(host.update(cam, 0.0, f + 1, W, H))
        p.render()
    digest = hashlib.sha256(np.ascontiguousarray(p.read_output()).tobytes()).hexdigest()
    if os.environ.get("DXR_RECORD_DIGEST"):
        print("\nC2_1080P_2FRAMES_SHA256", digest)
        return
    want = json.load(open(GOLDEN + "/reference_assets.json"))["c2_1080p_two_frames_sha256"]
    assert digest == want, "the 1080p bench frame changed: %s" % digest


def test_sponza_1080p_sample_sharding_sum_equals_mean(sponza_pipeline, capi):
    """Multi-GPU partitioning A on one device: R shards render disjoint frame subsets into SUM
    buffers; (sum of sums)/N must match the running mean within fp32 re-association."""
    p, _, W, H = sponza_pipeline
    cam = cam_array(scenes.sponza_camera(), W / H)
    N, R = 8, 4
    host = capi.ProgressiveHost(5)
    pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(N)]
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    p.clear_output()
    for f in range(N):
        p.update(pfcs[f]); p.render()
    mean = p.read_output()
    p.set_accumulation_mode(T.ACCUM_SUM)
    total = np.zeros_like(mean, dtype=np.float64)
    for r in range(R):
        p.clear_output()
        for f in range(r, N, R):
            p.update(pfcs[f]); p.render()
        total += p.read_output()
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    shard_mean = (total / N).astype(np.float32)
    assert rms(shard_mean, mean) <= 1e-5
    assert np.abs(shard_mean - mean).max() <= 1e-4 * max(1.0, float(mean.max()))


def test_count_walk_is_the_production_walk(gpu, capi):
    """rt_pipeline_count_walk re-walks the last frame's queues with the production traversal: its ray tallies are the
    frame's ray counts, it is repeatable, it leaves the image untouched, and a change of scene invalidates it."""
    W, H = 160, 96
    v, i = triangle_soup(3000, 5)
    inst = [(0, None)]
    p = make_gpu_pipeline(capi, gpu, [(v, i)], inst, [T.default_material()], W, H, env=scenes.sky_cubemap(16))
    host = capi.ProgressiveHost(9)
    cam = cam_array(dict(eye=(0, 0, 30), at=(0, 0, 0), up=(0, 1, 0), fov=0.8), W / H)
    p.update(host.update(cam, 0.0, 1, W, H))
    p.render()
    img = p.read_output()
    st = p.stats()
    w1 = p.count_walk()
    w2 = p.count_walk()
    assert w1 == w2
    assert w1["primary"]["rays"] == st["rays_primary"] == W * H
    assert w1["secondary"]["rays"] == st["rays_secondary"]
    assert w1["shadow0"]["rays"] + w1["shadow1"]["rays"] == st["rays_shadow"] and st["rays_shadow_skipped"] == 0
    for s in w1.values():
        assert s["instance_entries"] == 0                       # one identity instance: single-level walk
        assert s["nodes_global"] + s["nodes_lds"] >= s["rays"] // 2 and s["tris"] > 0
        assert 0 < s["lines"] <= 2 * s["nodes_global"] + 2 * s["tris"]        # 64-B lines; lanes of a wave share node records (two lines each)
        assert 0 < s["longest_walk"] < 500
    assert w1["primary"]["nodes_lds"] > 0                       # the top of the tree is LDS resident
    assert w1["primary"]["lines"] < w1["primary"]["nodes_global"] + w1["primary"]["tris"] * 3 // 2    # primary rays of a tile share lines
    assert np.array_equal(img, p.read_output())
    cw = p.count_work()
    assert cw["primary"]["rays"] == W * H
    # instanced (two-level) scene: instance entries are counted
    xf = random_xforms(5, 3)
    p2 = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, xf[k]) for k in range(5)], [T.default_material()] * 5, W, H)
    p2.update(host.update(cam, 0.0, 2, W, H))
    p2.render()
    w = p2.count_walk()
    assert w["primary"]["instance_entries"] > 0 and w["primary"]["rays"] == W * H
    # stale state is refused, not replayed (last_pd held raw device pointers)
    p2.add_material(T.default_material())
    with pytest.raises(capi.RtError):
        p2.count_walk()
    with pytest.raises(capi.RtError):
        p2.count_work()


def test_cornell_lit_by_the_reference_environment_map(gpu, capi, tmp_path):
    """A Cornell frame lit by the committed 32^2 down-sample of the reference's CathedralRadiance.dds
    (ProgressiveRaytracingPipeline.cpp:114-118) against the committed oracle images, for both cube filters, through the
    array entry point and through a DDS file; no oracle at run time."""
    import struct
    g = np.load(GOLDEN + "/cathedral32.npz")
    faces = g["faces32"]
    path = str(tmp_path / "cathedral32.dds")
    hdr = struct.pack("<4sI", b"DDS ", 124) + struct.pack("<IIIIII", 0x1007, 32, 32, 32 * 16, 0, 1) + b"\0" * 44
    hdr += struct.pack("<II4sIIIII", 32, 4, b"DX10", 0, 0, 0, 0, 0) + struct.pack("<IIIII", 0x1008, 0xFE00, 0, 0, 0)
    hdr += struct.pack("<IIIII", 2, 3, 4, 1, 0)
    with open(path, "wb") as f:
        f.write(hdr + faces.astype(np.float32).tobytes())
    assert np.array_equal(capi.dds_read_cube(path), faces)
    m = capi.Model(gpu, path=CORNELL_OBJ)
    for via_dds in (False, True):
        sc = capi.Scene(gpu)
        sc.add_model(m)
        p = capi.Pipeline(gpu)
        p.set_scene(sc)
        p.add_material(T.default_material())
        if via_dds:
            p.load_environment_dds(path)
        else:
            p.set_environment_cube(faces)
        p.create_output(64, 64)
        p.build_acceleration_structures()
        for seamless, key in ((True, "cornell_lit"), (False, "cornell_lit_clamp")):
            p.set_environment_filter(seamless)
            p.clear_output()
            p.update(g["cornell_pfc"])
            p.render()
            assert np.array_equal(p.read_output(), g[key]), "via_dds=%s seamless=%s" % (via_dds, seamless)
    assert not np.array_equal(g["cornell_lit"], g["cornell_lit_clamp"])


@pytest.mark.parametrize("world,band", [(1, 16), (3, 8), (4, 24)])
def test_render_bands_equals_the_whole_frame(gpu, capi, world, band):
    """rt_pipeline_render_bands (the per-frame call of a tile-partitioned multi-GPU run): the ranks' interleaved bands,
    rendered one rank after the other into one image, equal the whole frame bit for bit, ragged last band included;
    ray counts add up."""
    W, H = 200, 141
    v, i = triangle_soup(2500, 9)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(8))
    host = capi.ProgressiveHost(2)
    cam = cam_array(dict(eye=(0, 0, 30), at=(0, 0, 0), up=(0, 1, 0), fov=0.8), W / H)
    pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(2)]
    for pfc in pfcs:
        p.update(pfc)
        p.render()
    whole, st = p.read_output(), p.stats()
    p.clear_output()
    rays = 0
    for pfc in pfcs:
        p.update(pfc)
        rays = 0
        for r in range(world):
            p.render_bands(band, r, world)
            s = p.stats()
            rays += s["rays_primary"]
            assert s["rays_primary"] == W * sum(y1 - y0 for y0, y1 in capi.tile_bands(H, band, r, world))
    assert np.array_equal(p.read_output(), whole)
    assert rays == st["rays_primary"] == W * H
    with pytest.raises(capi.RtError):
        p.render_bands(12, 0, 1)                    # not a multiple of the 8x8 pixel tiles
    with pytest.raises(capi.RtError):
        p.render_bands(16, 2, 2)


def test_unlit_shadow_rays_are_counted_but_not_traversed(gpu, capi, oracle):
    """evaluateDirectionalLight / evaluatePointLight trace their shadow ray even when N.L == 0 and multiply its visibility by
    that zero (RaytracingCommon.hlsli:126-147).  Skipping the traversal of exactly those rays leaves image and emitted-ray counts
    (the oracle's) unchanged; only rays_shadow_skipped tells."""
    W, H = 120, 90
    v, i = oracle.obj_load(CORNELL_OBJ)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(8))
    host = capi.ProgressiveHost(12)
    c = scenes.cornell_camera()
    cam = cam_array(c, W / H)
    pfc = host.update(cam, 0.0, 1, W, H)
    imgs, stats = [], []
    for on in (True, False):
        p.set_skip_unlit_shadow_rays(on)
        p.clear_output()
        p.update(pfc)
        p.render()
        imgs.append(p.read_output())
        stats.append(p.stats())
    ref, ost = osc.render(T.default_material(), pfc, W, H, env_faces=scenes.sky_cubemap(8))
    assert np.array_equal(imgs[0], ref) and np.array_equal(imgs[1], ref)
    assert stats[0]["rays_shadow"] == stats[1]["rays_shadow"] == ost["rays_shadow"]
    assert stats[1]["rays_shadow_skipped"] == 0 and 0 < stats[0]["rays_shadow_skipped"] < stats[0]["rays_shadow"]
