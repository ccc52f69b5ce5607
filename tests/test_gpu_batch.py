"""rt_pipeline_render_batch: S frames through shared sets of launches == S x (update, render), bit for bit.

The reference renders one DispatchRays per frame and folds it into gOutput with that frame's accumCount
(src/ProgressiveRaytracingPipeline.cpp:188-195, assets/shaders/ProgressiveRaytracing.hlsl:36-38).  The batch mode (BASELINE
configs[2]: 256 spp accumulated) puts the rays of up to 32 frames into the same queues; the image and the ray counts must
not be able to tell.  Covered: partial batches, sets of the maximum size (32) and chunking beyond it, multi-bounce paths (the level-by-level resolve), the
debug / AO views (their own queue layouts), instanced two-level scenes, frames past maxIterations, SUM accumulation, and
config 3 end to end on one device (8 logical shards x 32 frames)."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from test_gpu_pipeline import make_gpu_pipeline, rms
from util import CORNELL_OBJ, cam_array, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def frames_of(capi, cam, n, W, H, seed=7, options=None, first=1):
    host = capi.ProgressiveHost(seed)
    for k, v in (options or {}).items():
        host.options[k] = v
    return [host.update(cam, 0.0, first + f, W, H) for f in range(n)]


def both_ways(p, pfcs):
    p.clear_output()
    p.reset_totals()
    for c in pfcs:
        p.update(c); p.render()
    one_by_one, t1 = p.read_output(), p.totals()
    p.clear_output()
    p.reset_totals()
    p.render_batch(pfcs)
    batched, t2 = p.read_output(), p.totals()
    assert np.array_equal(one_by_one, batched), "%d pixels differ" % int((one_by_one != batched).any(axis=2).sum())
    for k in ("rays_primary", "rays_secondary", "rays_shadow", "rays_shadow_skipped", "primary_hits", "secondary_hits", "frames"):
        assert t1[k] == t2[k], (k, t1[k], t2[k])
    return batched


@pytest.mark.parametrize("n", [1, 2, 5, 8, 11, 32, 35])
def test_cornell_batches(gpu, capi, n):
    W = H = 64
    m = capi.Model(gpu, path=CORNELL_OBJ)
    sc = capi.Scene(gpu)
    sc.add_model(m)
    p = capi.Pipeline(gpu)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.set_environment_constant((0.5, 0.5, 0.5))
    p.create_output(W, H)
    p.build_acceleration_structures()
    cam = cam_array(scenes.cornell_camera(), 1.0)
    img = both_ways(p, frames_of(capi, cam, n, W, H))
    assert img[..., 3].min() == 1.0 and img[..., 3].max() == 1.0


@pytest.mark.parametrize("case", ["default", "four_bounces", "ao", "one_light", "skip_unlit", "sum"])
def test_atrium_batches(gpu, capi, case):
    """the bench scene's camera at 192x108 (an odd number of 8x8 tiles per row), 6 frames"""
    W, H = 192, 108
    v, i = scenes.sponza_class(seed=42)
    mat = T.default_material()
    if case == "four_bounces":
        mat["type"] = 2; mat["reflectivity"] = 0.6; mat["roughness"] = 0.3
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H, env=scenes.sky_cubemap(16))
    opts = {"ao": {"showAmbientOcclusionOnly": 1}, "one_light": {"debug": 2}}.get(case)
    if case == "four_bounces":
        p.set_depth_limits(4, 3)
    if case == "skip_unlit":
        p.set_skip_unlit_shadow_rays(True)
    if case == "sum":
        p.set_accumulation_mode(T.ACCUM_SUM)
    cam = cam_array(scenes.sponza_camera(), W / H)
    both_ways(p, frames_of(capi, cam, 6, W, H, options=opts))


def test_instanced_scene_and_changing_constants(gpu, capi):
    """two-level walks; every frame of the batch with its own lights, options and camera jitter"""
    W, H = 96, 64
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(400, seed=4, extent=2.0, size=0.5)
    xf = random_xforms(24, seed=11, spread=6.0)
    mats = []
    for k in range(24):
        m = T.default_material()
        m["type"] = k % 3
        m["albedo"] = (0.2 + 0.03 * k, 0.9 - 0.03 * k, 0.3, 1.0)
        mats.append(m)
    p = make_gpu_pipeline(capi, gpu, [blob, soup], [(k % 2, xf[k]) for k in range(24)], mats, W, H, env=scenes.sky_cubemap(16))
    cam = np.array([0, 2, 16, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)
    pfcs = frames_of(capi, cam, 7, W, H)
    for f, c in enumerate(pfcs):                         # every frame its own light set-up and environment strength
        c["directionalLight"]["forwardDir"] = (0.3 - 0.1 * f, -0.2, -1.0 + 0.05 * f, 0.0)
        c["pointLight"]["worldPos"] = (0.5 * f, 1.0, 2.0 - f, 1.0)
        c["options"]["environmentStrength"] = 0.5 + 0.1 * f
        c["options"]["cosineHemisphereSampling"] = f % 2
    both_ways(p, pfcs)


def test_frames_past_max_iterations_are_skipped(gpu, capi):
    W = H = 48
    v, i = triangle_soup(2000, seed=9)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H)
    cam = np.array([0, 0, 25, 0, 0, 0, 0, 1, 0, 0.8, 1.0], np.float32)
    pfcs = frames_of(capi, cam, 9, W, H, options={"maxIterations": 4})      # accumCount 0..8: frames 4.. leave RayGen at once
    assert [int(c["cameraParams"]["accumCount"]) for c in pfcs][:5] == [0, 1, 2, 3, 4]
    both_ways(p, pfcs)
    assert p.totals()["frames"] == 4


def test_config3_on_one_device_8_shards_of_32_frames(gpu, capi):
    """BASELINE configs[2] end to end, one device standing in for eight: rank r renders frames {f : f mod 8 == r} of 256 into
    a SUM buffer in batches; (sum of the eight sums) / 256 against the 256-frame running mean: <= 1e-5 RMS."""
    W, H = 192, 108
    v, i = scenes.sponza_class(seed=42)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(16))
    cam = cam_array(scenes.sponza_camera(), W / H)
    N, R = 256, 8
    pfcs = frames_of(capi, cam, N, W, H, seed=1234)
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    p.clear_output()
    p.render_batch(pfcs)
    mean = p.read_output()
    p.set_accumulation_mode(T.ACCUM_SUM)
    total = np.zeros_like(mean, dtype=np.float64)
    for r in range(R):
        p.clear_output()
        p.render_batch(pfcs[r::R])
        total += p.read_output()
    shard_mean = (total / N).astype(np.float32)
    assert rms(shard_mean, mean) <= 1e-5
    assert shard_mean[..., 3].min() == 1.0 and np.isfinite(shard_mean).all()


def test_config3_at_full_size_1080p_256_frames(gpu, capi):
    """BASELINE configs[2] at the stated size on one device: 1920x1080, 256 frames.  The 256-frame running mean rendered in
    batches of 16 against the same 256 frames as eight SUM shards of 32 (what the eight ranks of a node would hold before
    their all-reduce): (sum of sums) / 256 within 1e-5 RMS; alpha stays exactly 1; and the first batch of the running mean
    equals 16 single frames bit for bit (the oracle comparison of this scene runs at 192x108, test_gpu_pipeline.py)."""
    W, H = 1920, 1080
    v, i = scenes.sponza_class(seed=42)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(64))
    cam = cam_array(scenes.sponza_camera(), W / H)
    N, R = 256, 8
    pfcs = frames_of(capi, cam, N, W, H, seed=1234)
    p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
    p.clear_output()
    for c in pfcs[:16]:
        p.update(c); p.render()
    first16 = p.read_output()
    p.clear_output()
    p.render_batch(pfcs[:16])
    assert np.array_equal(p.read_output(), first16), "a batch of 16 differs from 16 frames at 1080p"
    p.render_batch(pfcs[16:])
    mean = p.read_output()
    p.set_accumulation_mode(T.ACCUM_SUM)
    total = np.zeros_like(mean, dtype=np.float64)
    for r in range(R):
        p.clear_output()
        p.render_batch(pfcs[r::R])
        total += p.read_output()
    shard_mean = (total / N).astype(np.float32)
    assert rms(shard_mean, mean) <= 1e-5
    assert shard_mean[..., 3].min() == 1.0 and shard_mean[..., 3].max() == 1.0 and np.isfinite(shard_mean).all()


def test_reserve_batch(gpu, capi):
    """rt_pipeline_reserve_batch sizes the queues of a set of frames ahead of time: afterwards sets up to that size render
    (same bits as frame by frame) without the device allocator being called, and bad sizes are refused."""
    import ctypes

    def free_bytes():
        hip = ctypes.CDLL("libamdhip64.so")
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value
    W, H = 96, 64
    m = capi.Model(gpu, path=CORNELL_OBJ)
    sc = capi.Scene(gpu)
    sc.add_model(m)
    p = capi.Pipeline(gpu)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.set_environment_constant((0.5, 0.5, 0.5))
    p.create_output(W, H)
    p.build_acceleration_structures()
    cam = cam_array(scenes.cornell_camera(), W / H)
    pfcs = frames_of(capi, cam, 12, W, H)
    p.reserve_batch(12)
    gpu.synchronize()
    free0 = free_bytes()
    both_ways(p, pfcs)
    both_ways(p, pfcs[:5])
    gpu.synchronize()
    assert free_bytes() == free0, "a reserved set of frames allocated device memory"
    with pytest.raises(capi.RtError):
        p.reserve_batch(0)
    with pytest.raises(capi.RtError):
        p.reserve_batch(33)


def test_shadow_cache_is_invisible_in_instanced_scenes(gpu, capi):
    """the same for two-level walks: an entry is a (triangle, instance) pair and the ray starts inside that instance"""
    W, H = 96, 64
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(400, seed=4, extent=2.0, size=0.5)
    xf = random_xforms(24, seed=11, spread=6.0)
    mats = [T.default_material() for _ in range(24)]
    cam = np.array([0, 2, 16, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)
    pfcs = frames_of(capi, cam, 8, W, H)
    images, counts = [], []
    for cells in (0, -1, 16, 256):
        p = make_gpu_pipeline(capi, gpu, [blob, soup], [(k % 2, xf[k]) for k in range(24)], mats, W, H, env=scenes.sky_cubemap(16))
        p.set_shadow_cache(cells)
        images.append(both_ways(p, pfcs))
        assert (p.shadow_cache() != 0) == (cells != 0)
        t = p.totals()
        counts.append({k: t[k] for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits")})
    for img, cnt in zip(images[1:], counts[1:]):
        assert np.array_equal(img, images[0]) and cnt == counts[0]


@pytest.mark.parametrize("case", ["default", "four_bounces", "moving_lights"])
def test_shadow_cache_is_invisible(gpu, capi, case):
    """rt_pipeline_set_shadow_cache: the light buffer of occluders changes which triangle a shadow ray meets first, never
    whether it meets one -- the image and every ray count are the same with the cache off, automatic, with a table so coarse
    that most entries are wrong (16 cells per side), and after the lights have moved under it; frame by frame and in sets."""
    W, H = 192, 108
    v, i = scenes.sponza_class(seed=42)
    mat = T.default_material()
    if case == "four_bounces":
        mat["type"] = 2; mat["reflectivity"] = 0.6; mat["roughness"] = 0.3
    cam = cam_array(scenes.sponza_camera(), W / H)
    pfcs = frames_of(capi, cam, 9, W, H)
    if case == "moving_lights":
        for f, c in enumerate(pfcs):
            c["directionalLight"]["forwardDir"] = (0.3 - 0.15 * f, -0.2 - 0.05 * f, -1.0 + 0.1 * f, 0.0)
            c["pointLight"]["worldPos"] = (0.7 * f - 2.0, 1.0 + 0.2 * f, 2.0 - 0.5 * f, 1.0)
    images, counts = [], []
    for cells in (0, -1, 16, 512):
        p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H, env=scenes.sky_cubemap(16))
        if case == "four_bounces":
            p.set_depth_limits(4, 3)
        p.set_shadow_cache(cells)
        images.append(both_ways(p, pfcs))          # (frame by frame == in one set, and the image of the second pass)
        t = p.totals()
        counts.append({k: t[k] for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits")})
    for img, cnt in zip(images[1:], counts[1:]):
        assert np.array_equal(img, images[0]) and cnt == counts[0]
    with pytest.raises(capi.RtError):
        p.set_shadow_cache(9000)


def test_another_scene_on_the_same_pipeline(gpu, capi):
    """What a pipeline remembers about a scene -- shadow-cache entries (indices into its triangle array), the distance behind
    the point light's free sphere -- goes when it is given another scene, also one whose generation counter reads the same: the
    small scene rendered after the large one equals the small scene rendered by a fresh pipeline."""
    W, H = 96, 64
    big = scenes.sponza_class(seed=42)
    small = triangle_soup(300, seed=3, extent=3.0, size=1.2)
    cam_big = cam_array(scenes.sponza_camera(), W / H)
    cam_small = np.array([0, 0, 9, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)

    def scene_of(mesh):
        sc = capi.Scene(gpu)
        sc.add_model(capi.Model(gpu, *mesh))
        sc.build()
        return sc
    fresh = make_gpu_pipeline(capi, gpu, [small], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(16))
    want = both_ways(fresh, frames_of(capi, cam_small, 6, W, H))
    p = make_gpu_pipeline(capi, gpu, [big], [(0, None)], [T.default_material()], W, H, env=scenes.sky_cubemap(16))
    both_ways(p, frames_of(capi, cam_big, 6, W, H))            # fills the cache with indices up to 262 k
    p.set_scene(scene_of(small))
    p.build_acceleration_structures()
    got = both_ways(p, frames_of(capi, cam_small, 6, W, H))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scene", ["atrium", "instances"])
def test_free_sphere_against_the_oracle(gpu, oracle, capi, scene):
    """rt_pipeline_get_free_sphere: the point light's shadow rays stop at a sphere no triangle reaches into.  The radius arrives
    behind the first frame (a device pass, never waited for), is a lower bound of the light's distance from every vertex, is none
    when the light touches geometry, follows the light -- and the image stays the oracle's bit for bit, frame by frame and in sets."""
    from test_gpu_pipeline import make_oracle_scene
    W, H = 64, 40
    if scene == "atrium":
        models = [scenes.sponza_class(seed=42)]
        inst = [(0, None)]
        cam = cam_array(scenes.sponza_camera(), W / H)
        world = models[0][0]["position"]
    else:
        models = [scenes.blob_mesh(level=2), triangle_soup(300, seed=4, extent=2.0, size=0.5)]
        xf = random_xforms(12, seed=11, spread=6.0)
        inst = [(k % 2, xf[k]) for k in range(12)]
        cam = np.array([0, 2, 16, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)
        world = np.concatenate([models[mi][0]["position"] @ np.asarray(x, np.float64).reshape(3, 4)[:, :3].T + np.asarray(x, np.float64).reshape(3, 4)[:, 3]
                                for mi, x in inst])
    mats = [T.default_material() for _ in inst]
    env = scenes.sky_cubemap(8)
    p = make_gpu_pipeline(capi, gpu, models, inst, mats, W, H, env=env)
    osc = make_oracle_scene(oracle, models, inst)
    omats = np.stack(mats)
    pfcs = frames_of(capi, cam, 9, W, H)
    lights = [tuple(pfcs[0]["pointLight"]["worldPos"][:3]),              # the reference's
              tuple(np.float32(world[len(world) // 3])),                # on a vertex: no room for a sphere
              (60.0, 45.0, -30.0)]                                      # far outside the scene
    acc = np.zeros((H, W, 4), np.float32)
    f = 0
    for k, lp in enumerate(lights):
        for c in pfcs[3 * k:3 * k + 3]:
            c["pointLight"]["worldPos"][:3] = lp
        p.update(pfcs[f]); p.render()                   # the first frame with this light: queues the pass behind itself
        acc, _ = osc.render(omats, pfcs[f], W, H, accum=acc, env_faces=env, nthreads=8)
        assert np.array_equal(p.read_output(), acc)     # (read_output waits for the stream: the pass has landed)
        r = p.free_sphere()
        nearest = float(np.sqrt(((world.astype(np.float64) - np.array(lp, np.float64)) ** 2).sum(axis=1).min()))
        assert 0.0 <= r <= nearest
        if k == 1:
            assert r == 0.0
        elif k == 2:
            assert r > 0.8 * nearest - 6.0              # (instances: the bound is to their boxes)
        elif scene == "atrium":
            assert r > 1.0
        if k == 1:
            p.update(pfcs[f + 1]); p.render(); p.update(pfcs[f + 2]); p.render()
        else:
            p.render_batch(pfcs[f + 1:f + 3])           # these run with the sphere
        for c in pfcs[f + 1:f + 3]:
            acc, _ = osc.render(omats, c, W, H, accum=acc, env_faces=env, nthreads=8)
        assert np.array_equal(p.read_output(), acc)
        assert p.free_sphere() == r
        f += 3
