"""SURVEY 8(f) rows N2 / N3 on the GPU: RealtimeRaytracingPipeline (two AOVs) and DenoiseCompositor
(separable joint-bilateral + tone map) == their oracle restatements, bit for bit."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import CORNELL_OBJ, cam_array, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def realtime_pair(capi, oracle, ctx, models, instances, mats, W, H, env):
    sc = capi.Scene(ctx)
    gm = [capi.Model(ctx, v, i) for v, i in models]
    osc = oracle.Scene()
    for v, i in models:
        osc.add_model(v, i)
    for mi, x in instances:
        sc.add_model(gm[mi], x)
        osc.add_instance(mi, x)
    osc.build()
    p = capi.Pipeline(ctx, capi.PIPELINE_REALTIME)
    p.set_scene(sc)
    for m in mats:
        p.add_material(m)
    p.set_environment_cube(env)
    p.create_output(W, H)
    p.build_acceleration_structures()
    return p, osc


@pytest.mark.parametrize("mtype", [1, 0])
def test_realtime_pipeline_cornell(gpu, capi, oracle, mtype):
    W, H = 112, 80
    v, i = oracle.obj_load(CORNELL_OBJ)
    mat = T.default_material()
    mat["type"] = mtype
    env = scenes.sky_cubemap(16)
    p, osc = realtime_pair(capi, oracle, gpu, [(v, i)], [(0, None)], [mat], W, H, env)
    assert p.name == "Realtime Ray Tracing Pipeline" and p.num_outputs == 2
    host = capi.ProgressiveHost(9)
    cam = cam_array(scenes.cornell_camera(), W / H)
    for f in range(2):
        pfc = host.update_realtime(cam, 0.0, f + 1, W, H)
        assert int(pfc["cameraParams"]["accumCount"]) == 0 and float(pfc["options"]["environmentStrength"]) == 1.0
        assert int(pfc["options"]["maxIterations"]) == 0
        p.update(pfc)
        p.render()
        d, ind, ost = osc.render_realtime(mat, pfc, W, H, env_faces=env, nthreads=8)
        assert np.array_equal(p.read_output(0), d), "direct-lighting AOV differs (frame %d)" % f
        assert np.array_equal(p.read_output(1), ind), "indirect-specular AOV differs (frame %d)" % f
        gst = p.stats()
        for key in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
            assert gst[key] == ost[key], key


def test_realtime_pipeline_instanced_with_misses(gpu, capi, oracle):
    W, H = 96, 64
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(150, seed=4, extent=1.5, size=0.5)
    xf = random_xforms(12, seed=8, spread=5.0)
    inst = [(k % 2, xf[k]) for k in range(12)]
    mats = []
    for k in range(12):
        m = T.default_material()
        m["albedo"][:3] = (0.2 + 0.05 * k, 0.9 - 0.05 * k, 0.5)
        m["type"] = k % 3
        mats.append(m)
    env = scenes.sky_cubemap(8)
    p, osc = realtime_pair(capi, oracle, gpu, [blob, soup], inst, mats, W, H, env)
    host = capi.ProgressiveHost(10)
    cam = cam_array(dict(eye=(0, 2, 14), at=(0, 0, 0), up=(0, 1, 0), fov=0.8), W / H)
    pfc = host.update_realtime(cam, 0.0, 3, W, H)
    p.update(pfc)
    p.render()
    d, ind, ost = osc.render_realtime(np.stack(mats), pfc, W, H, env_faces=env, nthreads=8)
    assert np.array_equal(p.read_output(0), d) and np.array_equal(p.read_output(1), ind)
    assert 0 < ost["primary_hits"] < W * H
    # deeper specular chains (the reference compiles depth 1 in; the engine takes up to 4)
    p.set_depth_limits(3, 3)
    p.render()
    d, ind, ost = osc.render_realtime(np.stack(mats), pfc, W, H, env_faces=env, max_radiance_depth=3, max_shadow_depth=3, nthreads=8)
    assert np.array_equal(p.read_output(0), d) and np.array_equal(p.read_output(1), ind)
    gst = p.stats()
    for key in ("rays_secondary", "rays_shadow", "secondary_hits"):
        assert gst[key] == ost[key], key


def synthetic_aovs(W, H, seed):
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    direct = np.zeros((H, W, 4), np.float32)
    direct[..., 0] = 0.5 + 0.5 * np.sin(xx / 9.0)
    direct[..., 1] = (yy // 16 % 2) * 0.8               # hard edges the bilateral weight must respect
    direct[..., 2] = 0.3
    direct[..., 3] = 1.0
    ind = (r.uniform(0, 1, (H, W, 4)) ** 3).astype(np.float32)
    ind[..., 3] = 1.0
    return direct, ind


PARAM_CASES = [dict(), dict(maxKernelSize=1), dict(maxKernelSize=20), dict(maxKernelSize=0), dict(debugVisualize=1), dict(debugVisualize=2),
               dict(debugVisualize=3), dict(tonemap=0), dict(gammaCorrect=1), dict(exposure=2.5, gamma=1.8, gammaCorrect=1)]


@pytest.mark.parametrize("over", PARAM_CASES, ids=lambda o: "-".join("%s=%s" % kv for kv in o.items()) or "defaults")
def test_denoiser_vs_oracle(gpu, capi, oracle, over):
    W, H = 333, 141          # not multiples of the 256 / 32x32 tiles
    direct, ind = synthetic_aovs(W, H, 3)
    dn = capi.Denoiser(gpu)
    dn.create_output(W, H)
    prm = np.zeros((), oracle.DENOISE_PARAMS)
    prm["exposure"], prm["gamma"], prm["tonemap"], prm["gammaCorrect"], prm["maxKernelSize"], prm["debugVisualize"] = 1.0, 2.2, 1, 0, 12, 0
    assert dn.params[0].tobytes() == prm.tobytes()                       # reference defaults (DenoiseCompositor.cpp:44-49)
    for k, val in over.items():
        prm[k] = val
        dn.params[k] = val
    td, ti = gpu.upload(direct), gpu.upload(ind)
    dn.dispatch(td.ptr, ti.ptr)
    oh, ov = oracle.denoise(direct, ind, prm)
    gh, gv = dn.read_intermediate(), dn.read_output()
    same = lambda a, b: np.array_equal(a, b) or np.array_equal(np.nan_to_num(a, nan=-1.0), np.nan_to_num(b, nan=-1.0))
    assert same(gh, oh), "pass H differs on %d texels" % int((gh != oh).any(axis=2).sum())
    assert same(gv, ov), "pass V differs on %d texels" % int((gv != ov).any(axis=2).sum())
    assert dn.last_ms() >= 0


def test_denoiser_rejects_oversized_kernel_and_missing_output(gpu, capi):
    dn = capi.Denoiser(gpu)
    t = gpu.upload(np.zeros((8, 8, 4), np.float32))
    assert np.array_equal(t.download(), np.zeros(256, np.float32))
    with pytest.raises(capi.RtError):
        dn.dispatch(t.ptr, t.ptr)                          # no output resource yet
    dn.create_output(8, 8)
    dn.params["maxKernelSize"] = 25
    with pytest.raises(capi.RtError):
        dn.dispatch(t.ptr, t.ptr)


def test_realtime_then_denoise_1080p(gpu, capi):
    """Config 4's post chain at full size on the Sponza-class scene: runs, finite, and the filter smooths
    the indirect-specular AOV (lower variance than its input) while leaving direct lighting composited."""
    W, H = 1920, 1080
    v, i = scenes.sponza_class()
    sc = capi.Scene(gpu)
    sc.add_model(capi.Model(gpu, v, i))
    p = capi.Pipeline(gpu, capi.PIPELINE_REALTIME)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.set_environment_cube(scenes.sky_cubemap(64))
    p.create_output(W, H)
    p.build_acceleration_structures()
    host = capi.ProgressiveHost(1)
    p.update(host.update_realtime(cam_array(scenes.sponza_camera(), W / H), 0.0, 1, W, H))
    p.enable_timing(1)
    p.render()
    dn = capi.Denoiser(gpu)
    dn.create_output(W, H)
    dn.params["tonemap"] = 0
    dn.params["debugVisualize"] = 1                       # denoised indirect only
    dn.dispatch(p.output_device_ptr(0), p.output_device_ptr(1))
    ind, den = p.read_output(1), dn.read_output()
    assert np.isfinite(den).all() or np.isnan(den).sum() < den.size // 1000
    lap = lambda a: float(np.nanmean(np.abs(a[1:-1, 1:-1, :3] * 4 - a[:-2, 1:-1, :3] - a[2:, 1:-1, :3] - a[1:-1, :-2, :3] - a[1:-1, 2:, :3])))
    assert lap(den) < 0.5 * lap(ind)
    st = p.stats()
    assert st["rays_secondary"] <= st["primary_hits"] and st["rays_shadow"] == 2 * st["primary_hits"] + 2 * st["secondary_hits"]
    print("realtime frame %.3f ms, denoise %.3f ms" % (st["ms_total"], dn.last_ms()))


def test_structures_fixture_without_the_oracle(gpu, capi):
    """The committed fixture (tests/golden/cornell64_structures.npz) against the library alone: canonical LBVH arrays
    index-exact, both realtime AOVs and the denoised composite bit-exact."""
    import os
    from util import GOLDEN
    g = np.load(os.path.join(GOLDEN, "cornell64_structures.npz"))
    sc = capi.Scene(gpu)
    sc.add_model(capi.Model(gpu, path=CORNELL_OBJ))
    p = capi.Pipeline(gpu, capi.PIPELINE_REALTIME)
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.set_environment_constant((0.5, 0.5, 0.5))
    p.create_output(64, 64)
    p.build_acceleration_structures()
    nodes, keys, parents, depth = sc.bvh(0)
    assert np.ascontiguousarray(nodes).tobytes() == g["bvh_nodes"].tobytes()
    assert np.array_equal(keys, g["bvh_keys"]) and np.array_equal(parents, g["bvh_parents"]) and depth == int(g["bvh_depth"])
    p.update(g["realtime_pfc"])
    p.render()
    assert np.array_equal(p.read_output(0), g["direct"]) and np.array_equal(p.read_output(1), g["indirect"])
    dn = capi.Denoiser(gpu)
    prm = dn.params
    want = np.frombuffer(g["denoise_params"].tobytes(), prm.dtype)[0]
    for name in prm.dtype.names:
        prm[name] = want[name]
    dn.create_output(64, 64)
    dn.dispatch(p.output_device_ptr(0), p.output_device_ptr(1))
    assert np.array_equal(dn.read_output(), g["denoised"])


def test_denoiser_on_the_reference_mock_inputs(gpu, capi):
    """The reference's own denoiser test inputs (crops of assets/textures/DirectLighting.PNG and IndirectSpecular.PNG,
    loaded by DenoiseCompositor::loadResources(loadMockResources), src/DenoiseCompositor.cpp:52-68): both passes equal
    the committed oracle composite; no oracle at run time."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "denoise_mock.npz"))
    H, W = g["direct_rgba8"].shape[:2]
    direct = np.ones((H, W, 4), np.float32)
    indirect = np.ones((H, W, 4), np.float32)
    direct[..., :g["direct_rgba8"].shape[2]] = g["direct_rgba8"].astype(np.float32) / np.float32(255.0)
    indirect[..., :g["indirect_rgba8"].shape[2]] = g["indirect_rgba8"].astype(np.float32) / np.float32(255.0)
    dn = capi.Denoiser(gpu)
    dn.create_output(W, H)
    want = np.frombuffer(g["denoise_params"].tobytes(), capi.DENOISER_PARAMS)[0]
    for k in capi.DENOISER_PARAMS.names:
        dn.params[k] = want[k]
    td, ti = gpu.upload(direct), gpu.upload(indirect)
    dn.dispatch(td.ptr, ti.ptr)
    assert np.array_equal(dn.read_intermediate(), g["pass_h"])
    assert np.array_equal(dn.read_output(), g["composite"])
