"""Round 4: sets of frames BEHIND the reference's per-frame calls, bands x frames, and queue memory.

The reference's app loop is one update() + one render() per frame (src/DXRExperimentsApp.cpp:162-165, :194), each frame
folded into gOutput with its own accumCount (src/ProgressiveRaytracingPipeline.cpp:188-195,
assets/shaders/ProgressiveRaytracing.hlsl:36-38).  Deferred mode (rt_pipeline_set_deferred) keeps those calls and renders the
recorded frames through shared sets of launches; whatever reads or changes what they produce flushes them first.  The image and
the ray counts must not be able to tell -- under any interleaving of reads.

Queue memory: a set sizes its ray / hit / shadow queues for the worst case when that fits the budget, else level by level from
the counts of the compaction before (rt_pipeline_set_queue_budget); a growth that fails must leave the pipeline usable
(the out-of-memory fallback of rt_pipeline_render_batch)."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from test_gpu_batch import frames_of
from test_gpu_pipeline import make_gpu_pipeline, make_oracle_scene
from util import CORNELL_OBJ, cam_array, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu

COUNTS = ("rays_primary", "rays_secondary", "rays_shadow", "rays_shadow_skipped", "primary_hits", "secondary_hits", "frames")


def atrium(capi, gpu, W, H, mat=None):
    v, i = scenes.sponza_class(seed=42)
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat or T.default_material()], W, H, env=scenes.sky_cubemap(16))
    return p, cam_array(scenes.sponza_camera(), W / H)


def immediate(p, pfcs):
    p.set_deferred(0)
    p.clear_output()
    p.reset_totals()
    for c in pfcs:
        p.update(c); p.render()
    return p.read_output(), p.totals()


@pytest.mark.parametrize("S", [2, 5, 32])
def test_deferred_equals_immediate(gpu, capi, S):
    """11 frames through update() + render() only, in sets of S: same bits, same ray counts; the flush is the read"""
    W, H = 192, 108
    p, cam = atrium(capi, gpu, W, H)
    pfcs = frames_of(capi, cam, 11, W, H)
    want, t1 = immediate(p, pfcs)
    p.clear_output()
    p.reset_totals()
    p.set_deferred(S)
    for k, c in enumerate(pfcs):
        p.update(c); p.render()
        assert p.deferred() == (S, (k + 1) % S), "frames held after frame %d" % k
    got, t2 = p.read_output(), p.totals()          # (the read renders what is still held)
    assert p.deferred()[1] == 0
    assert np.array_equal(want, got), "%d pixels differ" % int((want != got).any(axis=2).sum())
    for k in COUNTS:
        assert t1[k] == t2[k], (k, t1[k], t2[k])


def test_deferred_interleaved_reads_and_changes(gpu, capi, tmp_path):
    """every kind of call that must see -- or must not disturb -- the frames recorded before it, between the frames of one run:
    the deferred run goes through the same sequence of calls as the immediate one and every intermediate read agrees"""
    W, H = 96, 64
    m = capi.Model(gpu, path=CORNELL_OBJ)
    sc = capi.Scene(gpu)
    sc.add_model(m)
    p = capi.Pipeline(gpu)
    p.set_scene(sc)
    mat = T.default_material()
    p.add_material(mat)
    p.set_environment_constant((0.5, 0.5, 0.5))
    p.create_output(W, H)
    p.build_acceleration_structures()
    cam = cam_array(scenes.cornell_camera(), W / H)
    pfcs = frames_of(capi, cam, 24, W, H)
    mat2 = T.default_material()
    mat2["albedo"] = (0.2, 0.4, 0.9, 1.0)
    mat2["type"] = 0

    def run(S):
        seen = []
        p.set_deferred(S)
        p.set_material(0, mat)
        p.set_environment_constant((0.5, 0.5, 0.5))
        p.set_depth_limits(1, 2)
        p.set_accumulation_mode(T.ACCUM_RUNNING_MEAN)
        p.clear_output()
        p.reset_totals()
        f = iter(pfcs)

        def frames(n):
            for _ in range(n):
                p.update(next(f)); p.render()
        frames(3)
        seen.append(p.read_output())                        # a read in the middle of a set
        frames(2)
        seen.append(p.totals()["rays_shadow"])              # statistics cover every frame render() has accepted
        frames(1)
        p.set_material(0, mat2)                             # a change: the six frames before it keep the old material
        frames(2)
        p.set_environment_constant((0.1, 0.7, 0.2))
        frames(2)
        p.set_depth_limits(2, 3)
        frames(2)
        p.update(next(f))                                   # an update() without its render(), then a flush: the frame is not lost,
        p.flush()                                           # and not rendered twice
        p.render()
        ck = str(tmp_path / ("ck%d.bin" % S))
        p.save_checkpoint(ck)                               # a checkpoint holds every frame rendered so far
        frames(3)
        gpu.synchronize()                                   # the context's synchronize covers recorded frames
        assert p.deferred()[1] == 0
        seen.append(p.totals()["frames"])
        frames(2)
        p.load_checkpoint(ck)                               # back to the checkpoint: the two frames before it are rendered first (and overwritten)
        frames(2)
        walk = p.count_walk()                               # the counting re-walk replays queues that must exist: it renders them
        assert p.deferred()[1] == 0 and walk["primary"]["rays"] > 0
        frames(2)
        p.clear_output()                                    # frames recorded before a clear do not leak into the cleared image
        frames(2)
        seen.append(p.read_output())
        seen.append({k: p.totals()[k] for k in COUNTS})
        return seen

    a, b = run(0), run(7)
    assert len(a) == len(b)
    for k, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, np.ndarray):
            assert np.array_equal(x, y), "observation %d: %d pixels differ" % (k, int((x != y).any(axis=2).sum()))
        else:
            assert x == y, (k, x, y)


def test_deferred_errors_surface_at_the_call(gpu, capi):
    W = H = 32
    p = capi.Pipeline(gpu)
    p.set_deferred(8)
    v, i = triangle_soup(50, seed=1)
    sc = capi.Scene(gpu)
    sc.add_model(capi.Model(gpu, v, i))
    p.set_scene(sc)
    p.add_material(T.default_material())
    p.create_output(W, H)
    with pytest.raises(capi.RtError):
        p.render()                                          # not built, no update(): refused now, not at a later flush
    p.build_acceleration_structures()
    with pytest.raises(capi.RtError):
        p.render()                                          # no update() yet
    assert p.deferred() == (8, 0)
    rt = capi.Pipeline(gpu, kind=capi.PIPELINE_REALTIME)
    with pytest.raises(capi.RtError):
        rt.set_deferred(4)                                  # nothing accumulates there
    rt.set_deferred(0)


def test_deferred_scene_change_flushes(gpu, capi):
    """frames recorded before a scene gains an instance see the scene as it was"""
    W, H = 64, 48
    blob = scenes.blob_mesh(level=2)
    xf = random_xforms(3, seed=5, spread=4.0)
    cam = np.array([0, 2, 14, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)
    pfcs = frames_of(capi, cam, 6, W, H)

    def run(S):
        gm = capi.Model(gpu, *blob)
        sc = capi.Scene(gpu)
        sc.add_model(gm, xf[0])
        sc.add_model(gm, xf[1])
        p = capi.Pipeline(gpu)
        p.set_scene(sc)
        for _ in range(3):
            p.add_material(T.default_material())
        p.set_environment_constant((0.5, 0.5, 0.5))
        p.create_output(W, H)
        p.build_acceleration_structures()
        p.set_deferred(S)
        for c in pfcs[:3]:
            p.update(c); p.render()
        sc.add_model(gm, xf[2])
        sc.build()
        for c in pfcs[3:]:
            p.update(c); p.render()
        return p.read_output()

    assert np.array_equal(run(0), run(16))


@pytest.mark.parametrize("world", [2, 8])
def test_bands_times_frames(gpu, capi, world):
    """rt_pipeline_render_bands_batch: every rank's bands of 9 frames through shared sets of launches, the ranks one after the
    other on one device == the whole frames rendered one by one, bit for bit (global pixel seed,
    assets/shaders/ProgressiveRaytracing.hlsl:89); four bounces, so the level-by-level resolve sees bands as well"""
    W, H, band = 320, 200, 16
    v, t = scenes.displaced_grid(96, seed=7)
    mat = T.default_material()
    mat["type"] = 2
    p = make_gpu_pipeline(capi, gpu, [(v, t)], [(0, None)], [mat], W, H, env=scenes.sky_cubemap(16))
    p.set_depth_limits(4, 2)
    cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
    pfcs = frames_of(capi, cam, 9, W, H, seed=5)
    want, t1 = immediate(p, pfcs)
    p.clear_output()
    p.reset_totals()
    for r in range(world):
        p.render_bands_batch(band, r, world, pfcs)
    got, t2 = p.read_output(), p.totals()
    assert np.array_equal(want, got), "%d pixels differ" % int((want != got).any(axis=2).sum())
    for k in COUNTS[:-1]:
        assert t1[k] == t2[k], (k, t1[k], t2[k])
    # ... and frame by frame, the round-3 call
    p.clear_output()
    for c in pfcs:
        p.update(c)
        for r in range(world):
            p.render_bands(band, r, world)
    assert np.array_equal(want, p.read_output())


@pytest.mark.parametrize("case", ["one_bounce", "four_bounces", "ao", "instanced"])
def test_counted_queues_equal_worst_case_queues(gpu, capi, case):
    """a budget of 1 byte forces every level to be sized by count: same bits, same counts, less memory"""
    W, H = 192, 108
    if case == "instanced":
        blob = scenes.blob_mesh(level=2)
        soup = triangle_soup(400, seed=4, extent=2.0, size=0.5)
        xf = random_xforms(12, seed=11, spread=6.0)
        p = make_gpu_pipeline(capi, gpu, [blob, soup], [(k % 2, xf[k]) for k in range(12)], [T.default_material() for _ in range(12)], W, H,
                              env=scenes.sky_cubemap(16))
        cam = np.array([0, 2, 16, 0, 0, 0, 0, 1, 0, 0.8, W / H], np.float32)
    else:
        mat = T.default_material()
        if case == "four_bounces":
            mat["type"] = 2; mat["reflectivity"] = 0.6; mat["roughness"] = 0.3
        p, cam = atrium(capi, gpu, W, H, mat)
        if case == "four_bounces":
            p.set_depth_limits(4, 3)
    pfcs = frames_of(capi, cam, 6, W, H, options={"showAmbientOcclusionOnly": 1} if case == "ao" else None)
    want, t1 = immediate(p, pfcs)
    mem_worst, counted = p.queue_memory()
    assert not counted
    # (this pipeline keeps the worst-case buffers it has -- buffers never shrink --, so the memory comparison is a test of
    # its own, on fresh pipelines; here: the bits)
    p.set_queue_budget(1)
    p.clear_output()
    p.reset_totals()
    p.render_batch(pfcs)
    got, t2 = p.read_output(), p.totals()
    assert p.queue_memory()[1], "the set did not size its levels by count"
    assert np.array_equal(want, got), "%d pixels differ" % int((want != got).any(axis=2).sum())
    for k in COUNTS:
        assert t1[k] == t2[k], (k, t1[k], t2[k])
    # single frames by count as well, and the counting re-walks on counted queues
    p.clear_output()
    for c in pfcs:
        p.update(c); p.render()
    assert np.array_equal(want, p.read_output())
    walk = p.count_walk()
    assert walk["primary"]["rays"] == t1["rays_primary"] // 6
    p.set_queue_budget(0)


def test_counted_queues_take_less_memory(gpu, capi):
    W, H = 384, 216
    mat = T.default_material()
    mat["type"] = 2; mat["reflectivity"] = 0.6; mat["roughness"] = 0.3
    mem = {}
    for budget in (0, 1):
        p, cam = atrium(capi, gpu, W, H, mat)
        p.set_depth_limits(4, 2)
        p.set_queue_budget(budget)
        p.render_batch(frames_of(capi, cam, 8, W, H))
        mem[budget], counted = p.queue_memory()
        assert counted == bool(budget)
        p.close()
    assert mem[1] < 0.85 * mem[0], mem          # (a closed atrium: almost every ray hits; open scenes save more)


def test_failed_queue_growth_leaves_the_pipeline_usable(gpu, capi):
    """a set of 8 succeeds, then sets of 32 and 16 cannot be allocated (injected: allocations above a limit fail as
    if the device were full) and the fallback of rt_pipeline_render_batch halves the set until it fits -- the retries used to
    run on freed queue pointers.  The image must be the one immediate rendering gives; a plain render() after the failure too."""
    W, H = 192, 108
    p, cam = atrium(capi, gpu, W, H)
    pfcs = frames_of(capi, cam, 40, W, H)
    want, t1 = immediate(p, pfcs)
    q, _ = atrium(capi, gpu, W, H)
    q.render_batch(pfcs[:8])
    biggest = max(W * H * 8 * 2 * 16, 1) * 9 // 8              # the largest buffer of a worst-case set of 8: level 1's rays
    try:
        capi.lib().rt_debug_set_alloc_limit(int(biggest * 1.3))
        q.clear_output()
        q.reset_totals()
        q.render_batch(pfcs)                                # 32 -> 16 -> 8 (fits), then the remaining 8
        got, t2 = q.read_output(), q.totals()
        assert np.array_equal(want, got), "%d pixels differ" % int((want != got).any(axis=2).sum())
        for k in COUNTS:
            assert t1[k] == t2[k], (k, t1[k], t2[k])
        # a growth that fails outright: the call reports it, and the pipeline works once there is room
        capi.lib().rt_debug_set_alloc_limit(0)
        r, _ = atrium(capi, gpu, 64, 48)
        r.update(frames_of(capi, cam, 1, 64, 48)[0])
        capi.lib().rt_debug_set_alloc_limit(1024)
        with pytest.raises(capi.RtError):
            r.render()
        capi.lib().rt_debug_set_alloc_limit(0)
        r.render()
        assert np.isfinite(r.read_output()).all()
        q.clear_output()
        for c in pfcs[:3]:
            q.update(c); q.render()
        p.clear_output()
        for c in pfcs[:3]:
            p.update(c); p.render()
        assert np.array_equal(p.read_output(), q.read_output())
    finally:
        capi.lib().rt_debug_set_alloc_limit(0)


def test_reserve_batch_leaves_nothing_to_allocate(gpu, capi):
    """rt_pipeline_reserve_batch(S) reserves EVERYTHING a set of S frames allocates -- queues, constants, the shadow cache's table
    and the traversal kernels' global stack rows (4.3 GB for 20 frames of 1080p; round 4: that one was missing, and on some boxes
    its hipMalloc took 126 ms inside the first set's render() call, i.e. inside bench.py's timed region).  With every device
    allocation forbidden after the call, a deferred set of S frames (after a smaller warm-up set, as bench.py issues them) must
    still render, and to the same bits."""
    W, H, S = 320, 180, 12
    p, cam = atrium(capi, gpu, W, H)
    pfcs = frames_of(capi, cam, 3 + S, W, H)
    want, t1 = immediate(p, pfcs)
    q, _ = atrium(capi, gpu, W, H)
    q.reserve_batch(S)
    q.set_deferred(S)
    for c in pfcs[:3]:                      # a smaller set first
        q.update(c); q.render()
    q.flush()
    try:
        capi.lib().rt_debug_set_alloc_limit(1)
        for c in pfcs[3:]:
            q.update(c); q.render()         # the S-th call renders the set
        assert q.deferred()[1] == 0
        capi.lib().rt_debug_set_alloc_limit(0)
        got, t2 = q.read_output(), q.totals()
    finally:
        capi.lib().rt_debug_set_alloc_limit(0)
    assert np.array_equal(want, got)
    for k in COUNTS:
        assert t1[k] == t2[k], (k, t1[k], t2[k])


def test_sponza_1080p_window_against_the_oracle(gpu, capi, oracle):
    """BASELINE config C2 at its stated size against the CPU restatement: a 96x32 window of the 1920x1080 frame (RayGen's
    addressing at full size: assets/shaders/ProgressiveRaytracing.hlsl:18-38), two accumulated frames, bit for bit -- the
    whole frame is digest-pinned beside it (test_gpu_pipeline.py::test_sponza_1080p_frame_digest); rendered in deferred mode,
    i.e. both frames through one set of launches, which is how bench.py's headline renders them"""
    W, H = 1920, 1080
    v, i = scenes.sponza_class()
    env = scenes.sky_cubemap(64)
    mat = T.default_material()
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H, env=env)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    host = capi.ProgressiveHost(1234)
    cam = cam_array(scenes.sponza_camera(), W / H)
    p.set_deferred(2)
    for tile in ((912, 524, 1008, 556), (1824, 1048, 1920, 1080)):      # the middle of the frame; its last rows and columns
        p.clear_output()
        host = capi.ProgressiveHost(1234)
        acc = np.zeros((H, W, 4), np.float32)
        for f in range(2):
            pfc = host.update(cam, 0.0, f + 1, W, H)
            p.update(pfc)
            p.render()
            acc, _ = osc.render(mat, pfc, W, H, accum=acc, env_faces=env, tile=tile, nthreads=8)
        img = p.read_output()
        x0, y0, x1, y1 = tile
        a, b = img[y0:y1, x0:x1], acc[y0:y1, x0:x1]
        assert np.array_equal(a, b), "%d of %d window pixels differ" % (int((a != b).any(axis=2).sum()), (x1 - x0) * (y1 - y0))
        assert (b[..., :3] > 0).any() and b[..., 3].min() == 1.0


def test_sponza_1080p_whole_frame_against_the_oracle(gpu, capi, oracle):
    """... and the WHOLE 1920x1080 frame, two accumulated frames, every pixel and every ray count against the CPU restatement (the
    oracle traces a 1080p frame's 15.9 M rays in about a second per frame on the box's 16 host cores)"""
    import os
    W, H = 1920, 1080
    v, i = scenes.sponza_class()
    env = scenes.sky_cubemap(64)
    mat = T.default_material()
    p = make_gpu_pipeline(capi, gpu, [(v, i)], [(0, None)], [mat], W, H, env=env)
    osc = make_oracle_scene(oracle, [(v, i)], [(0, None)])
    host = capi.ProgressiveHost(1234)
    cam = cam_array(scenes.sponza_camera(), W / H)
    p.set_deferred(2)
    p.reset_totals()
    acc = np.zeros((H, W, 4), np.float32)
    rays = {"rays_primary": 0, "rays_secondary": 0, "rays_shadow": 0, "primary_hits": 0, "secondary_hits": 0}
    for f in range(2):
        pfc = host.update(cam, 0.0, f + 1, W, H)
        p.update(pfc)
        p.render()
        acc, ost = osc.render(mat, pfc, W, H, accum=acc, env_faces=env, nthreads=max(1, len(os.sched_getaffinity(0))))
        for k in rays:
            rays[k] += ost[k]
    img = p.read_output()
    assert np.array_equal(img, acc), "%d of %d pixels differ" % (int((img != acc).any(axis=2).sum()), W * H)
    tot = p.totals()
    for k in rays:
        assert tot[k] == rays[k], (k, tot[k], rays[k])


@pytest.mark.parametrize("two_level", [False, True])
def test_primary_launch_without_stack_rows_hands_rays_to_the_retry_launch(gpu, capi, two_level):
    """Round 5: the one-tile-per-wave primary launch keeps no stack rows beyond LDS (one thread per pixel slot made them 224 MB per 1080p frame
    of a set); a ray that would need one goes to a list the persistent k_primary_retry walks, and a list that overflows stands for every pixel
    slot.  The 6-row build of the kernels sends most primary rays of a deep soup that way: the image and the ray counts must be those of the
    production build bit for bit -- with a list that holds them all, with one of 7 entries (overflow), frame by frame and in a set, single-level
    and (primary_persistent=0) two-level."""
    from util import triangle_soup, random_xforms
    W, H = 200, 120
    v, i = triangle_soup(60000, seed=77, extent=6.0, size=0.5)
    inst = [(0, None)] if not two_level else [(0, x) for x in random_xforms(3, seed=5, spread=3.0)]
    mat = T.default_material()
    cam = cam_array(dict(eye=(0.0, 1.0, 16.0), at=(0.0, 0.0, 0.0), up=(0, 1, 0), fov=0.8), W / H)

    def render(ctx, frames, deferred):
        p = make_gpu_pipeline(capi, ctx, [(v, i)], inst, [mat], W, H)
        host = capi.ProgressiveHost(9)
        p.set_deferred(deferred)
        p.reset_totals()
        for f in range(frames):
            p.update(host.update(cam, 0.0, f + 1, W, H))
            p.render()
        img = p.read_output()
        return img, p.totals()

    want, wt = render(gpu, 3, 0)
    for opts in ({"lds_stack_rows": 6}, {"lds_stack_rows": 6, "primary_retry_cap": 7}):
        ctx = capi.Context(0)
        for k, val in opts.items():
            ctx.set_option(k, val)
        if two_level:
            ctx.set_option("primary_persistent", 0)          # (two-level scenes run the primary stage as a persistent launch by default)
        for deferred in (0, 3):
            got, gt = render(ctx, 3, deferred)
            assert np.array_equal(got, want), (opts, deferred, int((got != want).any(axis=2).sum()))
            for k in ("rays_primary", "rays_secondary", "rays_shadow", "primary_hits", "secondary_hits"):
                assert gt[k] == wt[k], (opts, deferred, k)
