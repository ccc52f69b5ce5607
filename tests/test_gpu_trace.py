"""TraceRay on the GPU (production and canonical kernels) == oracle (BVH and brute force)."""
import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import ANY, CORNELL_OBJ, CULL, Pair, assert_hits_equal, primary_rays, random_rays, random_xforms, triangle_soup

pytestmark = pytest.mark.gpu


def compare_all(pair, O, D, brute=True):
    for flags in (0, CULL):
        ob = pair.o.trace(O, D, flags=flags, mode=1, nthreads=8)
        gf = pair.g.trace(O, D, flags=flags)
        gc = pair.g.trace(O, D, flags=flags, canonical=True)
        assert_hits_equal(gf, ob, "fast vs oracle bvh flags=%d" % flags)
        assert_hits_equal(gc, ob, "canonical vs oracle bvh flags=%d" % flags)
        assert np.array_equal(gc["nodes"], ob["nodes"]) and np.array_equal(gc["tris"], ob["tris"]), "traversal counters differ"
        if brute:
            assert_hits_equal(gf, pair.o.trace(O, D, flags=flags, mode=0, nthreads=8), "fast vs brute force flags=%d" % flags)
    oa = pair.o.trace(O, D, flags=ANY, mode=0, nthreads=8)
    assert_hits_equal(pair.g.trace(O, D, flags=ANY), oa, "any-hit fast vs brute", closest=False)
    gca = pair.g.trace(O, D, flags=ANY, canonical=True)
    oba = pair.o.trace(O, D, flags=ANY, mode=1, nthreads=8)
    assert_hits_equal(gca, oa, "any-hit canonical vs brute", closest=False)
    assert np.array_equal(gca["nodes"], oba["nodes"]) and np.array_equal(gca["tris"], oba["tris"])


def test_cornell_camera_and_random_rays(gpu, oracle, capi):
    v, i = oracle.obj_load(CORNELL_OBJ)
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    g = np.load(CORNELL_OBJ.replace("cornell.obj", "cornell64_golden.npz"))
    pf = np.frombuffer(g["pfc"][0].tobytes(), T.PER_FRAME_CONSTANTS)[0]
    O, D = primary_rays(pf, 64, 64)
    h = p.g.trace(O, D, flags=CULL)
    assert np.array_equal(h["prim"], g["prim"]) and np.array_equal(h["inst"], g["inst"])
    assert np.array_equal(h["t"], g["t"]) and np.array_equal(h["u"], g["u"]) and np.array_equal(h["v"], g["v"])
    O, D = random_rays(200000, 1, [-1, -1, -1], [1, 1, 1])
    compare_all(p, O, D)
    # axis-aligned and grazing rays along the walls (flat boxes)
    O2, D2 = random_rays(50000, 2, [-1, -1, -1], [1, 1, 1])
    D2[:, 1] = 0.0
    O2[::2, 1] = -1.0
    O2[1::4, 0] = 1.0
    compare_all(p, O2, D2)


def test_soup_with_tmin_tmax_windows(gpu, oracle, capi):
    p = Pair(oracle, capi, gpu, [triangle_soup(20000, seed=7)], [(0, None)])
    O, D = random_rays(100000, 3, [-10, -10, -10], [10, 10, 10])
    r = np.random.default_rng(4)
    O[:, 3] = r.uniform(0, 5, O.shape[0])
    D[:, 3] = O[:, 3] + r.uniform(-1, 20, O.shape[0])      # some windows empty / inverted
    compare_all(p, O, D, brute=True)


def test_instanced_two_level(gpu, oracle, capi):
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(300, seed=2, extent=1.5, size=0.4)
    xf = random_xforms(60, seed=5, spread=10.0)
    inst = [(k % 2, xf[k]) for k in range(60)] + [(1, None)]
    p = Pair(oracle, capi, gpu, [blob, soup], inst)
    O, D = random_rays(100000, 6, [-12, -12, -12], [12, 12, 12])
    compare_all(p, O, D, brute=True)


def test_edge_cases(gpu, oracle, capi):
    p = Pair(oracle, capi, gpu, [triangle_soup(1, seed=9, extent=0.1, size=1.0)], [(0, None)])
    # empty batch
    h = p.g.trace(np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32))
    assert h["t"].size == 0
    O, D = random_rays(4096, 8, [-1, -1, -1], [1, 1, 1])
    D[0, :3] = 0.0                      # zero direction
    D[1, 0] = np.nan                    # NaN direction
    O[2, 0] = np.nan                    # NaN origin
    D[3, :3] = [1, 0, 0]                # axis aligned (two infinite reciprocals)
    D[4, :3] = [0, -0.0, 1]
    D[5, 3] = -1.0                      # inverted window
    compare_all(p, O, D)


def test_sponza_class_rays(gpu, oracle, capi):
    v, i = scenes.sponza_class()
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    O, D = random_rays(300000, 10, [-16, -4, -7], [16, 7, 7])
    compare_all(p, O, D, brute=False)
    O, D = random_rays(3000, 11, [-16, -4, -7], [16, 7, 7])
    compare_all(p, O, D, brute=True)
    assert p.g.trace_last_ms() >= 0


def test_wide_node_culling_stays_conservative_in_hard_places(gpu, oracle, capi):
    """The production walk culls against quantised child boxes in the node's own frame with a rounding margin per axis, and
    sends steep rays down an exact path (rt_trace_wave.h, wide_step).  Places where a too-tight margin or a wrong path
    would lose a hit: geometry far from the coordinate origin (large |origin|, small extents), tiny triangles, rays with
    direction components from 1e-9 to 1e-3 (both sides of the 2^-16 steepness threshold) and exactly zero, rays that
    start on box planes and run inside axis-aligned walls.  Every hit must equal the brute-force oracle's, bit for bit."""
    r = np.random.default_rng(77)
    # (a) a soup of small triangles 4000 units away from the origin
    v, i = triangle_soup(12000, seed=21, extent=6.0, size=0.05)
    v["position"] += np.array([4000.0, -2500.0, 3000.0], np.float32)
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    c = np.array([4000.0, -2500.0, 3000.0])
    O, D = random_rays(30000, 31, c - 6, c + 6)
    compare_all(p, O, D, brute=True)
    # (b) an axis-aligned tessellated wall + floor (flat boxes, shared planes) and rays inside / along them
    wall, wt = scenes.displaced_grid(64, seed=3, extent=8.0)
    wall["position"][:, 1] = 0.0                                    # a perfectly flat floor at y = 0: every box is flat
    p = Pair(oracle, capi, gpu, [(wall, wt)], [(0, None)])
    n = 24000
    O = np.zeros((n, 4), np.float32); D = np.zeros((n, 4), np.float32)
    O[:, 0] = r.uniform(-4, 4, n); O[:, 2] = r.uniform(-4, 4, n)
    O[:, 1] = np.where(r.uniform(size=n) < 0.5, 0.0, r.uniform(-1e-6, 1e-6, n))       # on the floor plane or within a micron of it
    ang = r.uniform(0, 2 * np.pi, n)
    D[:, 0] = np.cos(ang); D[:, 2] = np.sin(ang)
    steep = 10.0 ** r.uniform(-9, -3, n) * np.where(r.uniform(size=n) < 0.5, 1.0, -1.0)
    D[:, 1] = np.where(r.uniform(size=n) < 0.25, 0.0, steep)        # a quarter exactly in the plane
    D[:, :3] /= np.linalg.norm(D[:, :3].astype(np.float64), axis=1, keepdims=True).astype(np.float32)
    O[:, 3] = 0.0; D[:, 3] = 1e38
    compare_all(p, O, D, brute=True)
    # (c) the same rays against a mesh with relief, origins ON vertices (ray origin exactly on box planes)
    hill, ht = scenes.displaced_grid(48, seed=5, extent=8.0)
    p = Pair(oracle, capi, gpu, [(hill, ht)], [(0, None)])
    pick = r.integers(0, hill.shape[0], n)
    O[:, :3] = hill["position"][pick]
    compare_all(p, O, D, brute=True)
    # (d) microscopic clusters seen from far away: nodes whose whole extent lies inside the culling margin, where even the
    # inverted interval of an UNUSED child slot passes the plane test and only its code (RT_NODE_NONE) keeps the walk out;
    # odd cluster sizes leave slots unused at every level
    parts = []
    for k, (cnt, size) in enumerate([(3, 1e-6), (5, 1e-7), (7, 3e-6), (11, 1e-8), (2, 1e-5), (37, 1e-6)]):
        cv, ci = triangle_soup(cnt, seed=200 + k, extent=size * 4, size=size)
        cv["position"] += r.uniform(-2, 2, 3).astype(np.float32)
        parts.append(cv)
    v = np.concatenate(parts)
    i = np.arange(v.shape[0], dtype=np.uint32).reshape(-1, 3)
    p = Pair(oracle, capi, gpu, [(v, i)], [(0, None)])
    tri = v["position"].reshape(-1, 3, 3)
    n = 20000
    O = np.zeros((n, 4), np.float32); D = np.zeros((n, 4), np.float32)
    target = tri[r.integers(0, tri.shape[0], n)]
    w = r.dirichlet((1, 1, 1), n).astype(np.float32)
    aim = (target * w[:, :, None]).sum(axis=1) + (r.normal(size=(n, 3)) * 1e-7).astype(np.float32)
    dirs = r.normal(size=(n, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    dist = 10.0 ** r.uniform(0, 4, n)                               # 1 .. 10^4 units away
    O[:, :3] = (aim - dirs * dist[:, None]).astype(np.float32)
    D[:, :3] = dirs.astype(np.float32)
    D[:, 3] = 1e38
    compare_all(p, O, D, brute=True)


def test_small_lds_stack_spills_to_global_rows():
    """The production kernels keep 24 stack rows per lane in LDS and continue in global memory beyond them
    (rt_trace_wave.h); real scenes rarely get there, so the 6-row instantiation (option lds_stack_rows=6) re-runs
    the deep-tree parity tests with most rays spilling."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RT_DEBUG_OPTIONS="lds_stack_rows=6")
    sel = ["test_gpu_trace.py::test_soup_with_tmin_tmax_windows", "test_gpu_trace.py::test_instanced_two_level",
           "test_gpu_pipeline.py::test_instanced_scene_materials_and_misses", "test_gpu_pipeline.py::test_material_types_and_depth_limits"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + [os.path.join(here, s) for s in sel],
                       env=env, cwd=os.path.dirname(here), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
