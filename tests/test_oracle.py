"""CPU-only tests of the oracle itself: it must be pinned before it is trusted.

The reference ships no golden vectors; what CAN be pinned is pinned
here: the integer RNG known-answer values derived from the HLSL text, the accuracy of
the engine-defined transcendental kernels against float64, sampler invariants, the
OBJ fixture, LBVH structural invariants, BVH == brute force, and the committed golden
image (guards the oracle against regressions between rounds)."""
import json
import os

import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import ANY, CORNELL_OBJ, CULL, GOLDEN, assert_hits_equal, primary_rays, random_rays, random_xforms, triangle_soup


def test_rng_known_answers(oracle):
    k = json.load(open(os.path.join(GOLDEN, "rng_kats.json")))
    for key, val in k["init_rand"].items():
        a, b = (int(x) for x in key.split(","))
        assert oracle.init_rand(a, b) == val, key
    for item in k["next_rand"]:
        s = item["seed"]
        for want_s, want_f in item["seq"]:
            s, f = oracle.next_rand(s)
            assert s == want_s and f == want_f
    # the six values SURVEY.md 8(c) lists, verbatim
    assert oracle.init_rand(0, 0) == 0x741c187d and oracle.init_rand(12345, 7) == 0xa025d928
    assert oracle.init_rand(2073599, 1) == 0x030170e4 and oracle.init_rand(130815, 1023) == 0x1f9db247


def ulps(got, ref64):
    ref32 = ref64.astype(np.float32)
    return np.abs(got.astype(np.float64) - ref64) / np.spacing(np.abs(ref32)).astype(np.float64)


def test_transcendental_kernels_accuracy(oracle):
    r = np.random.default_rng(0)
    x = r.uniform(0, 2 * np.pi, 1 << 18).astype(np.float32)
    assert np.abs(oracle.math("sin", x) - np.sin(x.astype(np.float64))).max() < 1.5e-7
    assert np.abs(oracle.math("cos", x) - np.cos(x.astype(np.float64))).max() < 1.5e-7
    x = r.uniform(-85, 88, 1 << 18).astype(np.float32)
    assert ulps(oracle.math("exp", x), np.exp(x.astype(np.float64))).max() < 1.5
    x = np.exp(r.uniform(-80, 80, 1 << 18)).astype(np.float32)
    assert ulps(oracle.math("log", x), np.log(x.astype(np.float64))).max() < 1.5
    x = r.uniform(0, 1, 1 << 18).astype(np.float32)
    y = np.full_like(x, 403.4288)
    ref = np.power(x.astype(np.float64), y.astype(np.float64))
    m = ref > 1e-30
    assert (np.abs(oracle.math("pow", x, y)[m] - ref[m]) / ref[m]).max() < 2e-5       # |y ln x| * 2^-24
    assert np.array_equal(oracle.math("pow", np.zeros(3, np.float32), np.array([5, 1, 0.1], np.float32)), np.zeros(3, np.float32))
    assert oracle.math("exp", np.array([-90.0], np.float32))[0] == 0.0
    assert np.isinf(oracle.math("exp", np.array([89.0], np.float32))[0])
    one = oracle.math("pow", np.array([1.0], np.float32), np.array([403.0], np.float32))[0]
    assert one == 1.0
    # sqrt / divide are the IEEE correctly rounded forms
    x = np.exp(r.uniform(-80, 80, 1 << 16)).astype(np.float32)
    assert np.array_equal(oracle.math("sqrt", x), np.sqrt(x))
    with np.errstate(over="ignore"):
        assert np.array_equal(oracle.math("div", x, x[::-1].copy()), x / x[::-1])


def test_sampler_invariants(oracle):
    r = np.random.default_rng(1)
    n = 20000
    seeds = r.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    v = r.standard_normal((n, 3))
    v = (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)
    for kind in ("cos", "uniform"):
        d, _, so = oracle.sample(kind, seeds, v)
        # the reference never normalises getPerpendicularVector's cross product (RaytracingUtils.hlsli:49-56,
        # 64-65), so the tangent frame is shorter than 1 and samples are NOT unit length: kept as is
        ln = np.linalg.norm(d, axis=1)
        assert (ln <= 1.0 + 1e-5).all() and (ln > 0.57).all()
        assert ((d * v).sum(1) > -1e-6).all()                      # upper hemisphere
        s = seeds.astype(np.uint64)
        for _ in range(2):                                         # exactly two LCG draws
            s = (1664525 * s + 1013904223) & 0xFFFFFFFF
        assert np.array_equal(so, s.astype(np.uint32))
    d, pb, _ = oracle.sample("phong", seeds, v, 403.4288)
    assert (np.linalg.norm(d, axis=1) <= 1.0 + 1e-5).all()
    ok = pb[:, 0] > 0
    assert np.allclose(pb[ok, 1] / pb[ok, 0], 405.4288 / 404.4288, rtol=1e-6)   # brdf/pdf = (e+2)/(e+1)
    p, _, _ = oracle.sample("perp", seeds, v)
    assert np.abs((p * v).sum(1)).max() < 1e-6
    f = oracle.fresnel([0, 0, -1], [0, 0, 1], [0.58, 0.58, 0.58])
    assert np.allclose(f, 0.58)                                     # normal incidence -> f0
    f = oracle.fresnel([1, 0, 0], [0, 0, 1], [0.58, 0.58, 0.58])
    assert np.allclose(f, 1.0)                                      # grazing -> 1


def test_cube_sampling_conventions(oracle):
    size = 4
    faces = np.zeros((6, size, size, 4), np.float32)
    for f in range(6):
        faces[f, :, :, 0] = f                  # red = face id
    faces[0, :, :, 1] = np.arange(size)[None, :]       # green ramps along +u on face +X
    d = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    c = oracle.sample_cube(faces, d)
    assert np.array_equal(c[:, 0], np.arange(6, dtype=np.float32))             # D3D face order +X -X +Y -Y +Z -Z
    c = oracle.sample_cube(faces, np.array([[1, 0, -0.5], [1, 0, 0.5]], np.float32))
    assert c[0, 1] > c[1, 1]                                                    # on +X, u grows towards -z
    c = oracle.sample_cube(faces, np.array([[0, 0, 0], [np.nan, np.nan, np.nan], [np.nan, 1, 0]], np.float32))
    assert np.array_equal(c[:2], np.zeros((2, 3), np.float32))     # no major axis: defined as black
    assert np.isnan(c[2]).all()                                     # NaN coordinate on a valid face propagates (RayGen clamps it to 0)


def test_obj_fixture(oracle):
    v, i = oracle.obj_load(CORNELL_OBJ)
    assert i.shape == (34, 3) and v.shape == (68,)          # 34 triangles, joined (position, normal) pairs
    assert v["position"].min() >= -1.0013 and v["position"].max() <= 1.0
    n = np.cross(v["position"][i[:, 1]] - v["position"][i[:, 0]], v["position"][i[:, 2]] - v["position"][i[:, 0]])
    agree = (n * v["normal"][i[:, 0]]).sum(1)
    assert (agree > 0).all()      # every face is wound so that cross(e1,e2) follows its stored normal (front faces see the room)


def test_obj_reader_features(oracle, tmp_path):
    p = tmp_path / "poly.obj"
    p.write_text("# quad + negative indices + texture slots + no normals\n"
                 "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\n"
                 "f 1/1 2/1 3/1 4/1\n"
                 "v 0 0 1\nf -1 -4 -3\n")
    v, i = oracle.obj_load(str(p))
    assert i.tolist() == [[0, 1, 2], [0, 2, 3], [4, 1, 2]]          # fan triangulation, first-use vertex order
    assert np.allclose(v["normal"][3], [0, 0, 1])                   # generated smooth normal
    with pytest.raises(IOError):
        oracle.obj_load(str(tmp_path / "missing.obj"))


def check_tree(nodes, keys, parents, depth):
    n = keys.size
    assert nodes.size == 2 * n - 1
    assert (np.diff(keys.astype(np.uint64)) > 0).all() if n > 1 else True          # strictly ascending, unique
    leaves = nodes[n - 1:]
    assert (leaves["right"] == T.RT_LEAF).all()
    assert np.array_equal(leaves["left"], (keys & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    assert sorted(leaves["left"].tolist()) == list(range(n))
    if n == 1:
        return
    inner = nodes[: n - 1]
    kids = np.concatenate([inner["left"], inner["right"]])
    assert sorted(kids.tolist()) == list(range(1, 2 * n - 1))                        # every node but the root has one parent
    assert parents[0] == 0xFFFFFFFF
    assert np.array_equal(parents[inner["left"]], np.arange(n - 1)) and np.array_equal(parents[inner["right"]], np.arange(n - 1))
    lo = np.minimum(nodes["bmin"][inner["left"]], nodes["bmin"][inner["right"]])
    hi = np.maximum(nodes["bmax"][inner["left"]], nodes["bmax"][inner["right"]])
    assert np.array_equal(inner["bmin"], lo) and np.array_equal(inner["bmax"], hi)   # exact union
    d = np.zeros(2 * n - 1, np.int64)
    for k in range(n - 1, 2 * n - 1):
        c, dd = parents[k], 0
        while c != 0xFFFFFFFF:
            c = parents[c]; dd += 1
        d[k] = dd
    assert d.max() == depth


@pytest.mark.parametrize("n", [1, 2, 3, 7, 500])
def test_lbvh_invariants(oracle, n):
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(*triangle_soup(n, seed=n)))
    sc.build()
    check_tree(*sc.bvh(0))
    check_tree(*sc.bvh(-1))


def test_lbvh_duplicates_and_tlas(oracle):
    v, i = triangle_soup(40, seed=3)
    tri = v["position"].reshape(-1, 3, 3)
    tri[1:20] = tri[0]
    sc = oracle.Scene()
    m = sc.add_model(v, i)
    for x in random_xforms(9, seed=1):
        sc.add_instance(m, x)
    sc.add_instance(m, None)
    sc.build()
    check_tree(*sc.bvh(0))
    check_tree(*sc.bvh(-1))
    box, inv = sc.instance_info(9)
    assert np.array_equal(inv, T.IDENTITY_3X4)
    nodes = sc.bvh(0)[0]
    assert np.array_equal(box[:3], nodes["bmin"][0]) and np.array_equal(box[3:], nodes["bmax"][0])


def test_instance_world_box_is_the_box_of_the_transformed_vertices(oracle):
    """oracle_bvh.h scene_build: a transformed instance's world box is the exact fp32 box of its triangles' transformed vertices"""
    blob = scenes.blob_mesh(level=2)
    xf = random_xforms(6, seed=4)
    sc = oracle.Scene()
    m = sc.add_model(*blob)
    for x in xf:
        sc.add_instance(m, x)
    sc.build()
    p = blob[0]["position"][np.asarray(blob[1]).reshape(-1)].astype(np.float32)
    for k, x in enumerate(xf):
        a = x.reshape(3, 4)
        w = np.stack([((a[r, 0] * p[:, 0] + a[r, 1] * p[:, 1]) + a[r, 2] * p[:, 2]) + a[r, 3] for r in range(3)], axis=1)
        box, _ = sc.instance_info(k)
        assert np.array_equal(box, np.concatenate([w.min(axis=0), w.max(axis=0)]))


def test_bvh_equals_brute_force(oracle):
    blob = scenes.blob_mesh(level=2)
    soup = triangle_soup(300, seed=2, extent=1.5, size=0.4)
    xf = random_xforms(20, seed=5, spread=8.0)
    sc = oracle.Scene()
    a, b = sc.add_model(*blob), sc.add_model(*soup)
    for k in range(20):
        sc.add_instance(a if k % 2 else b, xf[k])
    sc.add_instance(a, None)
    sc.build()
    O, D = random_rays(40000, 6, [-10, -10, -10], [10, 10, 10])
    for flags in (0, CULL):
        assert_hits_equal(sc.trace(O, D, flags=flags, mode=1, nthreads=8), sc.trace(O, D, flags=flags, mode=0, nthreads=8), "bvh vs brute")
    assert_hits_equal(sc.trace(O, D, flags=ANY, mode=1, nthreads=8), sc.trace(O, D, flags=ANY, mode=0, nthreads=8), "any", closest=False)


def test_back_face_rule_and_range(oracle):
    """One triangle in the z=0 plane wound so that cross(e1,e2) = +z."""
    v = np.zeros(3, T.VERTEX)
    v["position"] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    v["normal"] = [0, 0, 1]
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(v, [[0, 1, 2]]))
    sc.build()
    O = np.array([[0.25, 0.25, 1, 0], [0.25, 0.25, -1, 0], [0.25, 0.25, 1, 0], [0.25, 0.25, 1, 1.0], [0.25, 0.5, 1, 0]], np.float32)
    D = np.array([[0, 0, -1, 1e38], [0, 0, 1, 1e38], [0, 0, -1, 1.0], [0, 0, -1, 1e38], [0, 0, -1, 1e38]], np.float32)
    h = sc.trace(O, D, flags=CULL)
    assert h["prim"].tolist() == [0, T.RT_NO_HIT, T.RT_NO_HIT, T.RT_NO_HIT, 0]    # front seen from +z; t<tmax, t>tmin exclusive
    assert h["t"][0] == 1.0 and h["u"][0] == 0.25 and h["v"][0] == 0.25 and h["v"][4] == 0.5   # (u,v) weight v1, v2
    h = sc.trace(O, D, flags=0)
    assert h["prim"].tolist() == [0, 0, T.RT_NO_HIT, T.RT_NO_HIT, 0]              # no culling: both sides


def test_golden_image_reproduced(oracle):
    g = np.load(os.path.join(GOLDEN, "cornell64_golden.npz"))
    v, i = oracle.obj_load(CORNELL_OBJ)
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(v, i))
    sc.build()
    acc = np.zeros((64, 64, 4), np.float32)
    mat = T.default_material()
    for f in range(4):
        acc, st = sc.render(mat, g["pfc"][f], 64, 64, accum=acc, env_constant=(0.5, 0.5, 0.5))
        assert np.array_equal(acc, g["images"][f])
    pf = np.frombuffer(g["pfc"][0].tobytes(), T.PER_FRAME_CONSTANTS)[0]
    O, D = primary_rays(pf, 64, 64)
    h = sc.trace(O, D, flags=CULL, mode=1)
    assert np.array_equal(h["prim"], g["prim"]) and np.array_equal(h["t"], g["t"])
    # brute-force renderer == BVH renderer
    a1, _ = sc.render(mat, g["pfc"][0], 64, 64, use_brute=True)
    assert np.array_equal(a1, g["images"][0])
    # ray budget of a fully hit pixel: 1 primary + 2 secondary + 6 shadow
    assert st["rays_secondary"] == 2 * st["primary_hits"]
    assert st["rays_shadow"] == 2 * st["primary_hits"] + 2 * st["secondary_hits"]


def test_tiles_and_sum_mode(oracle):
    g = np.load(os.path.join(GOLDEN, "cornell64_golden.npz"))
    v, i = oracle.obj_load(CORNELL_OBJ)
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(v, i))
    sc.build()
    mat = T.default_material()
    acc = np.zeros((64, 64, 4), np.float32)
    for ty in range(2):
        for tx in range(2):
            sc.render(mat, g["pfc"][0], 64, 64, accum=acc, env_constant=(0.5, 0.5, 0.5), tile=(tx * 32, ty * 32, tx * 32 + 32, ty * 32 + 32))
    assert np.array_equal(acc, g["images"][0])
    s = np.zeros((64, 64, 4), np.float32)
    for f in range(4):
        sc.render(mat, g["pfc"][f], 64, 64, accum=s, env_constant=(0.5, 0.5, 0.5), accum_mode=T.ACCUM_SUM)
    assert np.abs(s / 4 - g["images"][3]).max() < 1e-5


def test_oracle_selftest_under_sanitizers():
    """the whole oracle (OBJ, LBVH, brute force == BVH, progressive + realtime shading on two threads,
    denoiser, host update) in one program under ASan + UBSan and under TSan."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    odir = os.path.join(root, "oracle")
    subprocess.run(["make", "-C", odir, "selftest_asan", "selftest_tsan"], check=True, stdout=subprocess.DEVNULL, timeout=600)
    obj = os.path.join(root, "tests", "golden", "cornell.obj")
    for exe in ("selftest_asan", "selftest_tsan"):
        r = subprocess.run([os.path.join(odir, exe), obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0 and "oracle selftest ok" in r.stdout, r.stdout[-3000:]


def test_structures_fixture_reproduced(oracle):
    """cornell64_structures.npz: canonical LBVH arrays, realtime AOVs and the denoised composite."""
    g = np.load(os.path.join(GOLDEN, "cornell64_structures.npz"))
    v, i = oracle.obj_load(CORNELL_OBJ)
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(v, i))
    sc.build()
    nodes, keys, parents, depth = sc.bvh(0)
    assert np.ascontiguousarray(nodes).tobytes() == g["bvh_nodes"].tobytes()
    assert np.array_equal(keys, g["bvh_keys"]) and np.array_equal(parents, g["bvh_parents"]) and depth == int(g["bvh_depth"])
    d, ind, _ = sc.render_realtime(T.default_material(), g["realtime_pfc"], 64, 64, env_constant=(0.5, 0.5, 0.5))
    assert np.array_equal(d, g["direct"]) and np.array_equal(ind, g["indirect"])
    prm = np.frombuffer(g["denoise_params"].tobytes(), oracle.DENOISE_PARAMS)[0]
    _, out = oracle.denoise(g["direct"], g["indirect"], prm)
    assert np.array_equal(out, g["denoised"])


def test_round_to_half_is_ieee(oracle):
    """the oracle's binary32 -> binary16 -> binary32 (the reference's RGBA16F accumulation texture, src/DXRExperimentsApp.cpp:28) against numpy's
    float16: every half, every midpoint between two halves and its fp32 neighbours, subnormals, overflow; toward-zero: representable, never larger in
    magnitude, and tight"""
    r = np.random.default_rng(1)
    h = np.arange(0, 0x7c00, dtype=np.uint16).view(np.float16).astype(np.float32)
    mid = ((h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2).astype(np.float32)
    x = np.concatenate([r.uniform(-70000, 70000, 100000), r.normal(size=100000) * 1e-5, r.normal(size=100000) * 1e-7, r.normal(size=100000),
                        [0, -0.0, 65504, 65519.99, 65520, 1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 3e-8, np.inf, -np.inf]]).astype(np.float32)
    x = np.concatenate([x, h, mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)), -mid])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(oracle.round_to_half(x, True).view(np.uint32), want.view(np.uint32))
    t = oracle.round_to_half(x, False)
    fin = np.isfinite(x)
    th = t.astype(np.float16)
    assert np.array_equal(th.astype(np.float32)[fin], t[fin]) and (np.abs(t[fin]) <= np.abs(x[fin])).all()
    with np.errstate(over="ignore"):
        up = np.nextafter(np.abs(th), np.float16(np.inf)).astype(np.float32)
    assert ((up[fin] > np.abs(x[fin])) | (np.abs(t[fin]) == 65504)).all()
