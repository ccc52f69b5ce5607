"""S2 (TraceRay semantics) held to the float64 geometric truth (oracle/truth64.h: Moller-Trumbore over every instance x triangle, no box of any
kind) -- CPU half: the oracle's TraceRay, which the kernels reproduce bit for bit (tests/test_gpu_s2_truth.py holds the kernels to the same
bounds through rt_trace_batch).  The reference gets DXR semantics from TraceRay (assets/shaders/ProgressiveRaytracing.hlsl:34,53,
RaytracingCommon.hlsli:84-96; opaque triangles, libs/DXRFramework/Helpers/BottomLevelASGenerator.h:127): a ray through a triangle hits it.  The
engine's fp32 definition may differ from geometry only as often as tests/golden/s2_bounds.json says, and never more often than its earlier
forms did."""
import json
import os

import numpy as np
import pytest

import s2_truth as S
from util import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BOUNDS = json.load(open(os.path.join(GOLDEN, "s2_bounds.json")))
CORES = max(1, len(os.sched_getaffinity(0)))


def test_the_rule_has_not_changed_without_new_bounds():
    """the code between the S2-RULE marks (slab test, Moller-Trumbore, box clause, tie-break, split rule; oracle and kernels) hashes to what
    the bounds were measured with: tools/s2_truth_report.py --images 96 54 --write after any change to it"""
    assert S.rule_hash(ROOT) == BOUNDS["rule_hash"]


def test_truth64_is_what_it_says(oracle):
    """the truth against numpy float64 on a scene small enough to do by hand: one triangle, rays through its inside, its edge, beside it,
    from behind, and windows that exclude the hit"""
    from dxrexperiments_amd import rtypes as T
    v = np.zeros(3, T.VERTEX)
    v["position"] = [[0, 0, 0], [1, 0, 0], [0, 1, 0]]
    sc = S.oracle_scene(oracle, [(v, np.array([[0, 1, 2]], np.uint32))], [(0, None)])
    O = np.array([[0.25, 0.25, -1, 0], [0.5, 0.5, -1, 0], [0.8, 0.8, -1, 0], [0.25, 0.25, 1, 0], [0.25, 0.25, -1, 0], [0.25, 0.25, -1, 1.5]], np.float32)
    D = np.array([[0, 0, 1, 1e30], [0, 0, 1, 1e30], [0, 0, 1, 1e30], [0, 0, -1, 1e30], [0, 0, 1, 1.0], [0, 0, 1, 1e30]], np.float32)
    h = sc.truth64(O, D, 0)
    assert h["inst"].tolist() == [0, 0, 0xFFFFFFFF, 0, 0xFFFFFFFF, 0xFFFFFFFF]         # inside, ON the edge (inclusive), outside, back, t == tmax, t < tmin
    assert np.allclose(h["t"][[0, 1, 3]], 1.0) and np.allclose(h["u"][0], 0.25) and np.allclose(h["v"][0], 0.25)
    # front face <=> det > 0: e1 x e2 = +z, so a ray travelling along -z sees the front (det = -d . (e1 x e2) > 0)
    hc = sc.truth64(O, D, S.CULL)
    assert hc["inst"].tolist() == [0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF, 0, 0xFFFFFFFF, 0xFFFFFFFF]
    assert (sc.truth64(O, D, S.ANY)["inst"] != 0xFFFFFFFF).tolist() == [True, True, False, True, False, False]
    # the engine's own definition agrees on all of these
    e = sc.trace(O, D, 0)
    assert e["inst"].tolist() == h["inst"].tolist()


@pytest.mark.parametrize("name", S.SCENES)
def test_oracle_trace_ray_against_the_geometric_truth(oracle, name):
    models, instances, aim = S.scene_models(name)
    models = S.load_arrays(oracle, models)
    sets = S.ray_sets(models, instances, aim, BOUNDS["rays_per_set"], seed=BOUNDS["seed"])
    sc = S.oracle_scene(oracle, models, instances)
    m = S.measure(lambda O, D, f: sc.trace(O, D, f, mode=1, nthreads=CORES), lambda O, D, f: sc.truth64(O, D, f, nthreads=CORES), sets)
    for sname in m:
        for mode, c in m[sname].items():
            b = BOUNDS["scenes"][name][sname][mode]
            assert c["rays"] == b["rays"]
            assert c["lost"] <= b["lost"] and c["phantom"] <= b["phantom"], (name, sname, mode, c, b)
            # ... and no leakier than the rule ever was: whole-AABB validation (rounds 1 - 4), reference boxes over [tmin, t] (round 5)
            for earlier in ("rounds_1_4", "round_5"):
                e = BOUNDS["earlier_rules"][earlier].get(name)          # (scenes added after the earlier rules were measured have no entry)
                if e is not None:
                    assert c["lost"] <= e[sname][mode]["lost"], (name, sname, mode, earlier, c, e[sname][mode])
    # brute force over the engine's own rule says the same as its traversal (the exactness argument of DESIGN.md section 2 on these very rays)
    O, D, _ = sets["aimed"]
    a, b = sc.trace(O[:4000], D[:4000], 0, mode=0, nthreads=CORES), sc.trace(O[:4000], D[:4000], 0, mode=1, nthreads=CORES)
    for k in ("t", "u", "v", "prim", "inst"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("name", ("atrium", "stadium"))
def test_frames_traced_by_the_truth(oracle, name):
    """image level: a frame of C2's scene / of the stress scene rendered by the oracle with the float64 truth as its tracer against the same
    frame under the engine's rule.  north_star's tolerance is 1e-5 RMS: C2's scene is inside it; the stress scene's frame holds a handful of
    pixels where fp32 Moller-Trumbore itself (not the box clause) differs from float64 on a grazing sliver -- the number is stated, and bounded."""
    W, H = S.BOUND_FRAME
    r = S.frame_rms(oracle, name, W, H, CORES)
    b = BOUNDS["frames"][name]
    assert r["pixels_off_by_more_than_1e_4"] <= b["pixels_off_by_more_than_1e_4"], (r, b)
    assert r["rms"] <= b["rms"] * 1.0001 + 1e-12 and r["rms_of_the_rest"] <= 1e-5, (r, b)
    if name == "atrium":
        assert r["rms"] <= 1e-5
