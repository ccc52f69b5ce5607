"""S2 held to the float64 geometric truth -- GPU half: the production kernels through the C ABI's rt_trace_batch on the ray sets of
tests/test_s2_truth.py, against oracle/truth64.h (no box of any kind), within the bounds of tests/golden/s2_bounds.json; and equal to the oracle
bit for bit on the same rays, so that the CPU half's numbers ARE the kernels' numbers."""
import json
import os

import numpy as np
import pytest

import s2_truth as S
from util import GOLDEN, Pair, assert_hits_equal

pytestmark = pytest.mark.gpu
BOUNDS = json.load(open(os.path.join(GOLDEN, "s2_bounds.json")))
CORES = max(1, len(os.sched_getaffinity(0)))


@pytest.mark.parametrize("name", S.SCENES)
def test_kernels_against_the_geometric_truth(gpu, oracle, capi, name):
    models, instances, aim = S.scene_models(name)
    models = S.load_arrays(oracle, models)
    sets = S.ray_sets(models, instances, aim, BOUNDS["rays_per_set"], seed=BOUNDS["seed"])
    p = Pair(oracle, capi, gpu, models, instances)
    m = S.measure(lambda O, D, f: p.g.trace(O, D, flags=f), lambda O, D, f: p.o.truth64(O, D, f, nthreads=CORES), sets)
    for sname in m:
        for mode, c in m[sname].items():
            b = BOUNDS["scenes"][name][sname][mode]
            assert c["lost"] <= b["lost"] and c["phantom"] <= b["phantom"], (name, sname, mode, c, b)
    for sname, (O, D, _) in sets.items():
        for mode, flags in S.MODES:
            assert_hits_equal(p.g.trace(O, D, flags=flags), p.o.trace(O, D, flags, mode=1, nthreads=CORES), "%s %s %s" % (name, sname, mode), closest=flags != S.ANY)
            if flags != S.ANY:
                assert_hits_equal(p.g.trace(O, D, flags=flags, canonical=True), p.o.trace(O, D, flags, mode=1, nthreads=CORES), "canonical %s %s %s" % (name, sname, mode))
