"""S2 (TraceRay semantics) against the float64 geometric truth: scenes, ray sets and the comparison, shared by tests/test_s2_truth.py (CPU: the
oracle), tests/test_gpu_s2_truth.py (the kernels through rt_trace_batch) and tools/s2_truth_report.py (the table in DESIGN.md section 2).

The truth (oracle/truth64.h) is float64 Moller-Trumbore over every instance x triangle with no box of any kind.  The engine's TraceRay is fp32
with a candidate rule; where the two name different primitives the case is one of
    tie      both hit, |t - t_truth| < 1e-4 t_truth: an edge shared by two triangles, coplanar overlap -- not a miss;
    lost     the truth hits at t_truth and the engine reports nothing, or something farther: the ray went THROUGH a triangle (a light leak);
    phantom  the engine reports a hit nearer than the truth's, or a hit where the truth has none: fp32 accepted a point just outside an edge.
"""
import numpy as np

from dxrexperiments_amd import scenes
from util import CORNELL_OBJ, random_xforms, sliver_soup

TIE = 1e-4
BOUND_RAYS, BOUND_SEED = 20000, 7            # the ray sets the committed bounds (tests/golden/s2_bounds.json) were measured on
BOUND_FRAME = (96, 54)


def scene_models(name):
    """-> (models [(verts, idx)], instances [(model, xform or None)], aim: indices of the triangles of model 0 rays are aimed at (None: all))"""
    if name == "cornell":
        return [CORNELL_OBJ], [(0, None)], None
    if name == "atrium":                         # C2's scene at a third of its detail (29 k triangles: the truth is a brute force)
        return [scenes.sponza_class(seed=42, detail=0.33)], [(0, None)], None
    if name == "instances":                      # C4 in small: rotated, scaled, shifted instances of a curved mesh
        return [scenes.blob_mesh(seed=3, level=2)], [(0, x) for x in random_xforms(24, seed=11, spread=6.0)], None
    if name == "stadium_slivers":                # the stress scene's cables and slats (20:1 ... 2000:1 slivers) in their hall
        v, t = scenes.stadium_class(seed=5, parts=("hall", "cables", "slats"))
        vh, th = scenes.stadium_class(seed=5, parts=("hall",))
        return [(v, t)], [(0, None)], np.arange(len(th), len(t))
    if name == "terrain":                        # C5's kind of mesh in small: a displaced grid (20 k triangles; the bench's has 10 M)
        return [scenes.displaced_grid(100, seed=7)], [(0, None)], None
    if name == "sliver_soup":
        v, t = sliver_soup(600, seed=77)
        return [(v, t)], [(0, None)], np.arange(600)
    raise KeyError(name)


SCENES = ("cornell", "atrium", "instances", "terrain", "stadium_slivers", "sliver_soup")


def load_arrays(orc, models):
    """file paths -> arrays through the ORACLE's OBJ reader (CPU tests have no product library to parse with)"""
    return [orc.obj_load(m) if isinstance(m, str) else m for m in models]


def oracle_scene(orc, models, instances):
    sc = orc.Scene()
    for v, t in models:
        sc.add_model(v, t)
    for mi, x in instances:
        sc.add_instance(mi, x)
    sc.build()
    return sc


def world_triangles(models, instances):
    """[n, 3, 3] float64 world-space corners of every instance's triangles + (instance, primitive) of each"""
    P, who = [], []
    for ii, (mi, x) in enumerate(instances):
        v, t = models[mi]
        p = v["position"][np.asarray(t).reshape(-1, 3)].astype(np.float64)
        if x is not None:
            m = np.asarray(x, np.float64).reshape(3, 4)
            p = p @ m[:, :3].T + m[:, 3]
        P.append(p)
        who.append(np.stack([np.full(len(p), ii), np.arange(len(p))], 1))
    return np.concatenate(P), np.concatenate(who)


def ray_sets(models, instances, aim, n, seed):
    """-> {"aimed": (O, D, dist), "random": (O, D, None)}: n rays aimed at random points of the chosen triangles from random origins in the
    scene's box (every one passes through its triangle), and n random rays through the box"""
    r = np.random.default_rng(seed)
    P, who = world_triangles(models, instances)
    lo, hi = P.reshape(-1, 3).min(0), P.reshape(-1, 3).max(0)
    cand = np.arange(len(P)) if aim is None else np.nonzero((who[:, 0] == 0) & np.isin(who[:, 1], aim))[0]
    pick = cand[r.integers(0, len(cand), n)]
    b = r.uniform(0, 1, (n, 2))
    flip = b.sum(1) > 1
    b[flip] = 1 - b[flip]
    target = P[pick, 0] + b[:, :1] * (P[pick, 1] - P[pick, 0]) + b[:, 1:] * (P[pick, 2] - P[pick, 0])
    origin = r.uniform(lo + 0.05 * (hi - lo), hi - 0.05 * (hi - lo), (n, 3))
    d = target - origin
    dist = np.linalg.norm(d, axis=1)
    d /= dist[:, None]
    O = np.concatenate([origin, np.full((n, 1), 1e-3)], 1).astype(np.float32)
    D = np.concatenate([d, np.full((n, 1), 1e30)], 1).astype(np.float32)
    o2 = r.uniform(lo, hi, (n, 3))
    d2 = r.normal(size=(n, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    O2 = np.concatenate([o2, np.full((n, 1), 1e-3)], 1).astype(np.float32)
    D2 = np.concatenate([d2, np.full((n, 1), 1e30)], 1).astype(np.float32)
    return {"aimed": (O, D, dist), "random": (O2, D2, None)}


def classify(engine, truth):
    """engine: dict t (fp32, -1 miss), prim, inst; truth: dict t (float64, -1 miss), prim, inst  ->  counts"""
    te, tt = engine["t"].astype(np.float64), truth["t"]
    he, ht = engine["inst"] != 0xFFFFFFFF, truth["inst"] != 0xFFFFFFFF
    same = (he == ht) & (~he | ((engine["prim"] == truth["prim"]) & (engine["inst"] == truth["inst"])))
    tie = ~same & he & ht & (np.abs(te - tt) < TIE * tt)
    rest = ~same & ~tie
    lost = rest & ht & (~he | (te > tt))
    phantom = rest & ~lost
    return dict(rays=int(len(te)), same=int(same.sum()), tie=int(tie.sum()), lost=int(lost.sum()), phantom=int(phantom.sum()),
                lost_idx=np.nonzero(lost)[0], phantom_idx=np.nonzero(phantom)[0])


def classify_any(engine_hit, truth):
    ht = truth["inst"] != 0xFFFFFFFF
    return dict(rays=int(len(ht)), lost=int((ht & ~engine_hit).sum()), phantom=int((~ht & engine_hit).sum()))


CULL, ANY = 0x10, 0x4 | 0x8
MODES = (("closest", 0), ("closest_cull", CULL), ("any_hit", ANY))


def measure(trace, truth, sets):
    """trace(O, D, flags) -> engine hits; truth(O, D, flags) -> truth hits  ->  {set: {mode: {rays, tie, lost, phantom}}}"""
    out = {}
    for sname, (O, D, _) in sets.items():
        out[sname] = {}
        for mname, flags in MODES:
            en, tr = trace(O, D, flags), truth(O, D, flags)
            if flags == ANY:
                c = classify_any(en["inst"] != 0xFFFFFFFF, tr)
            else:
                c = classify(en, tr)
            out[sname][mname] = {k: c[k] for k in ("rays", "tie", "lost", "phantom") if k in c}
    return out


# ---- the rule's text: a change to it without new bounds must fail a test -------------------------------------------------------------
RULE_FILES = ("oracle/oracle_bvh.h", "dxrexperiments_amd/csrc/rt_refs.h", "dxrexperiments_amd/csrc/rt_trace_device.h")


def rule_hash(root):
    """sha256 over the code between the S2-RULE-BEGIN / S2-RULE-END marks of the three files that state the rule -- comments and white space
    removed, so that only a change of the arithmetic counts"""
    import hashlib
    import os
    import re
    h = hashlib.sha256()
    for f in RULE_FILES:
        text = open(os.path.join(root, f)).read()
        parts = re.findall(r"S2-RULE-BEGIN.*?\n(.*?)(?://|/\*) S2-RULE-END", text, flags=re.S)
        assert parts, f
        for part in parts:
            part = re.sub(r"/\*.*?\*/", "", part, flags=re.S)
            part = re.sub(r"//[^\n]*", "", part)
            h.update(re.sub(r"\s+", "", part).encode())
    return h.hexdigest()


# ---- image level: a frame traced by the truth against the frame traced by the engine's rule ----------------------------------------------
def frame_setup(orc, name, W, H):
    """-> (models, instances, material, env, pfc) of a small frame of the atrium (C2's scene and camera) or the stress scene"""
    from dxrexperiments_amd import rtypes as T
    from util import cam_array
    if name == "atrium":
        models, cam = [scenes.sponza_class(seed=42)], scenes.sponza_camera()
    else:
        models, cam = [scenes.stadium_class(seed=5)], scenes.stadium_camera()
    mat = T.default_material()
    mat["type"] = 1
    mat["roughness"] = 0.4
    host = orc.Progressive(77)
    pfc = host.update(cam_array(cam, W / H), 0.0, 1, W, H)
    return models, [(0, None)], mat, scenes.sky_cubemap(32), pfc


def frame_rms(orc, name, W, H, nthreads):
    models, instances, mat, env, pfc = frame_setup(orc, name, W, H)
    sc = oracle_scene(orc, models, instances)
    a, _ = sc.render(mat, pfc, W, H, env_faces=env, nthreads=nthreads)
    b, _ = sc.render(mat, pfc, W, H, env_faces=env, nthreads=nthreads, use_brute=2)
    d = (a[..., :3].astype(np.float64) - b[..., :3].astype(np.float64))
    differing = int((np.abs(d).max(axis=2) > 1e-4).sum())
    return dict(width=W, height=H, rms=float(np.sqrt((d * d).mean())), pixels_off_by_more_than_1e_4=differing,
                rms_of_the_rest=float(np.sqrt((d[np.abs(d).max(axis=2) <= 1e-4] ** 2).mean())))
