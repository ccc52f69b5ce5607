"""Regenerates the committed fixtures under tests/golden/.  Run in the authoring
container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fixtures.py

Fixtures are DATA (inputs and expected outputs), never reference source:
  cornell.obj            the Cornell mesh of the reference's assets/models/cornell.obj
                         (34 triangles), re-emitted by this repo's own OBJ writer after
                         ingestion (joined vertices, one v/vn per vertex, faces a//a).
  rng_kats.json          integer known-answer values of initRand / nextRand derived from
                         the HLSL text (assets/shaders/RaytracingUtils.hlsli:26-45), as
                         listed in SURVEY.md 8(c), re-derived here by an independent
                         pure-Python evaluation.
  cornell64_golden.npz   oracle output for config C1 at 64x64: the four per-frame
                         constant buffers, primary hit ids and the fp32 accumulation
                         image after each of 4 frames.
  host_update_golden.npz the 188-byte constant buffers the host update logic produces
                         for a fixed camera / seed (6 frames incl. a camera move).
  cornell64_structures.npz  the canonical LBVH of the Cornell mesh (nodes, keys, parents, depth),
                         the two realtime AOVs of one 64x64 frame and their denoised composite.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import pyoracle as O                      # noqa: E402
from dxrexperiments_amd import rtypes as T, scenes    # noqa: E402

REF_OBJ = "/root/reference/assets/models/cornell.obj"


def py_init_rand(v0, v1):
    """Pure-Python TEA, written from the HLSL text, independent of the C++ oracle."""
    M = 0xFFFFFFFF
    s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & M
        v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c) & M ^ ((v1 + s0) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
        v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d) & M ^ ((v0 + s0) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
    return v0


def py_next_rand(s):
    s = (1664525 * s + 1013904223) & 0xFFFFFFFF
    return s, (s & 0x00FFFFFF) / float(0x01000000)


def main():
    # --- cornell.obj -------------------------------------------------------------
    v, tri = O.obj_load(REF_OBJ)
    assert tri.shape == (34, 3)
    out_obj = os.path.join(HERE, "cornell.obj")
    scenes.write_obj(out_obj, v, tri,
                     header="Cornell box, 34 triangles: geometry of philcn/DXRExperiments assets/models/cornell.obj\n"
                            "re-emitted by dxrexperiments_amd.scenes.write_obj (tests/golden/make_fixtures.py)")
    v2, tri2 = O.obj_load(out_obj)
    assert np.array_equal(v2, v) and np.array_equal(tri2, tri)

    # --- RNG KATs -------------------------------------------------------------------
    survey = {"0,0": 0x741c187d, "1,0": 0x8da6b311, "0,1": 0x70d3aef1, "12345,7": 0xa025d928,
              "2073599,1": 0x030170e4, "130815,1023": 0x1f9db247}
    kats = {"init_rand": {}, "next_rand": []}
    for k, val in survey.items():
        a, b = (int(x) for x in k.split(","))
        assert py_init_rand(a, b) == val, k
        kats["init_rand"][k] = val
    for a, b in ((7, 9), (65535, 3), (4294967295, 4294967295), (1920 * 1080 - 1, 1024)):
        kats["init_rand"]["%d,%d" % (a, b)] = py_init_rand(a, b)
    for seed in (0x741c187d, 0xa025d928, 0, 0xFFFFFFFF):
        s = seed
        seq = []
        for _ in range(4):
            s, f = py_next_rand(s)
            seq.append([s, f])
        kats["next_rand"].append({"seed": seed, "seq": seq})
    assert kats["next_rand"][0]["seq"][0] == [0xb7d2ffb8, 0.8242144584655762]
    assert kats["next_rand"][1]["seq"][1] == [0xb316e49a, 0.08942568302154541]
    with open(os.path.join(HERE, "rng_kats.json"), "w") as f:
        json.dump(kats, f, indent=1, sort_keys=True)

    # --- Cornell 64x64 golden ------------------------------------------------------
    W = H = 64
    c = scenes.cornell_camera()
    cam = np.array([*c["eye"], *c["at"], *c["up"], c["fov"], W / H], np.float32)
    sc = O.Scene()
    sc.add_instance(sc.add_model(v, tri))
    sc.build()
    host = O.Progressive(1234)
    mat = T.default_material()
    acc = np.zeros((H, W, 4), np.float32)
    pfcs, images = [], []
    for frame in range(4):
        pfc = host.update(cam, 0.0, frame + 1, W, H)
        acc, _ = sc.render(mat, pfc, W, H, accum=acc, env_constant=(0.5, 0.5, 0.5))
        pfcs.append(pfc.copy())
        images.append(acc.copy())
    # primary hits of frame 0
    pf0 = np.frombuffer(pfcs[0].tobytes(), T.PER_FRAME_CONSTANTS)[0]
    o, d = primary_rays(pf0, W, H)
    h = sc.trace(o, d, flags=T.RAY_FLAG_CULL_BACK_FACING_TRIANGLES, mode=0)
    np.savez_compressed(os.path.join(HERE, "cornell64_golden.npz"),
                        pfc=np.stack(pfcs), images=np.stack(images), camera=cam,
                        prim=h["prim"], inst=h["inst"], t=h["t"], u=h["u"], v=h["v"])

    # --- host update golden -------------------------------------------------------
    host = O.Progressive(99)
    c2 = scenes.sponza_camera()
    camA = np.array([*c2["eye"], *c2["at"], *c2["up"], c2["fov"], 1920 / 1080], np.float32)
    camB = camA.copy()
    camB[0] += 0.5
    seq = []
    for i, cm in enumerate((camA, camA, camA, camB, camB, camA)):
        seq.append(host.update(cm, 0.5 * i, 10 + i, 1920, 1080).copy())
    np.savez_compressed(os.path.join(HERE, "host_update_golden.npz"), cams=np.stack([camA, camA, camA, camB, camB, camA]),
                        pfc=np.stack(seq), seed=99)
    # --- Cornell BVH arrays, realtime AOVs, denoised image (the GPU box checks these without the oracle) -----
    nodes, keys, parents, depth = sc.bvh(0)
    # RealtimeRaytracingPipeline::update = the progressive update with accumCount 0 and options {environmentStrength 1}
    rpfc = np.frombuffer(bytearray(O.Progressive(77).update(cam, 0.0, 1, W, H).tobytes()), T.PER_FRAME_CONSTANTS).copy()
    rpfc["cameraParams"]["accumCount"] = 0
    opt = np.zeros((), T.DEBUG_OPTIONS)
    opt["environmentStrength"] = 1.0
    rpfc["options"] = opt
    direct, indirect, rst = sc.render_realtime(mat, rpfc, W, H, env_constant=(0.5, 0.5, 0.5))
    dparams = np.zeros((), O.DENOISE_PARAMS)
    dparams["exposure"], dparams["gamma"], dparams["tonemap"], dparams["gammaCorrect"], dparams["maxKernelSize"] = 1.0, 2.2, 1, 1, 12
    _, denoised = O.denoise(direct, indirect, dparams)
    np.savez_compressed(os.path.join(HERE, "cornell64_structures.npz"),
                        bvh_nodes=np.frombuffer(np.ascontiguousarray(nodes).tobytes(), np.uint8), bvh_keys=keys, bvh_parents=parents, bvh_depth=depth,
                        realtime_pfc=rpfc, direct=direct, indirect=indirect, denoise_params=np.frombuffer(dparams.tobytes(), np.uint8),
                        denoised=denoised)
    print("fixtures written to", HERE)


def primary_rays(pf, W, H):
    """numpy restatement of RayGen's ray set-up (fp32, same operation order)."""
    cp = pf["cameraParams"]
    f = np.float32
    xs, ys = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32), indexing="xy")
    dx = ((xs + f(0.5)) / f(W)) * f(2.0) - f(1.0)
    dy = ((ys + f(0.5)) / f(H)) * f(2.0) - f(1.0)
    U, V, Wv = cp["U"][:3], cp["V"][:3], cp["W"][:3]
    d = dx[..., None] * U + (-dy)[..., None] * V
    d = d + Wv
    dot = d[..., 0] * d[..., 0]
    dot = dot + d[..., 1] * d[..., 1]
    dot = dot + d[..., 2] * d[..., 2]
    r = f(1.0) / np.sqrt(dot)
    d = d * r[..., None]
    n = W * H
    o = np.zeros((n, 4), np.float32)
    j = cp["jitters"] * f(30.0)
    o[:, 0] = cp["worldEyePos"][0] + j[0]
    o[:, 1] = cp["worldEyePos"][1] + j[1]
    o[:, 2] = cp["worldEyePos"][2] + f(0.0)
    dd = np.zeros((n, 4), np.float32)
    dd[:, :3] = d.reshape(-1, 3)
    dd[:, 3] = f(1.0e38)
    return o, dd


if __name__ == "__main__":
    main()
