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
  susanne.obj            the mesh of the reference's assets/models/susanne.obj (968 triangles, v/vt/vn
                         faces), re-emitted like cornell.obj.
  reference_assets.json  what an INDEPENDENT reading of the reference-held assets gives (numpy / pure
                         Python written here, sharing no code with the product or the oracle): for
                         assets/textures/CathedralRadiance.dds the header fields, the sha256 of the mip-0
                         fp32 decode and 64 probe texels; for susanne.obj and cornell.obj the sha256 of the
                         ingested vertex and index arrays under the ordering rules DESIGN.md section 2 defines.
  cathedral32.npz        a 32x32-per-face box-filtered down-sample of CathedralRadiance.dds (fp32), 1024
                         directions with the oracle's sampleEnvironment values on the FULL-SIZE map in both
                         filter modes, and a 64x64 Cornell frame lit by the down-sampled map (seamless filter).
  denoise_mock.npz       256x144 crops (RGBA8) of assets/textures/DirectLighting.PNG and IndirectSpecular.PNG,
                         the inputs of the reference's own denoiser test mode (DenoiseCompositor::loadResources
                         with loadMockResources, src/DenoiseCompositor.cpp:52-68), and the oracle's composite of them.
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
REF_SUSANNE = "/root/reference/assets/models/susanne.obj"
REF_DDS = "/root/reference/assets/textures/CathedralRadiance.dds"
REF_PNGS = ("/root/reference/assets/textures/DirectLighting.PNG", "/root/reference/assets/textures/IndirectSpecular.PNG")


# ---- independent readers (no product / oracle code) ---------------------------------------------------------

def independent_dds_cube(path):
    """numpy reading of a DX10-header DDS cube map with RGBA16F texels: (header dict, faces[6, n, n, 4] float32 of mip 0).
    Layout per the public DDS documentation: 128-byte header, 20-byte DX10 extension, then for each face its whole mip chain."""
    import struct
    d = open(path, "rb").read()
    assert d[:4] == b"DDS " and struct.unpack_from("<I", d, 4)[0] == 124
    height, width = struct.unpack_from("<II", d, 12)
    mips = max(struct.unpack_from("<I", d, 28)[0], 1)
    assert d[84:88] == b"DX10"
    dxgi, dim, misc, array_size, _ = struct.unpack_from("<5I", d, 128)
    assert dxgi == 10 and (misc & 4) and width == height                 # R16G16B16A16_FLOAT, TEXTURECUBE
    chain = sum(max(width >> m, 1) ** 2 * 8 for m in range(mips))
    assert len(d) == 148 + 6 * chain
    faces = np.stack([np.frombuffer(d, "<f2", width * width * 4, 148 + f * chain).reshape(width, width, 4).astype(np.float32)
                      for f in range(6)])
    return dict(width=width, height=height, mips=mips, dxgi_format=dxgi, bytes=len(d)), faces


def independent_obj(path):
    """Pure-Python OBJ ingestion under DESIGN.md's rules: primitive id = face order (n-gons fanned from the first corner),
    one vertex per distinct (position index, normal index) pair in first-use order, vt ignored.  -> (pos+normal float32[n,6], uint32[m,3])"""
    pos, nrm, corners = [], [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            pos.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "vn":
            nrm.append([np.float32(x) for x in t[1:4]])
        elif t[0] == "f":
            cs = []
            for c in t[1:]:
                f = c.split("/")
                pi = int(f[0])
                ni = int(f[2]) if len(f) > 2 and f[2] else 0
                assert ni != 0, "fixture meshes carry normals"
                cs.append((pi - 1 if pi > 0 else len(pos) + pi, ni - 1 if ni > 0 else len(nrm) + ni))
            for k in range(1, len(cs) - 1):
                corners += [cs[0], cs[k], cs[k + 1]]
    seen, verts, idx = {}, [], []
    for c in corners:
        if c not in seen:
            seen[c] = len(verts)
            verts.append(pos[c[0]] + nrm[c[1]])
        idx.append(seen[c])
    return np.array(verts, np.float32).reshape(-1, 6), np.array(idx, np.uint32).reshape(-1, 3)


def independent_png_rgba8(path):
    """Pure-Python PNG decode (8-bit RGBA or RGB, non-interlaced): uint8[h, w, channels]."""
    import struct
    import zlib
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    off, idat, hdr = 8, b"", None
    while off < len(d):
        n, typ = struct.unpack_from(">I4s", d, off)
        body = d[off + 8:off + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat += body
        off += 12 + n
    w, h, depth, ctype, _, _, interlace = hdr
    assert depth == 8 and ctype in (2, 6) and interlace == 0
    ch = 4 if ctype == 6 else 3
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.int32)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        cur = np.zeros(w * ch, np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:                                   # Sub / Average / Paeth carry a left-neighbour dependency
            for x in range(w * ch):
                a = cur[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) >> 1
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[x] = (line[x] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, ch)


def geometric_neighbour(face, x, y, n):
    """Texel of the cube that the direction through the centre of the off-face position (x, y) of `face` lands in: the
    D3D cube-map parameterisation evaluated numerically (float64), independent of the oracle's adjacency table."""
    sc, tc = (x + 0.5) * 2.0 / n - 1.0, (y + 0.5) * 2.0 / n - 1.0
    dx, dy, dz = [(1, -tc, -sc), (-1, -tc, sc), (sc, 1, tc), (sc, -1, -tc), (sc, -tc, 1), (-sc, -tc, -1)][face]
    ax, ay, az = abs(dx), abs(dy), abs(dz)
    if ax >= ay and ax >= az:
        g, ma, s2, t2 = (0 if dx > 0 else 1), ax, (-dz if dx > 0 else dz), -dy
    elif ay >= az:
        g, ma, s2, t2 = (2 if dy > 0 else 3), ay, dx, (dz if dy > 0 else -dz)
    else:
        g, ma, s2, t2 = (4 if dz > 0 else 5), az, (dx if dz > 0 else -dx), -dy
    return g, int(np.floor((s2 / ma + 1) / 2 * n)), int(np.floor((t2 / ma + 1) / 2 * n))


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def reference_asset_fixtures():
    """Everything a reference-held file can pin."""
    out = {}
    # --- CathedralRadiance.dds (ProgressiveRaytracingPipeline.cpp:114-118, sampled at RaytracingCommon.hlsli:152) ---
    hdr, faces = independent_dds_cube(REF_DDS)
    r = np.random.default_rng(2026)
    probes = np.stack([r.integers(0, 6, 64), r.integers(0, hdr["width"], 64), r.integers(0, hdr["width"], 64)], 1)
    out["CathedralRadiance.dds"] = dict(hdr, mip0_fp32_sha256=sha(faces),
                                        probes=[[int(f), int(y), int(x)] + [float(v) for v in faces[f, y, x]] for f, y, x in probes])
    n = hdr["width"]
    # The map was exported with edge fix-up: the outermost texel row of every face equals the row of the neighbouring
    # face across that edge.  Checked with the geometric neighbour above (24 edges x 256 texels).
    worst = 0.0
    for face in range(6):
        for k in range(n):
            for (x, y, ex, ey) in ((-1, k, 0, k), (n, k, n - 1, k), (k, -1, k, 0), (k, n, k, n - 1)):
                g, gx, gy = geometric_neighbour(face, x, y, n)
                worst = max(worst, float(np.abs(faces[face, ey, ex] - faces[g, gy, gx]).max()))
    out["CathedralRadiance.dds"]["edge_fixup_max_abs_diff"] = worst
    small = faces.astype(np.float64).reshape(6, 32, n // 32, 32, n // 32, 4).mean(axis=(2, 4)).astype(np.float32)
    dirs = r.normal(size=(1024, 3)).astype(np.float32)
    dirs[:64] = np.array([[1, 1, 0.3], [1, -1, 0.3], [-1, 0.2, 1], [0.1, 1, 1]] * 16, np.float32) * r.uniform(0.999, 1.001, (64, 3)).astype(np.float32)
    O.set_cube_seamless(True)
    env_seamless = O.sample_cube(faces, dirs)
    O.set_cube_seamless(False)
    env_clamp = O.sample_cube(faces, dirs)
    O.set_cube_seamless(True)
    # a Cornell frame lit by the down-sampled map: the committed image the GPU test checks without the oracle
    v, tri = O.obj_load(REF_OBJ)
    W = H = 64
    c = scenes.cornell_camera()
    cam = np.array([*c["eye"], *c["at"], *c["up"], c["fov"], W / H], np.float32)
    sc = O.Scene()
    sc.add_instance(sc.add_model(v, tri))
    sc.build()
    pfc = O.Progressive(4321).update(cam, 0.0, 1, W, H)
    lit, _ = sc.render(T.default_material(), pfc, W, H, env_faces=small)
    O.set_cube_seamless(False)
    lit_clamp, _ = sc.render(T.default_material(), pfc, W, H, env_faces=small)
    O.set_cube_seamless(True)
    rms = float(np.sqrt(np.mean((lit.astype(np.float64) - lit_clamp.astype(np.float64))[..., :3] ** 2)))
    out["CathedralRadiance.dds"]["cornell64_rms_seamless_vs_face_clamp"] = rms
    out["CathedralRadiance.dds"]["env_probe_max_abs_seamless_vs_face_clamp"] = float(np.abs(env_seamless - env_clamp).max())
    np.savez_compressed(os.path.join(HERE, "cathedral32.npz"), faces32=small, dirs=dirs, env_seamless=env_seamless, env_clamp=env_clamp,
                        cornell_pfc=pfc, cornell_lit=lit, cornell_lit_clamp=lit_clamp)
    # --- the two OBJ assets: ingestion digests from the independent parser ---
    for name, path in (("cornell.obj", REF_OBJ), ("susanne.obj", REF_SUSANNE)):
        vv, ii = independent_obj(path)
        ov, oi = O.obj_load(path)
        assert np.array_equal(np.concatenate([ov["position"], ov["normal"]], 1), vv) and np.array_equal(oi, ii), name
        out[name] = dict(n_verts=int(vv.shape[0]), n_tris=int(ii.shape[0]), verts_sha256=sha(vv), indices_sha256=sha(ii))
    sv, st = O.obj_load(REF_SUSANNE)
    scenes.write_obj(os.path.join(HERE, "susanne.obj"), sv, st,
                     header="Suzanne, 968 triangles: geometry of philcn/DXRExperiments assets/models/susanne.obj\n"
                            "re-emitted by dxrexperiments_amd.scenes.write_obj (tests/golden/make_fixtures.py)")
    v2, t2 = independent_obj(os.path.join(HERE, "susanne.obj"))
    assert sha(v2) == out["susanne.obj"]["verts_sha256"] and sha(t2) == out["susanne.obj"]["indices_sha256"]
    # --- the denoiser's own test inputs (src/DenoiseCompositor.cpp:52-68) ---
    crops = []
    for path in REF_PNGS:
        img = independent_png_rgba8(path)
        assert img.shape[:2] == (1126, 1922)
        crops.append(np.ascontiguousarray(img[400:544, 800:1056, :]))         # 256 x 144 window with geometry edges in it
        out[os.path.basename(path)] = dict(width=int(img.shape[1]), height=int(img.shape[0]), channels=int(img.shape[2]), rgba8_sha256=sha(img))
    direct = np.ones((144, 256, 4), np.float32)
    indirect = np.ones((144, 256, 4), np.float32)
    direct[..., :crops[0].shape[2]] = crops[0].astype(np.float32) / np.float32(255.0)      # UNORM decode, as WIC hands it to the SRV
    indirect[..., :crops[1].shape[2]] = crops[1].astype(np.float32) / np.float32(255.0)
    dparams = np.zeros((), O.DENOISE_PARAMS)
    dparams["exposure"], dparams["gamma"], dparams["tonemap"], dparams["gammaCorrect"], dparams["maxKernelSize"] = 1.0, 2.2, 1, 0, 12
    passh, final = O.denoise(direct, indirect, dparams)
    np.savez_compressed(os.path.join(HERE, "denoise_mock.npz"), direct_rgba8=crops[0], indirect_rgba8=crops[1],
                        denoise_params=np.frombuffer(dparams.tobytes(), np.uint8), pass_h=passh, composite=final)
    with open(os.path.join(HERE, "reference_assets.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("reference-held assets: seamless vs face-clamped filtering differs by RMS %.3g on the Cornell frame, max %.3g on the probe directions"
          % (rms, out["CathedralRadiance.dds"]["env_probe_max_abs_seamless_vs_face_clamp"]))


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
    reference_asset_fixtures()
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
