"""Re-emits the reference's assets/models/ground.fbx (the one binary FBX the reference ships; its default scene Machines.fbx is
missing from the checkout, src/DXRExperimentsApp.cpp:91) as the fixture tests/golden/ground.fbx: the geometry as the independent
parser tests/fbx_tools.py ingests it (441 vertices, 800 triangles: a 400 x 400 plane), written by that module's own FBX writer --
data, like susanne.obj / cornell.obj beside it, so that the GPU box, where /root/reference does not exist, can run BASELINE
configs[3] ("FBX multi-mesh scene") with a BLAS that comes from the reference's own file through rt_model_create_from_file.
Run here (needs /root/reference):  python tests/golden/make_ground_fbx.py"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import fbx_tools as F  # noqa: E402

REF = "/root/reference/assets/models/ground.fbx"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    v, i = F.ingest(REF)
    pins = json.load(open(os.path.join(HERE, "reference_assets.json")))["ground.fbx"]
    assert sha(v) == pins["verts_sha256"] and sha(i) == pins["indices_sha256"], "the reference file is not the pinned one"
    mesh = dict(positions=v[:, :3].astype(np.float64), polygons=[list(map(int, t)) for t in i],
                normals=v[i.reshape(-1), 3:6].astype(np.float64), mapping="ByPolygonVertex")
    out = os.path.join(HERE, "ground.fbx")
    F.write(out, [mesh], version=7500, compress=True)
    v2, i2 = F.ingest(out)
    assert np.array_equal(v2, v) and np.array_equal(i2, i), "the re-emission does not ingest to the same arrays"
    print("wrote %s: %d vertices, %d triangles, %d bytes" % (out, v.shape[0], i.shape[0], os.path.getsize(out)))


if __name__ == "__main__":
    main()
