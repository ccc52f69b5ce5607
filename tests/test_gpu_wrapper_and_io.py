"""The reference-shaped C++ API (dxrexperiments_amd/include) end to end, the headless example,
DDS environment ingestion and the RGBA16F output view."""
import os
import struct
import subprocess
import time

import numpy as np
import pytest

from dxrexperiments_amd import rtypes as T, scenes
from util import CORNELL_OBJ, GOLDEN

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "dxrexperiments_amd", "lib")


def test_cpp_wrapper_realtime_and_denoiser(tmp_path, oracle):
    """RealtimeRaytracingPipeline + DenoiseCompositor through the C++ mirror == oracle (same host seed 77, frame 5)."""
    exe = os.path.join(LIBDIR, "test_wrapper")
    out, out3 = tmp_path / "img.f32", tmp_path / "three.f32"
    r = subprocess.run([exe, CORNELL_OBJ, str(out), str(out3)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    three = np.fromfile(out3, np.float32).reshape(3, 64, 64, 4)
    v, i = oracle.obj_load(CORNELL_OBJ)
    sc = oracle.Scene()
    sc.add_instance(sc.add_model(v, i))
    sc.build()
    host = oracle.Progressive(77)
    cam = np.array([0, 0, 3.2, 0, 0, 0, 0, 1, 0, np.float32(3.14159265358979 / 4.0), 1.0], np.float32)
    pfc = np.frombuffer(host.update(cam, 0.0, 5, 64, 64).tobytes(), T.PER_FRAME_CONSTANTS).copy()
    pfc["cameraParams"]["accumCount"] = 0
    pfc["options"] = np.zeros((), T.DEBUG_OPTIONS)
    pfc["options"]["environmentStrength"] = 1.0
    d, ind, _ = sc.render_realtime(T.default_material(), pfc, 64, 64, env_constant=(0.5, 0.5, 0.5), nthreads=4)
    prm = np.zeros((), oracle.DENOISE_PARAMS)
    prm["exposure"], prm["gamma"], prm["tonemap"], prm["maxKernelSize"] = 1.0, 2.2, 1, 12
    _, ov = oracle.denoise(d, ind, prm)
    assert np.array_equal(three[0], d) and np.array_equal(three[1], ind)
    assert np.array_equal(np.nan_to_num(three[2], nan=-1), np.nan_to_num(ov, nan=-1))


def test_cpp_wrapper_reproduces_golden_frames(tmp_path):
    """ProgressiveRaytracingPipeline::create/setScene/addMaterial/setCamera/update/render through the C++ mirror,
    4 frames of Cornell 64x64 with host seed 1234 == the committed oracle image."""
    exe = os.path.join(LIBDIR, "test_wrapper")
    assert os.path.exists(exe), "run `make` first"
    out = tmp_path / "img.f32"
    r = subprocess.run([exe, CORNELL_OBJ, str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    img = np.fromfile(out, np.float32).reshape(64, 64, 4)
    g = np.load(os.path.join(GOLDEN, "cornell64_golden.npz"))
    assert np.array_equal(img, g["images"][3])


def test_headless_example_writes_pfm(tmp_path):
    exe = os.path.join(LIBDIR, "progressive")
    out = tmp_path / "out.pfm"
    r = subprocess.run([exe, CORNELL_OBJ, "96", "64", "3", str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "Million Primary Rays/s" in r.stdout and "Progressive Ray Tracing Pipeline" in r.stdout
    data = open(out, "rb").read()
    assert data.startswith(b"PF\n96 64\n-1.0\n") and len(data) == len(b"PF\n96 64\n-1.0\n") + 96 * 64 * 12
    # DXR_SETS=n: the same frames through shared sets of launches (renderBatch of the C++ mirror): the same file, byte for byte
    out2 = tmp_path / "out_sets.pfm"
    r = subprocess.run([exe, CORNELL_OBJ, "96", "64", "7", str(out)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    r = subprocess.run([exe, CORNELL_OBJ, "96", "64", "7", str(out2)], env=dict(os.environ, DXR_SETS="3"), stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert open(out, "rb").read() == open(out2, "rb").read()


def write_dds_cube(path, faces, fmt):
    """Minimal DX10-header DDS cube map writer (test data only). fmt: 10 = RGBA16F, 2 = RGBA32F."""
    size = faces.shape[1]
    hdr = struct.pack("<4sI", b"DDS ", 124) + struct.pack("<IIIIII", 0x1007 | 0x20000, size, size, size * (8 if fmt == 10 else 16), 0, 2)
    hdr += b"\0" * 44
    hdr += struct.pack("<II4sIIIII", 32, 4, b"DX10", 0, 0, 0, 0, 0)
    hdr += struct.pack("<IIIII", 0x401008, 0xFE00, 0, 0, 0)
    hdr += struct.pack("<IIIII", fmt, 3, 4, 1, 0)
    assert len(hdr) == 148
    with open(path, "wb") as f:
        f.write(hdr)
        for k in range(6):
            mip0 = faces[k].astype(np.float16 if fmt == 10 else np.float32)
            f.write(mip0.tobytes())
            half = mip0[::2, ::2]                      # a second mip level the reader must skip
            f.write(half.tobytes())


@pytest.mark.parametrize("fmt", [10, 2])
def test_dds_environment_equals_array_environment(gpu, capi, oracle, tmp_path, fmt):
    faces = scenes.sky_cubemap(16)
    if fmt == 10:
        faces = faces.astype(np.float16).astype(np.float32)      # what survives an fp16 file
    path = tmp_path / "env.dds"
    write_dds_cube(str(path), faces, fmt)
    v, i = oracle.obj_load(CORNELL_OBJ)
    imgs = []
    for use_dds in (True, False):
        sc = capi.Scene(gpu)
        sc.add_model(capi.Model(gpu, v, i))
        p = capi.Pipeline(gpu)
        p.set_scene(sc)
        p.add_material(T.default_material())
        if use_dds:
            p.load_environment_dds(str(path))
        else:
            p.set_environment_cube(faces)
        p.create_output(48, 48)
        p.build_acceleration_structures()
        host = capi.ProgressiveHost(2)
        cam = capi.camera_array((0.0, 0.0, 3.2), (0.3, 0.2, 0.0), (0, 1, 0), 0.9, 1.0)
        p.update(host.update(cam, 0.0, 1, 48, 48))
        p.render()
        imgs.append(p.read_output())
    assert np.array_equal(imgs[0], imgs[1])
    with pytest.raises(capi.RtError):
        capi.Pipeline(gpu).load_environment_dds(str(tmp_path / "missing.dds"))
    bad = tmp_path / "bad.dds"
    bad.write_bytes(b"not a dds file at all" * 10)
    with pytest.raises(capi.RtError):
        capi.Pipeline(gpu).load_environment_dds(str(bad))


def test_realtime_denoise_example_writes_png(tmp_path):
    """The second mode of the reference app (realtime pipeline -> compositor) through the C++ mirror, end to end."""
    import struct
    import zlib
    exe = os.path.join(LIBDIR, "realtime_denoise")
    out = tmp_path / "out.png"
    r = subprocess.run([exe, CORNELL_OBJ, "96", "64", "3", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Realtime Ray Tracing Pipeline + denoise" in r.stdout
    raw = out.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n" and struct.unpack(">II", raw[16:24]) == (96, 64)
    pos, idat = 8, b""
    while pos < len(raw):
        n, typ = struct.unpack(">I4s", raw[pos:pos + 8])
        if typ == b"IDAT":
            idat += raw[pos + 8:pos + 8 + n]
        pos += 12 + n
    px = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(64, 96 * 3 + 1)[:, 1:]
    assert px.max() > 32 and px.std() > 1.0            # an image, not a constant


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_multi_gpu_example_from_the_c_abi(tmp_path, mode):
    """examples/progressive_multi.cpp: launcher forks one process per GPU before any GPU call, RCCL is opened from the
    C ABI (rt_dist_*), the rank renders and runs the collective.  A test box has ONE GPU (and RCCL wants a device per
    rank), so this runs world size 1: it proves the dlopen, the entry points, communicator creation and both
    collectives on the context stream, and that the result equals the single-process example's image."""
    multi, single = os.path.join(LIBDIR, "progressive_multi"), os.path.join(LIBDIR, "progressive")
    a, b = tmp_path / "multi.pfm", tmp_path / "single.pfm"
    r = subprocess.run([multi, CORNELL_OBJ, "96", "64", "4", str(a), "1", mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    assert "on 1 GPU(s)" in r.stdout
    r = subprocess.run([single, CORNELL_OBJ, "96", "64", "4", str(b)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    x = np.frombuffer(a.read_bytes()[-96 * 64 * 12:], np.float32)
    y = np.frombuffer(b.read_bytes()[-96 * 64 * 12:], np.float32)
    if mode == "tiles":
        assert np.array_equal(x, y)                      # bands of the same frames: bit for bit
    else:
        assert float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2))) <= 1e-5      # sum / n against the running mean
    # more ranks than GPUs is refused up front, not deadlocked
    r = subprocess.run([multi, CORNELL_OBJ, "32", "32", "1", str(a), "2", mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and "visible GPUs" in r.stdout
    # two ranks forced onto ONE device: rt_dist_create compares the ranks' PCI bus ids before RCCL is asked, and says so
    # (RCCL itself would fail late or hang); the launcher returns promptly with a non-zero code
    t0 = time.time()
    r = subprocess.run([multi, CORNELL_OBJ, "32", "32", "1", str(a), "2", mode], env=dict(os.environ, DXR_MULTI_DEVICE="0"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and "are both on the device at PCI" in r.stdout, r.stdout
    assert time.time() - t0 < 120
