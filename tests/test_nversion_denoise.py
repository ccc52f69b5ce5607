"""N-version check of the DenoiseCompositor oracle.

oracle/oracle_shade.h's denoiser is what the GPU kernels (rt_denoise.hip) are compared with bit for bit;
tests/golden/nversion_denoise.py is a SECOND restatement of BilateralFilter.hlsli + DenoiseCommon.hlsli, written from the HLSL text
alone in float32 numpy (vectorised shifts instead of per-pixel loops, no code shared with oracle/ or the product).  Both run on the
reference's own mock inputs (crops of assets/textures/DirectLighting.PNG and IndirectSpecular.PNG, tests/golden/denoise_mock.npz)
under parameter sets that reach every branch of the two passes: kernel sizes 0 ... 20 (the cache's MAX_EXTENT), the four debug views,
tone map and gamma on and off.  Everything but the gamma's pow() is additions, multiplications and one division per pixel in a
fixed order, so the two must agree BIT FOR BIT there; with gamma correction on, to 1e-6 (numpy's pow against the oracle's)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.dirname(HERE))

import nversion_denoise as ND                            # noqa: E402

CASES = {
    "reference_defaults": {},
    "kernel_3": {"maxKernelSize": 3},
    "kernel_20_no_tonemap": {"maxKernelSize": 20, "tonemap": 0},
    "kernel_0": {"maxKernelSize": 0},
    "view_indirect_filtered": {"debugVisualize": 1},
    "view_indirect_unfiltered": {"debugVisualize": 2},
    "view_direct": {"debugVisualize": 3},
    "exposure_2_5": {"exposure": 2.5, "tonemap": 0},
    "gamma_2_2": {"gammaCorrect": 1},
    "gamma_1_8_no_tonemap": {"gammaCorrect": 1, "gamma": 1.8, "tonemap": 0, "exposure": 0.7},
}


def inputs():
    g = np.load(os.path.join(HERE, "golden", "denoise_mock.npz"))
    direct = np.ones((144, 256, 4), np.float32)
    indirect = np.ones((144, 256, 4), np.float32)
    direct[..., :g["direct_rgba8"].shape[2]] = g["direct_rgba8"].astype(np.float32) / np.float32(255.0)
    indirect[..., :g["indirect_rgba8"].shape[2]] = g["indirect_rgba8"].astype(np.float32) / np.float32(255.0)
    return g, direct, indirect


@pytest.mark.parametrize("name", sorted(CASES))
def test_second_restatement_of_the_denoiser_agrees_with_the_oracle(name):
    from oracle import pyoracle as O
    g, direct, indirect = inputs()
    p = np.frombuffer(g["denoise_params"].tobytes(), O.DENOISE_PARAMS)[0].copy()
    for k, v in CASES[name].items():
        p[k] = v
    oh, ov = O.denoise(direct, indirect, p)
    nh, nv = ND.denoise(direct, indirect, p["exposure"], p["gamma"], p["tonemap"], p["gammaCorrect"], p["maxKernelSize"], p["debugVisualize"])
    assert np.array_equal(oh, nh), "H pass: %d values differ" % int((oh != nh).sum())
    if p["gammaCorrect"]:
        assert np.abs(ov.astype(np.float64) - nv).max() <= 1e-6
    else:
        assert np.array_equal(ov, nv), "composite: %d values differ" % int((ov != nv).sum())
    assert np.isfinite(ov).all() and ov[..., 3].min() == 1.0


def test_the_weights_table_is_the_shader_s():
    """BilateralFilter.hlsli:84-93 at the reference's default radius (12): taps 0..1 -> 1, then 0.9, 0.75, 0.6, 0.5 in steps of
    12 * 0.8 / 5 = 1.92 taps, zero beyond the radius"""
    w = ND.gaussian_weights(12.0)[ND.MAX_EXTENT:]
    assert list(w[:13]) == [1.0, 1.0, 1.0, 1.0, np.float32(0.9), np.float32(0.9), np.float32(0.75), np.float32(0.75), np.float32(0.6), np.float32(0.6), 0.5, 0.5, 0.0]
    assert (w[12:] == 0).all()


# ---- the environment cube: a second statement of D3D's cube addressing --------------------------------------------

def cube_sample_second_statement(faces, dirs):
    """TextureCube.SampleLevel(linear, dir, 0) as the D3D functional spec states it: the face is the major axis (ties: x before y
    before z), (sc, tc) per its face table -- +X (-z, -y), -X (z, -y), +Y (x, z), -Y (x, -z), +Z (x, -y), -Z (-x, -y) --, texel
    coordinates (s * N - 0.5, t * N - 0.5), bilinear with the taps clamped to the face.  float32 throughout, one direction at a time."""
    f32 = np.float32
    n = faces.shape[1]
    out = np.zeros((dirs.shape[0], 3), f32)
    for i, (x, y, z) in enumerate(dirs.astype(f32)):
        ax, ay, az = abs(x), abs(y), abs(z)
        if ax >= ay and ax >= az:
            face, ma, sc, tc = (0, ax, -z, -y) if x > 0 else (1, ax, z, -y)
        elif ay >= az:
            face, ma, sc, tc = (2, ay, x, z) if y > 0 else (3, ay, x, -z)
        else:
            face, ma, sc, tc = (4, az, x, -y) if z > 0 else (5, az, -x, -y)
        u = (f32(sc) / f32(ma) + f32(1)) * f32(0.5)
        v = (f32(tc) / f32(ma) + f32(1)) * f32(0.5)
        fx, fy = u * f32(n) - f32(0.5), v * f32(n) - f32(0.5)
        x0, y0 = int(np.floor(fx)), int(np.floor(fy))
        wx, wy = f32(fx - x0), f32(fy - y0)
        c = lambda a: min(max(a, 0), n - 1)                                  # noqa: E731
        top = faces[face, c(y0), c(x0), :3] * (f32(1) - wx) + faces[face, c(y0), c(x0 + 1), :3] * wx
        bot = faces[face, c(y0 + 1), c(x0), :3] * (f32(1) - wx) + faces[face, c(y0 + 1), c(x0 + 1), :3] * wx
        out[i] = top * (f32(1) - wy) + bot * wy
    return out


def test_second_statement_of_the_cube_addressing_agrees_with_the_oracle():
    """the oracle's sample_cube (what sample_environment of the kernels is compared with bit for bit) against the statement above:
    2000 random directions and the 26 axis / diagonal ones on a random 8 x 8 cube, face-clamped filter: <= 1e-6 (rounding order);
    and the seamless filter (the default) equals the clamped one wherever the four taps lie inside one face"""
    from oracle import pyoracle as O
    r = np.random.default_rng(3)
    faces = r.uniform(0, 1, (6, 8, 8, 4)).astype(np.float32)
    special = np.array([[a, b, c] for a in (-1, 0, 1) for b in (-1, 0, 1) for c in (-1, 0, 1) if (a, b, c) != (0, 0, 0)], np.float32)
    dirs = np.concatenate([r.normal(0, 1, (2000, 3)).astype(np.float32), special])
    try:
        O.set_cube_seamless(False)
        clamped = O.sample_cube(faces, dirs)
        O.set_cube_seamless(True)
        seamless = O.sample_cube(faces, dirs)
    finally:
        O.set_cube_seamless(True)
    want = cube_sample_second_statement(faces, dirs)
    assert np.abs(clamped[:, :3] - want).max() <= 1e-6
    # interior of a face: |sc|, |tc| <= (N - 1) / N of the major axis
    a = np.abs(dirs)
    ma = a.max(axis=1)
    second = np.sort(a, axis=1)[:, 1]
    inside = second <= ma * np.float32(7.0 / 8.0 - 1e-3)
    assert inside.sum() > 500 and np.array_equal(clamped[inside], seamless[inside])
    assert np.abs(clamped - seamless).max() > 0.01            # (and the two filters do differ across the edges)
