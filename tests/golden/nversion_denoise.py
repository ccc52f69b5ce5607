"""A SECOND, independent restatement of the reference's DenoiseCompositor, in float32 numpy.

Written from the HLSL text alone --

    assets/shaders/BilateralFilter.hlsli:1-127      the separated joint bilateral filter (weights table, colour weight, the kernel loop)
    assets/shaders/DenoiseCommon.hlsli:1-84         the two passes' main(), composite, exposure, Reinhard tone map, gamma
    assets/shaders/DenoiseCompositorH.hlsl / V.hlsl PASS 0 = along x, PASS 1 = along y

-- vectorised over the image, no code shared with oracle/ or the product; numpy's own power function.  It is test infrastructure:
tests/test_nversion_shading.py compares it with the oracle's orc_denoise on the reference's own mock inputs
(assets/textures/DirectLighting.PNG / IndirectSpecular.PNG crops, tests/golden/denoise_mock.npz).

Texture reads outside the image return zero (D3D: out-of-bounds loads of a Texture2D), the groupshared cache of the shaders
(PREFETCH_TEXTURES, 64 + 2 x 20 entries per group, filled from 32 texels either side) holds exactly the texels an unclipped read
would return as long as |gMaxKernelSize| <= 20 < 32, which the reference's UI guarantees -- so the cache is not modelled."""
import numpy as np

f32 = np.float32
KERNEL_TAPS = 6
MAX_EXTENT = 20


def gaussian_weights(kernel_radius):
    """sPrecalculatedGaussianWeights (BilateralFilter.hlsli:84-93): index i + MAX_EXTENT"""
    w = np.zeros(2 * MAX_EXTENT + 1, f32)
    for i in range(-MAX_EXTENT, MAX_EXTENT + 1):
        q = f32(abs(i) * (KERNEL_TAPS - 1)) / (f32(0.001) + f32(abs(f32(kernel_radius) * f32(0.8))))
        idx = min(max(int(q), 0), KERNEL_TAPS)
        w[i + MAX_EXTENT] = 1.0 if idx < 2 else (0.9 if idx < 3 else (0.75 if idx < 4 else (0.6 if idx < 5 else (0.5 if idx < 6 else 0.0))))
    return w


def shifted(img, off, axis):
    """img[p + off] along `axis` (0 = y, 1 = x), zero outside the image"""
    out = np.zeros_like(img)
    n = img.shape[axis]
    if abs(off) >= n:
        return out
    src = [slice(None)] * img.ndim
    dst = [slice(None)] * img.ndim
    if off >= 0:
        src[axis] = slice(off, n); dst[axis] = slice(0, n - off)
    else:
        src[axis] = slice(0, n + off); dst[axis] = slice(-off, n)
    out[tuple(dst)] = img[tuple(src)]
    return out


def filter_pass(inp, joint, k, axis):
    """filterKernel (:80-120) over the whole image: inp filtered along `axis`, weights from `joint`"""
    gw = gaussian_weights(float(k))
    color = np.zeros_like(inp)
    weight = np.zeros(inp.shape[:2], f32)
    for i in range(-k, k + 1):
        s = shifted(inp, i, axis)
        sj = shifted(joint, i, axis)
        dist = (np.abs(sj[..., 0] - joint[..., 0]) + np.abs(sj[..., 1] - joint[..., 1]) + np.abs(sj[..., 2] - joint[..., 2])) * f32(10.0)
        cw = f32(1.0) - np.clip(dist, f32(0.0), f32(1.0))
        bw = (gw[i + MAX_EXTENT] * cw).astype(f32)
        color = color + s * bw[..., None]
        weight = weight + bw
    with np.errstate(invalid="ignore", divide="ignore"):
        return (color / weight[..., None]).astype(f32)


def denoise(direct, indirect, exposure, gamma, tonemap, gamma_correct, max_kernel_size, debug_visualize):
    """-> (the H pass's output, the V pass's output = the composite), float32[H, W, 4] each, alpha 1"""
    direct = np.asarray(direct, f32); indirect = np.asarray(indirect, f32)
    k = int(max_kernel_size)
    # PASS 0: gInput = the indirect-specular AOV, joint = the direct-lighting AOV
    if debug_visualize == 2:
        h = indirect[..., :3].copy()
    else:
        h = filter_pass(indirect, direct, k, 1)[..., :3]
    hp = np.concatenate([h, np.ones(h.shape[:2] + (1,), f32)], axis=2)
    # PASS 1: gInput = PASS 0's output
    if debug_visualize == 2:
        c = hp[..., :3].copy()
    else:
        c = filter_pass(hp, direct, k, 0)[..., :3]
    if debug_visualize == 0:
        c = c + direct[..., :3]
    elif debug_visualize == 3:
        c = direct[..., :3].copy()
    c = (c * f32(exposure)).astype(f32)
    if tonemap:
        lum = c[..., 0] * f32(0.299) + c[..., 1] * f32(0.587) + c[..., 2] * f32(0.114)
        with np.errstate(invalid="ignore", divide="ignore"):
            reinhard = lum / (lum + f32(1.0))
            c = np.maximum(c * (reinhard / lum)[..., None], f32(0.0))      # max(NaN, 0) = 0 in HLSL
        c = np.where(np.isnan(c), f32(0.0), c).astype(f32)
    if gamma_correct:
        with np.errstate(invalid="ignore"):
            c = np.clip(np.power(c, f32(1.0) / f32(gamma), dtype=f32), f32(0.0), f32(1.0))
        c = np.where(np.isnan(c), f32(0.0), c).astype(f32)
    return hp, np.concatenate([c.astype(f32), np.ones(c.shape[:2] + (1,), f32)], axis=2)
