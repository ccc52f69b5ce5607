"""A SECOND, independent restatement of the reference's progressive shading path, in float32 numpy.

Purpose ("N-version check"): oracle/oracle_shade.h is the only restatement of RayGen -> PrimaryClosestHit ->
shade() the GPU kernels are compared with, and the kernels were written by the same hand, so a transcription error common to
both would be invisible.  This file was written from the HLSL text alone --

    assets/shaders/ProgressiveRaytracing.hlsl:11-182   RayGen, shootSecondaryRay, evaluateIndirectDiffuse, shade, hit / miss shaders
    assets/shaders/RealtimeRaytracing.hlsl:22-126      (round 4) the realtime pipeline's RayGen, shadeAOV, hit / miss shaders: two AOVs
    assets/shaders/RaytracingCommon.hlsli:53-159       interpolateVertexAttributes, shootShadowRay, evaluateAO, the two lights, env
    assets/shaders/RaytracingUtils.hlsli:22-130,209    M_PI, initRand, nextRand, perpendicular vector, the three samplers, Fresnel

-- vectorised over pixels, with BRUTE-FORCE ray/triangle intersection (no BVH), numpy's own sqrt / sin / cos / exp / power
(not the polynomial kernels of oracle_math.h / rt_device_math.h) and no code shared with oracle/ or the product.  It is test
infrastructure: tests/test_nversion_shading.py (a CPU test) renders the Cornell box with it and with the oracle and compares;
the values measured in the authoring run are committed in tests/golden/reference_assets.json -> "nversion_shading".

TraceRay semantics come from the DXR functional spec as the call sites use it: closest hit, t in (TMin, TMax) exclusive,
RAY_FLAG_CULL_BACK_FACING_TRIANGLES on primary rays only (front face = clockwise seen from the origin in a left-handed frame,
i.e. the algebraic normal (v1-v0)x(v2-v0) points towards the origin), ACCEPT_FIRST_HIT for shadow rays, barycentrics weight
v1 and v2, geometry opaque.
"""
import numpy as np

f32 = np.float32
M_PI = f32(3.1415927)             # RaytracingUtils.hlsli:22
PI2 = f32(3.14159265)             # the literal the samplers use (:71, :92, :103): the same float32
RAY_MAX_T = f32(1.0e38)
RAY_EPSILON = f32(0.0001)
MAX_RADIANCE_RAY_DEPTH = 1
MAX_SHADOW_RAY_DEPTH = 2


# ---- RNG (integer exact) ---------------------------------------------------------------------------------------

def init_rand(v0, v1):
    v0 = v0.astype(np.uint32).copy()
    v1 = np.broadcast_to(np.uint32(v1), v0.shape).copy()
    s0 = np.uint32(0)
    with np.errstate(over="ignore"):
        for _ in range(16):
            s0 = np.uint32((int(s0) + 0x9e3779b9) & 0xFFFFFFFF)
            v0 = v0 + ((((v1 << np.uint32(4)) + np.uint32(0xa341316c)) ^ (v1 + s0)) ^ ((v1 >> np.uint32(5)) + np.uint32(0xc8013ea4)))
            v1 = v1 + ((((v0 << np.uint32(4)) + np.uint32(0xad90777d)) ^ (v0 + s0)) ^ ((v0 >> np.uint32(5)) + np.uint32(0x7e95761e)))
    return v0


class Rng:
    def __init__(self, seed):
        self.s = seed.astype(np.uint32).copy()

    def next(self, mask=None):
        """nextRand on the lanes of `mask` (all if None); returns float32 values for every lane (unmasked lanes: garbage)."""
        with np.errstate(over="ignore"):
            new = np.uint32(1664525) * self.s + np.uint32(1013904223)
        if mask is None:
            self.s = new
        else:
            self.s = np.where(mask, new, self.s)
        return (new & np.uint32(0x00FFFFFF)).astype(f32) / f32(0x01000000)


# ---- small vector helpers (float32, HLSL operation order) -------------------------------------------------------

def dot(a, b):
    return a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1] + a[..., 2] * b[..., 2]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def normalize(v):
    with np.errstate(invalid="ignore", divide="ignore"):
        return v / np.sqrt(dot(v, v))[..., None]


def saturate(x):
    return np.minimum(np.maximum(x, f32(0)), f32(1))


def perpendicular(u):
    a = np.abs(u)
    xm = ((a[..., 0] - a[..., 1] < 0) & (a[..., 0] - a[..., 2] < 0)).astype(np.uint32)
    ym = np.where(a[..., 1] - a[..., 2] < 0, np.uint32(1) ^ xm, np.uint32(0))
    zm = np.uint32(1) ^ (xm | ym)
    return cross(u, np.stack([xm, ym, zm], axis=-1).astype(f32))


def cos_hemisphere(rng, n, mask):
    r1, r2 = rng.next(mask), rng.next(mask)
    bit = perpendicular(n)
    tan = cross(bit, n)
    r = np.sqrt(r1)
    phi = f32(2.0) * PI2 * r2
    x, z, y = r * np.cos(phi), r * np.sin(phi), np.sqrt(f32(1.0) - r1)
    return x[..., None] * tan + y[..., None] * n + z[..., None] * bit


def uniform_hemisphere(rng, n, mask):
    r1, r2 = rng.next(mask), rng.next(mask)
    bit = perpendicular(n)
    tan = cross(bit, n)
    cos_t = r1
    sin_t = np.sqrt(f32(1.0) - cos_t * cos_t)
    phi = f32(2.0) * PI2 * r2
    x, z, y = sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t
    return x[..., None] * tan + y[..., None] * n + z[..., None] * bit


def phong_lobe(rng, mirror, exponent, mask):
    r1, r2 = rng.next(mask), rng.next(mask)
    bit = perpendicular(mirror)
    tan = cross(bit, mirror)
    with np.errstate(invalid="ignore", divide="ignore"):
        cos_t = np.power(r1, f32(1.0) / (exponent + f32(1.0)), dtype=f32)
        sin_t = np.sqrt(f32(1.0) - cos_t * cos_t)
        phi = f32(2.0) * PI2 * r2
        powered = np.power(cos_t, exponent, dtype=f32)
    pdf = (exponent + f32(1.0)) / (f32(2.0) * PI2) * powered
    brdf = (exponent + f32(2.0)) / (f32(2.0) * PI2) * powered
    x, z, y = sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t
    return x[..., None] * tan + y[..., None] * mirror + z[..., None] * bit, pdf, brdf


def fresnel_schlick(i, n, f0):
    cosi = saturate(dot(-i, n))
    return f0 + (f32(1) - f0) * np.power(f32(1) - cosi, f32(5), dtype=f32)[..., None]


def reflect(i, n):                     # HLSL intrinsic: i - 2 * dot(i, n) * n
    return i - (f32(2.0) * dot(i, n))[..., None] * n


# ---- brute-force TraceRay ---------------------------------------------------------------------------------------

class Scene:
    def __init__(self, positions, normals, tris):
        self.p0, self.p1, self.p2 = (positions[tris[:, k]].astype(f32) for k in range(3))
        self.n0, self.n1, self.n2 = (normals[tris[:, k]].astype(f32) for k in range(3))

    def _all(self, o, d, tmin, tmax, cull):
        """t[rays, tris] (inf where no valid intersection), u, v.  Moller-Trumbore."""
        e1 = (self.p1 - self.p0)[None]
        e2 = (self.p2 - self.p0)[None]
        dd, oo = d[:, None, :], o[:, None, :]
        with np.errstate(invalid="ignore", divide="ignore", over="ignore"):
            pv = cross(dd, e2)
            det = dot(e1, pv)
            inv = f32(1.0) / det
            tv = oo - self.p0[None]
            u = dot(tv, pv) * inv
            qv = cross(tv, e1)
            v = dot(dd, qv) * inv
            t = dot(e2, qv) * inv
            ok = (det > 0) if cull else (det != 0)
            ok &= (u >= 0) & (u <= 1) & (v >= 0) & (u + v <= 1) & (t > tmin[:, None]) & (t < tmax[:, None])
        return np.where(ok, t, f32(np.inf)), u, v

    def closest(self, o, d, tmin, tmax, cull):
        t, u, v = self._all(o, d, tmin, tmax, cull)
        k = np.argmin(t, axis=1)                     # ties -> the lower primitive index
        r = np.arange(o.shape[0])
        tt = t[r, k]
        hit = np.isfinite(tt)
        return hit, np.where(hit, k, -1), tt, u[r, k], v[r, k]

    def visible(self, o, d, tmin, tmax, active):
        """1.0 where nothing lies in (tmin, tmax) (ShadowMiss), else 0.0; inactive lanes 1.0"""
        vis = np.ones(o.shape[0], f32)
        idx = np.nonzero(active)[0]
        if idx.size:
            t, _, _ = self._all(o[idx], d[idx], tmin[idx], tmax[idx], False)
            vis[idx] = np.where(np.isfinite(t).any(axis=1), f32(0.0), f32(1.0))
        return vis


# ---- shading ----------------------------------------------------------------------------------------------------

class Frame:
    def __init__(self, scene, pfc, mat, W, H, env_rgb):
        self.sc, self.pf, self.mat, self.W, self.H = scene, pfc, mat, W, H
        self.env = np.asarray(env_rgb, f32)
        self.opt = pfc["options"]
        px, py = np.meshgrid(np.arange(W, dtype=np.uint32), np.arange(H, dtype=np.uint32), indexing="xy")
        self.pix = (px + py * np.uint32(W)).reshape(-1)
        self.frame_count = np.uint32(pfc["cameraParams"]["frameCount"])

    def shadow(self, o, d, tmin, tmax, depth, active):
        if depth >= MAX_SHADOW_RAY_DEPTH:
            return np.ones(o.shape[0], f32)
        return self.sc.visible(o, d, tmin, tmax, active)

    def ao(self, P, N, active):
        rng = Rng(init_rand(self.pix, self.frame_count))
        vis = np.zeros(P.shape[0], f32)
        for _ in range(4):
            if self.opt["cosineHemisphereSampling"]:
                s = cos_hemisphere(rng, N, None)
                nol = saturate(dot(N, s))
                pdf = nol / M_PI
            else:
                s = uniform_hemisphere(rng, N, None)
                nol = saturate(dot(N, s))
                pdf = np.full_like(nol, f32(1.0) / (f32(2.0) * M_PI))
            sh = self.shadow(P, s, np.full(P.shape[0], RAY_EPSILON), np.full(P.shape[0], f32(10.0)), 1, active)
            with np.errstate(invalid="ignore", divide="ignore"):
                vis = vis + sh * nol / pdf
        return (vis / f32(4.0))[..., None] * np.ones(3, f32)

    def directional(self, P, N, depth, active):
        L = normalize(-self.pf["directionalLight"]["forwardDir"][:3].astype(f32))
        Lb = np.broadcast_to(L, P.shape).copy()
        nol = saturate(dot(N, Lb))
        vis = self.shadow(P, Lb, np.full(P.shape[0], RAY_EPSILON), np.full(P.shape[0], RAY_MAX_T), depth, active)
        col = self.pf["directionalLight"]["color"].astype(f32)
        return (col[:3] * col[3])[None] * (nol * vis)[..., None]

    def point(self, P, N, depth, active):
        path = self.pf["pointLight"]["worldPos"][:3].astype(f32)[None] - P
        dist = np.sqrt(dot(path, path))
        L = normalize(path)
        nol = saturate(dot(N, L))
        vis = self.shadow(P, L, np.full(P.shape[0], RAY_EPSILON), dist - RAY_EPSILON, depth, active)
        with np.errstate(invalid="ignore", divide="ignore"):
            falloff = f32(1.0) / (f32(2) * M_PI * dist * dist)
        col = self.pf["pointLight"]["color"].astype(f32)
        return (col[:3] * col[3])[None] * (nol * vis * falloff)[..., None]

    def secondary(self, o, d, depth, active):
        """shootSecondaryRay: radiance along (o, d) for the lanes of `active`; zero beyond the depth limit"""
        n = o.shape[0]
        if depth >= MAX_RADIANCE_RAY_DEPTH:
            return np.zeros((n, 3), f32)
        return self.trace_radiance(o, d, np.full(n, RAY_EPSILON), depth + 1, False, active)[0]

    def trace_radiance(self, o, d, tmin, depth, cull, active):
        """TraceRay with the primary hit group / miss shader: (rgb, distance)"""
        n = o.shape[0]
        rgb = np.zeros((n, 3), f32)
        dist = np.full(n, f32(-1.0))
        idx = np.nonzero(active)[0]
        if idx.size == 0:
            return rgb, dist
        hit, prim, t, u, v = self.sc.closest(o[idx], d[idx], tmin[idx], np.full(idx.size, RAY_MAX_T), cull)
        full_hit = np.zeros(n, bool)
        full_hit[idx] = hit
        fp = np.zeros(n, np.int64); ft = np.zeros(n, f32); fu = np.zeros(n, f32); fv = np.zeros(n, f32)
        fp[idx], ft[idx], fu[idx], fv[idx] = np.maximum(prim, 0), t, u, v
        # PrimaryMiss
        miss = active & ~full_hit
        rgb[miss] = self.env * f32(self.opt["environmentStrength"])
        # PrimaryClosestHit
        b0 = f32(1.0) - fu - fv
        nrm = self.sc.n0[fp] * b0[..., None] + self.sc.n1[fp] * fu[..., None] + self.sc.n2[fp] * fv[..., None]
        P = o + ft[..., None] * d
        col = self.shade(P, normalize(nrm), d, depth, full_hit)
        rgb[full_hit] = col[full_hit]
        dist[full_hit] = ft[full_hit]
        return rgb, dist

    def shade(self, P, N, D, depth, active):
        opt, m = self.opt, self.mat
        if opt["showAmbientOcclusionOnly"]:
            return self.ao(P, N, active)
        rng = Rng(init_rand(self.pix, self.frame_count))
        if opt["debug"] == 2:
            pick = rng.next(None) < f32(0.5)
            a = self.directional(P, N, depth, active & pick) * f32(2)
            b = self.point(P, N, depth, active & ~pick) * f32(2)
            direct = np.where(pick[..., None], a, b)
        else:
            direct = self.directional(P, N, depth, active)
            direct = direct + self.point(P, N, depth, active)
        indirect = np.zeros_like(P)
        if depth < 1 and not opt["noIndirectDiffuse"]:
            if opt["cosineHemisphereSampling"]:
                s = cos_hemisphere(rng, N, None)
                indirect = indirect + self.secondary(P, s, depth, active) * M_PI
            else:
                s = uniform_hemisphere(rng, N, None)
                nol = saturate(dot(N, s))
                pdf = f32(1.0) / (f32(2.0) * M_PI)
                indirect = indirect + self.secondary(P, s, depth, active) * nol[..., None] / pdf
            indirect = indirect / f32(1.0)
        diffuse = (direct + indirect) / M_PI
        fresnel = np.zeros_like(P)
        spec = np.zeros_like(P)
        if m["type"] in (1, 2) and m["reflectivity"] > 0.001:
            exponent = np.exp((f32(1.0) - f32(m["roughness"])) * f32(12.0), dtype=f32)
            mirror = reflect(D, N)
            s, pdf, brdf = phong_lobe(rng, mirror, exponent, None)
            refl = self.secondary(P, s, depth, active)
            with np.errstate(invalid="ignore", divide="ignore"):
                spec = spec + refl * brdf[..., None] / pdf[..., None]
            fresnel = fresnel_schlick(D, N, m["specular"][:3].astype(f32)[None])
        albedo = m["albedo"][:3].astype(f32)[None]
        refl_w = f32(m["reflectivity"])
        if depth == 0:
            if opt["showIndirectDiffuseOnly"]:
                return albedo * indirect / M_PI
            if opt["showIndirectSpecularOnly"]:
                return refl_w * spec * fresnel
            if opt["showFresnelTerm"]:
                return fresnel
            if opt["showGBufferAlbedoOnly"]:
                return np.broadcast_to(albedo, P.shape).copy()
            if opt["showDirectLightingOnly"]:
                return albedo * direct / M_PI
        em = m["emissive"].astype(f32)
        return (em[:3] * em[3])[None] + albedo * diffuse + refl_w * spec * fresnel


def primary_rays(pfc, W, H):
    """RayGen's ray set-up (ProgressiveRaytracing.hlsl:18-32): origins and unit directions, float32[W*H, 3] each"""
    cp = pfc["cameraParams"]
    xs, ys = np.meshgrid(np.arange(W, dtype=f32), np.arange(H, dtype=f32), indexing="xy")
    dx = (((xs + f32(0.5)) / f32(W)) * f32(2.0) - f32(1.0)).reshape(-1)
    dy = (((ys + f32(0.5)) / f32(H)) * f32(2.0) - f32(1.0)).reshape(-1)
    jit = cp["jitters"].astype(f32) * f32(30.0)
    n = W * H
    o = np.broadcast_to(cp["worldEyePos"][:3].astype(f32) + np.array([jit[0], jit[1], 0.0], f32), (n, 3)).copy()
    # normalize() of the float4 sum, then .xyz (the w components of U, V, W are zero)
    d4 = dx[:, None] * cp["U"].astype(f32)[None] + (-dy)[:, None] * cp["V"].astype(f32)[None] + cp["W"].astype(f32)[None]
    len2 = d4[:, 0] * d4[:, 0] + d4[:, 1] * d4[:, 1] + d4[:, 2] * d4[:, 2] + d4[:, 3] * d4[:, 3]
    return o, (d4 / np.sqrt(len2)[:, None])[:, :3]


def render_frame(scene, pfc, mat, W, H, prev, env_rgb=(0.5, 0.5, 0.5)):
    """RayGen over the whole image: returns (new accumulation float32[H, W, 4], primary hit prim int[H*W] (-1 = miss))."""
    cp = pfc["cameraParams"]
    if cp["accumCount"] >= pfc["options"]["maxIterations"]:
        return prev, None
    fr = Frame(scene, pfc, mat, W, H, env_rgb)
    n = W * H
    o, d = primary_rays(pfc, W, H)
    rgb, _ = fr.trace_radiance(o, d, np.zeros(n, f32), 0, True, np.ones(n, bool))
    hit, prim, t, _, _ = scene.closest(o, d, np.zeros(n, f32), np.full(n, RAY_MAX_T), True)      # the primary hit ids, for the caller
    cur = np.concatenate([np.maximum(rgb, f32(0.0)), np.ones((n, 1), f32)], axis=1).reshape(H, W, 4)
    cnt = f32(cp["accumCount"])
    out = (cnt * prev + cur) / (cnt + f32(1.0))
    return out.astype(f32), np.where(hit, prim, -1)


# ---- the realtime pipeline (RealtimeRaytracing.hlsl) --------------------------------------------------------------------

class RealtimeFrame(Frame):
    """shadeAOV instead of shade (:65-103): direct light and ONE Phong-lobe bounce, no indirect diffuse, no debug views;
    PrimaryClosestHit / PrimaryMiss write the two AOVs (:105-126); the payload's AOVs of the depth-0 hit are what RayGen stores."""

    def shade_aov(self, P, N, D, depth, active):
        m = self.mat
        rng = Rng(init_rand(self.pix, self.frame_count))
        direct = self.directional(P, N, depth, active)
        direct = direct + self.point(P, N, depth, active)
        fresnel = np.zeros_like(P)
        spec = np.zeros_like(P)
        if m["type"] in (1, 2) and m["reflectivity"] > 0.001:
            exponent = np.exp((f32(1.0) - f32(m["roughness"])) * f32(12.0), dtype=f32)
            mirror = reflect(D, N)
            s, pdf, brdf = phong_lobe(rng, mirror, exponent, None)
            refl = self.secondary(P, s, depth, active)
            with np.errstate(invalid="ignore", divide="ignore"):
                spec = spec + refl * brdf[..., None] / pdf[..., None]
            fresnel = fresnel_schlick(D, N, m["specular"][:3].astype(f32)[None])
        albedo = m["albedo"][:3].astype(f32)[None]
        dl = albedo * direct / M_PI
        isp = f32(m["reflectivity"]) * spec * fresnel
        return dl + isp, dl, isp

    def trace_radiance(self, o, d, tmin, depth, cull, active):
        """TraceRay with the realtime hit group / miss shader: (rgb, distance); the AOVs of this call are left in self.aov"""
        n = o.shape[0]
        rgb = np.zeros((n, 3), f32)
        dist = np.full(n, f32(-1.0))
        aov_d = np.zeros((n, 3), f32)
        aov_i = np.zeros((n, 3), f32)
        idx = np.nonzero(active)[0]
        if idx.size:
            hit, prim, t, u, v = self.sc.closest(o[idx], d[idx], tmin[idx], np.full(idx.size, RAY_MAX_T), cull)
            full_hit = np.zeros(n, bool)
            full_hit[idx] = hit
            fp = np.zeros(n, np.int64); ft = np.zeros(n, f32); fu = np.zeros(n, f32); fv = np.zeros(n, f32)
            fp[idx], ft[idx], fu[idx], fv[idx] = np.maximum(prim, 0), t, u, v
            miss = active & ~full_hit
            rgb[miss] = self.env * f32(self.opt["environmentStrength"])            # PrimaryMiss: colour = environment,
            aov_d[miss] = rgb[miss]                                                 #   aov.directLighting = colour, indirectSpecular = 0
            b0 = f32(1.0) - fu - fv
            nrm = self.sc.n0[fp] * b0[..., None] + self.sc.n1[fp] * fu[..., None] + self.sc.n2[fp] * fv[..., None]
            P = o + ft[..., None] * d
            col, dl, isp = self.shade_aov(P, normalize(nrm), d, depth, full_hit)     # (recursion below overwrites self.aov: saved after)
            rgb[full_hit] = col[full_hit]
            dist[full_hit] = ft[full_hit]
            if depth == 0:
                aov_d[full_hit] = dl[full_hit]
                aov_i[full_hit] = isp[full_hit]
        self.aov = (aov_d, aov_i)
        return rgb, dist


def render_frame_realtime(scene, pfc, mat, W, H, env_rgb=(0.5, 0.5, 0.5)):
    """RealtimeRaytracing.hlsl RayGen over the whole image: (direct lighting float32[H, W, 4], indirect specular float32[H, W, 4],
    primary hit prim int[H*W]); no accumulation, the jitter scaled by 10 (:33)."""
    fr = RealtimeFrame(scene, pfc, mat, W, H, env_rgb)
    n = W * H
    cp = pfc["cameraParams"]
    o, d = primary_rays(pfc, W, H)
    jit = cp["jitters"].astype(f32) * f32(10.0)
    o = np.broadcast_to(cp["worldEyePos"][:3].astype(f32) + np.array([jit[0], jit[1], 0.0], f32), (n, 3)).copy()
    fr.trace_radiance(o, d, np.zeros(n, f32), 0, True, np.ones(n, bool))
    aov_d, aov_i = fr.aov
    hit, prim, _, _, _ = scene.closest(o, d, np.zeros(n, f32), np.full(n, RAY_MAX_T), True)
    one = np.ones((n, 1), f32)
    return (np.concatenate([np.maximum(aov_d, f32(0.0)), one], axis=1).reshape(H, W, 4),
            np.concatenate([np.maximum(aov_i, f32(0.0)), one], axis=1).reshape(H, W, 4), np.where(hit, prim, -1))
