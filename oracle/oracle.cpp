/*
 * oracle.cpp -- TEST INFRASTRUCTURE ONLY.
 *
 * C entry points of the CPU oracle: a scalar restatement of the reference's
 * ProgressiveRaytracingPipeline path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product
 * (dxrexperiments_amd/) never does and fails loudly without its HIP library.
 *
 * PARITY STATUS: the reference ships no tests, golden vectors or fixtures and
 * cannot be built or run here (Windows/D3D12/DXC + un-vendored Fallback
 * Layer).  The shading/RNG/raygen half follows the reference's HLSL text line
 * by line and is pinned by the integer RNG known-answer values derived from
 * that text; the acceleration-structure and intersection
 * half is "parity unpinned" -- see oracle_bvh.h.
 *
 * Host-side restatements in this file:
 *   RtModel::RtModel mesh ingestion      libs/DXRFramework/RtModel.cpp:24-82
 *   calculateCameraVariables             src/ProgressiveRaytracingPipeline.cpp:151-168
 *   ProgressiveRaytracingPipeline::update src/ProgressiveRaytracingPipeline.cpp:177-213
 *   BaseCamera::SetLookDirection         libs/MiniEngine/Camera.cpp:19-36
 */
#include <stdio.h>
#include <stdlib.h>
#include <map>
#include <string>
#include <thread>
#include <utility>
#include "oracle_shade.h"
#include "oracle.h"
#include "truth64.h"

using namespace orc;

struct orc_scene {
    Scene s;
    truth64::Scene truth;       /* the same arrays for the float64 geometric truth (truth64.h); filled by orc_scene_build */
};

/* RenderCtx::truth: the float64 truth as the frame's tracer (its t, u, v rounded to fp32 for the shading code) */
static Hit truth_hook(const void *ts, const Ray &r, uint32_t flags)
{
    const float o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
    const truth64::Hit h = truth64::trace(*(const truth64::Scene *)ts, o, r.tmin, d, r.tmax,
                                          (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0,
                                          (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0);
    Hit out = miss_hit(r);
    if (h.inst != 0xFFFFFFFFu) { out.t = (float)h.t; out.u = (float)h.u; out.v = (float)h.v; out.prim = h.prim; out.inst = h.inst; }
    return out;
}

extern "C" {

/* ---- known-answer primitives ------------------------------------------- */

uint32_t orc_init_rand(uint32_t v0, uint32_t v1) { return initRand(v0, v1); }
float orc_next_rand(uint32_t *s) { return nextRand(s); }

void orc_round_to_half(const float *x, float *out, size_t n, int nearest)
{
    for (size_t i = 0; i < n; i++) out[i] = round_to_half(x[i], nearest != 0);
}

void orc_math_batch(int fn, const float *x, const float *y, float *out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        float s, c;
        switch (fn) {
        case ORC_FN_SIN: sincos_(x[i], &s, &c); out[i] = s; break;
        case ORC_FN_COS: sincos_(x[i], &s, &c); out[i] = c; break;
        case ORC_FN_EXP: out[i] = exp_(x[i]); break;
        case ORC_FN_LOG: out[i] = log_(x[i]); break;
        case ORC_FN_POW: out[i] = pow_(x[i], y[i]); break;
        case ORC_FN_SQRT: out[i] = sqrtf(x[i]); break;
        case ORC_FN_DIV: out[i] = x[i] / y[i]; break;
        case ORC_FN_MIN: out[i] = fmin_(x[i], y[i]); break;
        case ORC_FN_MAX: out[i] = fmax_(x[i], y[i]); break;
        default: out[i] = 0; break;
        }
    }
}

void orc_sample_batch(int kind, const uint32_t *seeds, const float *vec3_in, float exponent,
                      float *vec3_out, float *pdf_brdf, uint32_t *seeds_out, size_t n)
{
    for (size_t i = 0; i < n; i++) {
        uint32_t s = seeds[i];
        V3 in = v3(vec3_in[3 * i], vec3_in[3 * i + 1], vec3_in[3 * i + 2]);
        V3 o = v3(0, 0, 0);
        float pdf = 0, brdf = 0;
        switch (kind) {
        case ORC_SAMPLE_COS:     o = getCosHemisphereSample(&s, in); break;
        case ORC_SAMPLE_UNIFORM: o = getUniformHemisphereSample(&s, in); break;
        case ORC_SAMPLE_PHONG:   o = samplePhongLobe(&s, in, exponent, &pdf, &brdf); break;
        case ORC_SAMPLE_PERP:    o = getPerpendicularVector(in); break;
        default: break;
        }
        vec3_out[3 * i] = o.x; vec3_out[3 * i + 1] = o.y; vec3_out[3 * i + 2] = o.z;
        if (pdf_brdf) { pdf_brdf[2 * i] = pdf; pdf_brdf[2 * i + 1] = brdf; }
        if (seeds_out) seeds_out[i] = s;
    }
}

void orc_fresnel(const float I[3], const float N[3], const float f0[3], float out[3])
{
    V3 r = FresnelReflectanceSchlick(v3(I[0], I[1], I[2]), v3(N[0], N[1], N[2]), v3(f0[0], f0[1], f0[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* cube-map filter used by orc_sample_cube / orc_render / orc_render_realtime: 1 = seamless (default), 0 = clamp to the face */
static int g_cube_seamless = 1;
void orc_set_cube_seamless(int on) { g_cube_seamless = on; }

void orc_sample_cube(const float *faces, int size, const float *dirs, float *out, size_t n)
{
    Env e; e.faces = faces; e.size = size; e.constant[0] = e.constant[1] = e.constant[2] = 0;
    e.seamless = g_cube_seamless != 0;
    for (size_t i = 0; i < n; i++) {
        V3 c = sampleCube(e, v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
        out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
    }
}

/* ---- mesh ingestion ------------------------------------------------------ */

void orc_free(void *p) { free(p); }

/*
 * Minimal Wavefront OBJ reader standing in for the reference's Assimp import
 * (RtModel.cpp:26-27: Triangulate | GenSmoothNormals | JoinIdenticalVertices |
 * PreTransformVertices; FlipUVs is moot, UVs are dropped).  Assimp itself is
 * absent, so vertex and primitive ORDER are this engine's definition:
 *   primitive id = face order in the file, polygons fan-triangulated;
 *   one output vertex per distinct (position index, normal index) pair, in
 *   first-use order; a corner without a normal gets the normalised sum of the
 *   (area-weighted) face normals of every face using that position index.
 */
int orc_obj_load(const char *path, rt_vertex **verts_out, uint32_t *nv_out, uint32_t **idx_out, uint32_t *nt_out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return -1;
    std::vector<V3> pos, nrm;
    struct Corner { int p, n; };
    std::vector<Corner> corners;            /* 3 per triangle */
    char line[4096];
    while (fgets(line, sizeof line, f)) {
        char *s = line;
        while (*s == ' ' || *s == '\t') s++;
        if (s[0] == 'v' && (s[1] == ' ' || s[1] == '\t')) {
            char *e = s + 1;
            float x = strtof(e, &e), y = strtof(e, &e), z = strtof(e, &e);
            pos.push_back(v3(x, y, z));
        } else if (s[0] == 'v' && s[1] == 'n' && (s[2] == ' ' || s[2] == '\t')) {
            char *e = s + 2;
            float x = strtof(e, &e), y = strtof(e, &e), z = strtof(e, &e);
            nrm.push_back(v3(x, y, z));
        } else if (s[0] == 'f' && (s[1] == ' ' || s[1] == '\t')) {
            std::vector<Corner> poly;
            char *e = s + 1;
            for (;;) {
                while (*e == ' ' || *e == '\t') e++;
                if (*e == 0 || *e == '\n' || *e == '\r' || *e == '#') break;
                char *q;
                long pi = strtol(e, &q, 10);
                if (q == e) break;
                long ni = 0; bool has_n = false;
                e = q;
                if (*e == '/') {
                    e++;
                    if (*e != '/') { strtol(e, &q, 10); e = q; }        /* texture index, dropped */
                    if (*e == '/') { e++; ni = strtol(e, &q, 10); has_n = (q != e); e = q; }
                }
                Corner c;
                c.p = (int)(pi < 0 ? (long)pos.size() + pi : pi - 1);
                c.n = has_n ? (int)(ni < 0 ? (long)nrm.size() + ni : ni - 1) : -1;
                poly.push_back(c);
            }
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                corners.push_back(poly[0]); corners.push_back(poly[k]); corners.push_back(poly[k + 1]);
            }
        }
    }
    fclose(f);
    const size_t nt = corners.size() / 3;
    for (const Corner &c : corners)
        if (c.p < 0 || c.p >= (int)pos.size() || c.n >= (int)nrm.size()) return -2;

    /* smooth normals for corners that carry none */
    std::vector<V3> gen;
    bool need_gen = false;
    for (const Corner &c : corners) if (c.n < 0) need_gen = true;
    if (need_gen) {
        gen.assign(pos.size(), v3(0, 0, 0));
        for (size_t t = 0; t < nt; t++) {
            V3 a = pos[corners[3 * t].p], b = pos[corners[3 * t + 1].p], c = pos[corners[3 * t + 2].p];
            V3 fn = cross3(vsub(b, a), vsub(c, a));
            for (int k = 0; k < 3; k++) gen[corners[3 * t + k].p] = vadd(gen[corners[3 * t + k].p], fn);
        }
        for (V3 &g : gen) {
            float l = sqrtf(dot3(g, g));
            if (l > 0.0f) g = vdivs(g, l);
        }
    }

    std::map<std::pair<int, int>, uint32_t> seen;
    std::vector<rt_vertex> verts;
    std::vector<uint32_t> idx;
    idx.reserve(corners.size());
    for (const Corner &c : corners) {
        std::pair<int, int> key(c.p, c.n);
        std::map<std::pair<int, int>, uint32_t>::iterator it = seen.find(key);
        if (it == seen.end()) {
            rt_vertex v;
            v.position.x = pos[c.p].x; v.position.y = pos[c.p].y; v.position.z = pos[c.p].z;
            V3 n = c.n >= 0 ? nrm[c.n] : gen[c.p];
            v.normal.x = n.x; v.normal.y = n.y; v.normal.z = n.z;
            uint32_t id = (uint32_t)verts.size();
            verts.push_back(v);
            seen[key] = id;
            idx.push_back(id);
        } else idx.push_back(it->second);
    }
    *verts_out = (rt_vertex *)malloc(sizeof(rt_vertex) * (verts.size() ? verts.size() : 1));
    *idx_out = (uint32_t *)malloc(sizeof(uint32_t) * (idx.size() ? idx.size() : 1));
    memcpy(*verts_out, verts.data(), sizeof(rt_vertex) * verts.size());
    memcpy(*idx_out, idx.data(), sizeof(uint32_t) * idx.size());
    *nv_out = (uint32_t)verts.size();
    *nt_out = (uint32_t)nt;
    return 0;
}

/* ---- scene ---------------------------------------------------------------- */

orc_scene *orc_scene_create(void) { return new orc_scene(); }
void orc_scene_destroy(orc_scene *s) { delete s; }

int orc_scene_add_model(orc_scene *sc, const rt_vertex *verts, uint32_t nv, const uint32_t *idx, uint32_t nt)
{
    for (uint32_t i = 0; i < 3 * nt; i++) if (idx[i] >= nv) return -1;
    Model m;
    m.verts.assign(verts, verts + nv);
    m.idx.assign(idx, idx + 3 * (size_t)nt);
    m.ntris = nt;
    sc->s.models.push_back(std::move(m));
    sc->s.built = false;
    return (int)sc->s.models.size() - 1;
}

int orc_scene_add_instance(orc_scene *sc, uint32_t model, const float xform3x4[12])
{
    if (model >= sc->s.models.size()) return -1;
    Instance in;
    in.model = model;
    memcpy(in.m, xform3x4, sizeof in.m);
    in.identity = false;
    sc->s.inst.push_back(in);
    sc->s.built = false;
    return (int)sc->s.inst.size() - 1;
}

int orc_scene_build(orc_scene *sc)
{
    scene_build(sc->s);
    sc->truth = truth64::Scene();
    for (const Model &m : sc->s.models)
        truth64::add_model(sc->truth, m.verts.empty() ? nullptr : &m.verts[0].position.x, sizeof(rt_vertex) / sizeof(float), m.idx.data(), m.ntris);
    for (const Instance &in : sc->s.inst) truth64::add_instance(sc->truth, in.model, in.m);
    return 0;
}

/* the float64 geometric truth (truth64.h) of every ray: t (float64; -1 on a miss), u, v, prim, inst */
int orc_truth64_trace(const orc_scene *sc, const float *origin_tmin, const float *dir_tmax, size_t n, uint32_t flags,
                      double *t, double *u, double *v, uint32_t *prim, uint32_t *inst, int nthreads)
{
    if (!sc->s.built) return -1;
    if (nthreads < 1) nthreads = 1;
    const bool cull = (flags & RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES) != 0, first = (flags & RT_RAY_FLAG_ACCEPT_FIRST_HIT_AND_END_SEARCH) != 0;
    auto work = [&](int k) {
        for (size_t i = (size_t)k; i < n; i += (size_t)nthreads) {
            const truth64::Hit h = truth64::trace(sc->truth, origin_tmin + 4 * i, origin_tmin[4 * i + 3], dir_tmax + 4 * i, dir_tmax[4 * i + 3], cull, first);
            const bool miss = h.inst == 0xFFFFFFFFu;
            if (t) t[i] = miss ? -1.0 : h.t;
            if (u) u[i] = h.u;
            if (v) v[i] = h.v;
            if (prim) prim[i] = h.prim;
            if (inst) inst[i] = h.inst;
        }
    };
    if (nthreads == 1) { work(0); return 0; }
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; k++) th.emplace_back(work, k);
    for (std::thread &x : th) x.join();
    return 0;
}

static const Bvh *pick(const orc_scene *sc, int which)
{
    if (!sc->s.built) return NULL;
    if (which < 0) return &sc->s.tlas;
    if ((size_t)which >= sc->s.models.size()) return NULL;
    return &sc->s.models[which].blas;
}

/* which = -1: TLAS, else model index */
int orc_scene_bvh_info(const orc_scene *sc, int which, uint32_t *n_prims, uint32_t *n_nodes, uint32_t *max_depth)
{
    const Bvh *b = pick(sc, which);
    if (!b) return -1;
    *n_prims = b->n; *n_nodes = (uint32_t)b->nodes.size(); *max_depth = b->max_depth;
    return 0;
}

int orc_scene_bvh_read(const orc_scene *sc, int which, rt_bvh_node *nodes, uint64_t *keys, uint32_t *parents)
{
    const Bvh *b = pick(sc, which);
    if (!b) return -1;
    if (nodes) memcpy(nodes, b->nodes.data(), sizeof(rt_bvh_node) * b->nodes.size());
    if (keys) memcpy(keys, b->keys.data(), sizeof(uint64_t) * b->keys.size());
    if (parents) memcpy(parents, b->parent.data(), sizeof(uint32_t) * b->parent.size());
    return 0;
}

void orc_set_split_refs(int on) { split_refs_enabled() = on != 0; }

/* the validation boxes of a model's triangles (oracle_bvh.h "Split references"): *n_refs = 0 when no triangle of the model is split;
 * off[n_tris + 1], boxes[6 * n_refs] = lo.xyz hi.xyz per reference, in primitive order */
int orc_scene_refs_info(const orc_scene *sc, uint32_t model, uint32_t *n_refs)
{
    if (!sc->s.built || model >= sc->s.models.size()) return -1;
    *n_refs = (uint32_t)sc->s.models[model].refs.size();
    return 0;
}

int orc_scene_refs_read(const orc_scene *sc, uint32_t model, uint32_t *off, float *boxes)
{
    if (!sc->s.built || model >= sc->s.models.size()) return -1;
    const Model &m = sc->s.models[model];
    if (off) memcpy(off, m.ref_off.data(), sizeof(uint32_t) * m.ref_off.size());
    if (boxes)
        for (size_t k = 0; k < m.refs.size(); k++) {
            const Box &b = m.refs[k];
            const float v[6] = {b.lo.x, b.lo.y, b.lo.z, b.hi.x, b.hi.y, b.hi.z};
            memcpy(boxes + 6 * k, v, sizeof v);
        }
    return 0;
}

int orc_scene_instance_info(const orc_scene *sc, uint32_t inst, float world_box[6], float inv[12])
{
    if (!sc->s.built || inst >= sc->s.inst.size()) return -1;
    const Instance &in = sc->s.inst[inst];
    world_box[0] = in.world.lo.x; world_box[1] = in.world.lo.y; world_box[2] = in.world.lo.z;
    world_box[3] = in.world.hi.x; world_box[4] = in.world.hi.y; world_box[5] = in.world.hi.z;
    memcpy(inv, in.inv, sizeof in.inv);
    return 0;
}

/* mode 0 = brute force over every instance x triangle, 1 = BVH traversal */
int orc_trace(const orc_scene *sc, const float *origin_tmin, const float *dir_tmax, size_t n, uint32_t flags, int mode,
              float *t, float *u, float *v, uint32_t *prim, uint32_t *inst, uint32_t *cnt_nodes, uint32_t *cnt_tris,
              int nthreads)
{
    if (!sc->s.built) return -1;
    if (nthreads < 1) nthreads = 1;
    auto work = [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            Ray r;
            r.o = v3(origin_tmin[4 * i], origin_tmin[4 * i + 1], origin_tmin[4 * i + 2]);
            r.tmin = origin_tmin[4 * i + 3];
            r.d = v3(dir_tmax[4 * i], dir_tmax[4 * i + 1], dir_tmax[4 * i + 2]);
            r.tmax = dir_tmax[4 * i + 3];
            Counters c = {0, 0};
            Hit h = mode == 0 ? trace_brute(sc->s, r, flags) : trace_bvh(sc->s, r, flags, c);
            bool miss = h.inst == RT_NO_HIT;
            if (t) t[i] = miss ? -1.0f : h.t;
            if (u) u[i] = h.u;
            if (v) v[i] = h.v;
            if (prim) prim[i] = h.prim;
            if (inst) inst[i] = h.inst;
            if (cnt_nodes) cnt_nodes[i] = c.nodes;
            if (cnt_tris) cnt_tris[i] = c.tris;
        }
    };
    if (nthreads == 1 || n < 1024) { work(0, n); return 0; }
    std::vector<std::thread> th;
    for (int k = 0; k < nthreads; k++) th.emplace_back(work, n * k / nthreads, n * (k + 1) / nthreads);
    for (std::thread &x : th) x.join();
    return 0;
}

/* ---- one progressive frame ------------------------------------------------ */

int orc_render(const orc_scene *sc, const rt_material_params *mats, uint32_t nmats,
               const float *env_faces, int env_size, const float env_constant[3],
               const rt_per_frame_constants *pfc, uint32_t width, uint32_t height,
               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               uint32_t accum_mode, uint32_t max_radiance_depth, uint32_t max_shadow_depth, int use_brute,
               float *accum, int nthreads, orc_render_stats *stats_out)
{
    if (!sc->s.built || nmats == 0) return -1;
    RenderCtx rc;
    rc.scene = &sc->s;
    rc.mats = mats; rc.nmats = nmats;
    rc.env.faces = env_faces; rc.env.size = env_size; rc.env.seamless = g_cube_seamless != 0;
    for (int k = 0; k < 3; k++) rc.env.constant[k] = env_constant ? env_constant[k] : 0.0f;
    rc.pfc = *pfc;
    rc.width = width; rc.height = height;
    rc.max_radiance_depth = max_radiance_depth;
    rc.max_shadow_depth = max_shadow_depth;
    rc.use_brute = use_brute == 1;
    if (use_brute == 2) { rc.truth = truth_hook; rc.truth_scene = &sc->truth; }      /* (2: the float64 geometric truth as the tracer) */
    if (x1 > width) x1 = width;
    if (y1 > height) y1 = height;
    if (nthreads < 1) nthreads = 1;
    std::vector<PixelStats> st(nthreads);
    auto work = [&](int k) {
        PixelStats &ps = st[k];
        memset(&ps, 0, sizeof ps);
        uint32_t rows = y1 > y0 ? y1 - y0 : 0;
        uint32_t ya = y0 + (uint32_t)((uint64_t)rows * k / nthreads);
        uint32_t yb = y0 + (uint32_t)((uint64_t)rows * (k + 1) / nthreads);
        for (uint32_t y = ya; y < yb; y++)
            for (uint32_t x = x0; x < x1; x++) {
                PixelCtx pc = { &rc, x, y, &ps };
                float cur[4];
                if (!rayGen(pc, cur)) continue;
                accumulate(accum + ((size_t)y * width + x) * 4, cur, pfc->cameraParams.accumCount, accum_mode);
            }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int k = 0; k < nthreads; k++) th.emplace_back(work, k);
        for (std::thread &x : th) x.join();
    }
    if (stats_out) {
        memset(stats_out, 0, sizeof *stats_out);
        for (const PixelStats &p : st) {
            stats_out->rays_primary += p.rays_primary;
            stats_out->rays_secondary += p.rays_secondary;
            stats_out->rays_shadow += p.rays_shadow;
            stats_out->primary_hits += p.primary_hits;
            stats_out->secondary_hits += p.secondary_hits;
            stats_out->nodes += p.nodes;
            stats_out->tris += p.tris;
            stats_out->shaded_hits += p.shaded_hits;
        }
    }
    return 0;
}

/* ---- N2: one RealtimeRaytracingPipeline frame (two AOV images, no accumulation) -------- */

int orc_render_realtime(const orc_scene *sc, const rt_material_params *mats, uint32_t nmats,
                        const float *env_faces, int env_size, const float env_constant[3],
                        const rt_per_frame_constants *pfc, uint32_t width, uint32_t height,
                        uint32_t max_radiance_depth, uint32_t max_shadow_depth,
                        float *direct, float *indirect, int nthreads, orc_render_stats *stats_out)
{
    if (!sc->s.built || nmats == 0) return -1;
    RenderCtx rc;
    rc.scene = &sc->s;
    rc.mats = mats; rc.nmats = nmats;
    rc.env.faces = env_faces; rc.env.size = env_size; rc.env.seamless = g_cube_seamless != 0;
    for (int k = 0; k < 3; k++) rc.env.constant[k] = env_constant ? env_constant[k] : 0.0f;
    rc.pfc = *pfc;
    rc.width = width; rc.height = height;
    rc.max_radiance_depth = max_radiance_depth;
    rc.max_shadow_depth = max_shadow_depth;
    rc.use_brute = false;
    if (nthreads < 1) nthreads = 1;
    std::vector<PixelStats> st(nthreads);
    auto work = [&](int k) {
        PixelStats &ps = st[k];
        memset(&ps, 0, sizeof ps);
        uint32_t ya = (uint32_t)((uint64_t)height * k / nthreads), yb = (uint32_t)((uint64_t)height * (k + 1) / nthreads);
        for (uint32_t y = ya; y < yb; y++)
            for (uint32_t x = 0; x < width; x++) {
                PixelCtx pc = { &rc, x, y, &ps };
                rayGenRealtime(pc, direct + ((size_t)y * width + x) * 4, indirect + ((size_t)y * width + x) * 4);
            }
    };
    if (nthreads == 1) work(0);
    else {
        std::vector<std::thread> th;
        for (int k = 0; k < nthreads; k++) th.emplace_back(work, k);
        for (std::thread &x : th) x.join();
    }
    if (stats_out) {
        memset(stats_out, 0, sizeof *stats_out);
        for (const PixelStats &p : st) {
            stats_out->rays_primary += p.rays_primary; stats_out->rays_secondary += p.rays_secondary;
            stats_out->rays_shadow += p.rays_shadow; stats_out->primary_hits += p.primary_hits;
            stats_out->secondary_hits += p.secondary_hits; stats_out->nodes += p.nodes; stats_out->tris += p.tris;
            stats_out->shaded_hits += p.shaded_hits;
        }
    }
    return 0;
}

/* ---- N3: DenoiseCompositor::dispatch (src/DenoiseCompositor.cpp:109-148): H pass then V pass ---- */

int orc_denoise(const float *direct, const float *indirect, uint32_t width, uint32_t height, const void *params24,
                float *out_h, float *out_v, int nthreads)
{
    DenoiseParams P;
    memcpy(&P, params24, sizeof P);
    if (P.maxKernelSize < 0 || P.maxKernelSize > 20) return -1;
    Img D = { direct, (int)width, (int)height }, I = { indirect, (int)width, (int)height }, H = { out_h, (int)width, (int)height };
    if (nthreads < 1) nthreads = 1;
    for (int pass = 0; pass < 2; pass++) {
        auto work = [&](int k) {
            int ya = (int)((uint64_t)height * k / nthreads), yb = (int)((uint64_t)height * (k + 1) / nthreads);
            for (int y = ya; y < yb; y++)
                for (int x = 0; x < (int)width; x++)
                    denoisePixel(pass, P, x, y, D, pass == 0 ? I : H, (pass == 0 ? out_h : out_v) + ((size_t)y * width + x) * 4);
        };
        std::vector<std::thread> th;
        for (int k = 0; k < nthreads; k++) th.emplace_back(work, k);
        for (std::thread &x : th) x.join();
    }
    return 0;
}

/* ---- host logic ----------------------------------------------------------- */

static inline V3 host_normalize(V3 v)      /* XMVector3Normalize: v / length */
{
    float l = sqrtf(dot3(v, v));
    return vdivs(v, l);
}

/* BaseCamera::SetLookDirection (Camera.cpp:19-36): forward/up -> orthonormal
 * forward and up as GetForwardVec()/GetUpVec() later return them. */
void orc_camera_look(const float eye[3], const float at[3], const float up_in[3], float fwd_out[3], float up_out[3])
{
    V3 f = host_normalize(vsub(v3(at[0], at[1], at[2]), v3(eye[0], eye[1], eye[2])));
    V3 r = host_normalize(cross3(f, v3(up_in[0], up_in[1], up_in[2])));
    V3 u = cross3(r, f);
    fwd_out[0] = f.x; fwd_out[1] = f.y; fwd_out[2] = f.z;
    up_out[0] = u.x; up_out[1] = u.y; up_out[2] = u.z;
}

/* calculateCameraVariables, ProgressiveRaytracingPipeline.cpp:151-168 */
void orc_camera_basis(const float forward[3], const float up[3], float fov, float aspect,
                      float U[4], float V[4], float W[4])
{
    V3 w = v3(forward[0], forward[1], forward[2]);
    float wlen = sqrtf(dot3(w, w));
    V3 u = host_normalize(cross3(w, v3(up[0], up[1], up[2])));
    V3 v = host_normalize(cross3(u, w));
    float vlen = wlen * tanf(0.5f * fov);
    float ulen = vlen * aspect;
    u = vscale(u, ulen);
    v = vscale(v, vlen);
    U[0] = u.x; U[1] = u.y; U[2] = u.z; U[3] = 0.0f;
    V[0] = v.x; V[1] = v.y; V[2] = v.z; V[3] = 0.0f;
    W[0] = w.x; W[1] = w.y; W[2] = w.z; W[3] = 0.0f;
}

/* mt19937 (the standard's fully specified engine), hand-rolled */
struct Mt { uint32_t s[624]; int i; };
static void mt_seed(Mt &m, uint32_t seed)
{
    m.s[0] = seed;
    for (int i = 1; i < 624; i++) m.s[i] = 1812433253u * (m.s[i - 1] ^ (m.s[i - 1] >> 30)) + (uint32_t)i;
    m.i = 624;
}
static uint32_t mt_next(Mt &m)
{
    if (m.i >= 624) {
        for (int k = 0; k < 624; k++) {
            uint32_t y = (m.s[k] & 0x80000000u) | (m.s[(k + 1) % 624] & 0x7fffffffu);
            m.s[k] = m.s[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        m.i = 0;
    }
    uint32_t y = m.s[m.i++];
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

struct orc_progressive { Mt rng; uint32_t accum; bool have_last; float last_cam[11]; rt_debug_options opt;
                         bool accumulation_enabled, animation_paused; };

orc_progressive *orc_progressive_create(uint32_t rng_seed)
{
    orc_progressive *p = new orc_progressive();
    mt_seed(p->rng, rng_seed);
    p->accum = 0; p->have_last = false;
    p->accumulation_enabled = true; p->animation_paused = true;
    /* ProgressiveRaytracingPipeline.cpp:74-84 */
    memset(&p->opt, 0, sizeof p->opt);
    p->opt.maxIterations = 1024;
    p->opt.cosineHemisphereSampling = 1;
    p->opt.environmentStrength = 1.0f;
    return p;
}
void orc_progressive_destroy(orc_progressive *p) { delete p; }
rt_debug_options *orc_progressive_options(orc_progressive *p) { return &p->opt; }
void orc_progressive_set_flags(orc_progressive *p, int accumulation_enabled, int animation_paused)
{ p->accumulation_enabled = accumulation_enabled != 0; p->animation_paused = animation_paused != 0; }

/* ProgressiveRaytracingPipeline::update, :177-213.  camera = eye[3] at[3] up[3] fov aspect */
void orc_progressive_update(orc_progressive *p, const float camera[11], float elapsedTime, uint32_t elapsedFrames,
                            uint32_t width, uint32_t height, rt_per_frame_constants *out)
{
    if (p->animation_paused) elapsedTime = 142.0f;
    bool moved = !p->have_last || memcmp(p->last_cam, camera, sizeof p->last_cam) != 0;
    if (moved || !p->accumulation_enabled) {
        p->accum = 0;
        memcpy(p->last_cam, camera, sizeof p->last_cam);
        p->have_last = true;
    }
    memset(out, 0, sizeof *out);
    float fwd[3], up[3];
    orc_camera_look(camera, camera + 3, camera + 6, fwd, up);
    rt_camera_params &cp = out->cameraParams;
    cp.worldEyePos.x = camera[0]; cp.worldEyePos.y = camera[1]; cp.worldEyePos.z = camera[2]; cp.worldEyePos.w = 1.0f;
    orc_camera_basis(fwd, up, camera[9], camera[10], &cp.U.x, &cp.V.x, &cp.W.x);
    float r0 = (float)(mt_next(p->rng) >> 8) * (1.0f / 16777216.0f);
    float r1 = (float)(mt_next(p->rng) >> 8) * (1.0f / 16777216.0f);
    cp.jitters.x = (r0 - 0.5f) / (float)width;
    cp.jitters.y = (r1 - 0.5f) / (float)height;
    cp.frameCount = elapsedFrames;
    cp.accumCount = p->accum++;

    float angle = sinf(elapsedTime * 0.2f) * 3.14f * 0.5f;
    float s = sinf(angle), c = cosf(angle);
    const float lx = 0.3f, ly = -0.2f, lz = -1.0f;
    out->directionalLight.forwardDir.x = lx * c + lz * s;
    out->directionalLight.forwardDir.y = ly;
    out->directionalLight.forwardDir.z = lx * (-s) + lz * c;
    out->directionalLight.forwardDir.w = 0.0f;
    out->directionalLight.color.x = 0.9f; out->directionalLight.color.y = 0.9f;
    out->directionalLight.color.z = 0.9f; out->directionalLight.color.w = 1.0f;
    out->pointLight.worldPos.x = 0.0f; out->pointLight.worldPos.y = 0.0f;
    out->pointLight.worldPos.z = 0.0f; out->pointLight.worldPos.w = 1.0f;
    out->pointLight.color.x = 0.2f; out->pointLight.color.y = 0.8f;
    out->pointLight.color.z = 0.6f; out->pointLight.color.w = 2.0f;
    out->options = p->opt;
}

}  /* extern "C" */
