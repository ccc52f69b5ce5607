/*
 * oracle.h -- TEST INFRASTRUCTURE ONLY: C entry points of the CPU oracle.
 * See oracle.cpp for the parity status and the reference citations.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/dxr_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_FN_SIN = 0, ORC_FN_COS, ORC_FN_EXP, ORC_FN_LOG, ORC_FN_POW, ORC_FN_SQRT, ORC_FN_DIV, ORC_FN_MIN, ORC_FN_MAX };
enum { ORC_SAMPLE_COS = 0, ORC_SAMPLE_UNIFORM, ORC_SAMPLE_PHONG, ORC_SAMPLE_PERP };

typedef struct orc_scene orc_scene;
typedef struct orc_progressive orc_progressive;

typedef struct orc_render_stats {
    uint64_t rays_primary, rays_secondary, rays_shadow;
    uint64_t primary_hits, secondary_hits;
    uint64_t nodes, tris;       /* traversal counters over every ray of the frame */
    uint64_t shaded_hits;
} orc_render_stats;

uint32_t orc_init_rand(uint32_t v0, uint32_t v1);
float    orc_next_rand(uint32_t *s);
void     orc_round_to_half(const float *x, float *out, size_t n, int nearest);      /* binary32 -> binary16 -> binary32 */
void     orc_math_batch(int fn, const float *x, const float *y, float *out, size_t n);
void     orc_sample_batch(int kind, const uint32_t *seeds, const float *vec3_in, float exponent,
                          float *vec3_out, float *pdf_brdf, uint32_t *seeds_out, size_t n);
void     orc_fresnel(const float I[3], const float N[3], const float f0[3], float out[3]);
void     orc_set_cube_seamless(int on);   /* 1 (default): cross-face bilinear taps; 0: taps clamped to the face */
void     orc_sample_cube(const float *faces, int size, const float *dirs, float *out, size_t n);

int  orc_obj_load(const char *path, rt_vertex **verts, uint32_t *nv, uint32_t **idx, uint32_t *nt);
void orc_free(void *p);

orc_scene *orc_scene_create(void);
void orc_scene_destroy(orc_scene *s);
int  orc_scene_add_model(orc_scene *s, const rt_vertex *verts, uint32_t nv, const uint32_t *idx, uint32_t nt);
int  orc_scene_add_instance(orc_scene *s, uint32_t model, const float xform3x4[12]);
int  orc_scene_build(orc_scene *s);
int  orc_scene_bvh_info(const orc_scene *s, int which, uint32_t *n_prims, uint32_t *n_nodes, uint32_t *max_depth);
int  orc_scene_bvh_read(const orc_scene *s, int which, rt_bvh_node *nodes, uint64_t *keys, uint32_t *parents);
void orc_set_split_refs(int on);     /* scenes built from now on: 0 = no triangle is split (the candidate rule of rounds 1 - 4) */
int  orc_scene_refs_info(const orc_scene *s, uint32_t model, uint32_t *n_refs);
int  orc_scene_refs_read(const orc_scene *s, uint32_t model, uint32_t *off, float *boxes);
int  orc_scene_instance_info(const orc_scene *s, uint32_t inst, float world_box[6], float inv[12]);

int  orc_trace(const orc_scene *s, const float *origin_tmin, const float *dir_tmax, size_t n, uint32_t flags, int mode,
               float *t, float *u, float *v, uint32_t *prim, uint32_t *inst, uint32_t *cnt_nodes, uint32_t *cnt_tris,
               int nthreads);

/* the float64 geometric truth (truth64.h: Moller-Trumbore over every instance x triangle, no box of any kind); t = -1 on a miss */
int  orc_truth64_trace(const orc_scene *s, const float *origin_tmin, const float *dir_tmax, size_t n, uint32_t flags,
                       double *t, double *u, double *v, uint32_t *prim, uint32_t *inst, int nthreads);

/* use_brute: 0 = BVH traversal, 1 = the brute-force loop of oracle_bvh.h, 2 = the float64 geometric truth as the tracer */
int  orc_render(const orc_scene *s, const rt_material_params *mats, uint32_t nmats,
                const float *env_faces, int env_size, const float env_constant[3],
                const rt_per_frame_constants *pfc, uint32_t width, uint32_t height,
                uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                uint32_t accum_mode, uint32_t max_radiance_depth, uint32_t max_shadow_depth, int use_brute,
                float *accum, int nthreads, orc_render_stats *stats);

int  orc_render_realtime(const orc_scene *s, const rt_material_params *mats, uint32_t nmats,
                         const float *env_faces, int env_size, const float env_constant[3],
                         const rt_per_frame_constants *pfc, uint32_t width, uint32_t height,
                         uint32_t max_radiance_depth, uint32_t max_shadow_depth,
                         float *direct, float *indirect, int nthreads, orc_render_stats *stats);
int  orc_denoise(const float *direct, const float *indirect, uint32_t width, uint32_t height, const void *params24,
                 float *out_h, float *out_v, int nthreads);

void orc_camera_look(const float eye[3], const float at[3], const float up_in[3], float fwd_out[3], float up_out[3]);
void orc_camera_basis(const float forward[3], const float up[3], float fov, float aspect,
                      float U[4], float V[4], float W[4]);
orc_progressive *orc_progressive_create(uint32_t rng_seed);
void orc_progressive_destroy(orc_progressive *p);
rt_debug_options *orc_progressive_options(orc_progressive *p);
void orc_progressive_set_flags(orc_progressive *p, int accumulation_enabled, int animation_paused);
void orc_progressive_update(orc_progressive *p, const float camera[11], float elapsedTime, uint32_t elapsedFrames,
                            uint32_t width, uint32_t height, rt_per_frame_constants *out);

#ifdef __cplusplus
}
#endif
#endif
