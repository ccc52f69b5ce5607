/*
 * dxr_amd.h -- C ABI of the MI355X-native progressive ray tracer.
 *
 * This is the drop-in boundary for ONE path of philcn/DXRExperiments: the
 * ProgressiveRaytracingPipeline (BLAS/TLAS build, traversal, ray-triangle
 * intersection, raygen / closest-hit / miss shading, float accumulation).
 * Each export names the reference interface it stands in for (file:line in
 * the reference tree).  D3D12 objects become opaque handles; the program /
 * state / bindings objects of the reference (RtProgram, RtState, RtBindings)
 * survive only in the C++ wrapper (dxrexperiments_amd/include), because HIP
 * kernels are linked at build time and there is no shader table to fill.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative RT_ERR_* code and never
 *     throws; rt_last_error() returns the message of the calling thread's
 *     most recent failure.
 *   - a context is bound to one GPU and one HIP stream and is not thread
 *     safe; distinct contexts may be used from distinct threads.
 *   - host pointers passed in are copied before the call returns.
 *   - render / trace calls are asynchronous on the context's stream unless
 *     they return data to host memory; rt_context_synchronize() joins.
 *   - there is NO CPU fallback: without a usable HIP device every call that
 *     needs one fails with RT_ERR_HIP.
 */
#ifndef DXR_AMD_H
#define DXR_AMD_H

#include <stddef.h>
#include "dxr_amd_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_context  rt_context;
typedef struct rt_model    rt_model;
typedef struct rt_scene    rt_scene;
typedef struct rt_pipeline rt_pipeline;
typedef struct rt_progressive_host rt_progressive_host;

/* ---- library ------------------------------------------------------------- */

const char *rt_version(void);
const char *rt_last_error(void);
/* number of visible HIP devices, or a negative RT_ERR_* code */
int rt_device_count(void);

/* ---- RtContext (libs/DXRFramework/RtContext.h:15-46) ----------------------- */

/* RtContext::create (RtContext.h:15, RtContext.cpp:12-29): device + queue. */
int rt_context_create(int device, rt_context **out);
/* Same, but work is issued on the caller's hipStream_t (e.g. PyTorch's). */
int rt_context_create_on_stream(int device, void *hip_stream, rt_context **out);
int rt_context_destroy(rt_context *ctx);
/* DeviceResources::WaitForGpu (src/utils/DeviceResources.cpp:559) */
int rt_context_synchronize(rt_context *ctx);
int rt_context_get_stream(rt_context *ctx, void **hip_stream_out);
int rt_context_get_device(rt_context *ctx, int *device_out);
/* Bytes of the traversal kernels' global stack rows the context holds (the part of a walk's stack beyond the LDS rows).  Round 5: rows for
 * the threads of the PERSISTENT launches only, i.e. for what can be resident at once (57 MB on an MI355X for the bench scene, whatever the
 * size of a set; rounds 1 - 4: for every thread of the largest launch, and the one-tile-per-wave primary launch has one per pixel slot --
 * 224 MB per 1080p frame of a set, 4.3 GB for a set of 20); that launch now keeps none and hands the rays that would need one to a small
 * retry launch.  Shared by every pipeline of the context; the bench scenes never touch them.  No reference counterpart:
 * the Fallback Layer keeps its traversal stack inside DispatchRays (RtContext.cpp:218-221). */
int rt_context_get_stack_memory(rt_context *ctx, size_t *bytes);

/* Plain device buffers on the context's GPU (CreateBuffer / AllocateUploadBuffer of
 * Helpers/DirectXRaytracingHelper.h:187-208): for callers that feed device pointers to
 * rt_trace_batch(RT_MEM_DEVICE) or rt_denoiser_dispatch without another runtime. */
int rt_device_alloc(rt_context *ctx, size_t bytes, void **device_ptr);
int rt_device_free(rt_context *ctx, void *device_ptr);
int rt_device_upload(rt_context *ctx, void *device_dst, const void *host_src, size_t bytes);     /* synchronous */
int rt_device_download(rt_context *ctx, void *host_dst, const void *device_src, size_t bytes);   /* synchronous */

/* ---- RtModel (libs/DXRFramework/RtModel.h:13, RtModel.cpp:24-118) ---------- */

/* RtModel::create(ctx, filePath) (RtModel.h:13; the reference imports through Assimp, RtModel.cpp:26-27).  By extension:
 * ".fbx" -> the binary-FBX mesh reader (rt_fbx.cpp: Geometry nodes' Vertices / PolygonVertexIndex / LayerElementNormal,
 * zlib arrays, Lcl transforms of the owning Model; enough for the reference's assets/models/ground.fbx), anything else ->
 * Wavefront OBJ.  See DESIGN.md for the vertex / primitive ordering these readers define. */
int rt_model_create_from_file(rt_context *ctx, const char *path, rt_model **out);
int rt_model_create_from_obj(rt_context *ctx, const char *path, rt_model **out);
/* The arrays RtModel's constructor builds (RtModel.cpp:33-81). */
int rt_model_create_from_arrays(rt_context *ctx, const rt_vertex *verts, uint32_t n_verts,
                                const uint32_t *indices, uint32_t n_tris, rt_model **out);
int rt_model_get_counts(const rt_model *m, uint32_t *n_verts, uint32_t *n_tris);
/* copy the ingested geometry back (host buffers sized by get_counts) */
int rt_model_read_geometry(const rt_model *m, rt_vertex *verts, uint32_t *indices);
int rt_model_retain(rt_model *m);
int rt_model_destroy(rt_model *m);

/* ---- RtScene (libs/DXRFramework/RtScene.h:14-37, RtScene.cpp:18-52) -------- */

int rt_scene_create(rt_context *ctx, rt_scene **out);                     /* RtScene::create  RtScene.h:14 */
/* RtScene::addModel(model, XMMATRIX) (RtScene.h:30): object-to-world as the
 * 3x4 row-major rows the instance desc stores (TopLevelASGenerator.cpp:355-357). */
int rt_scene_add_model(rt_scene *s, rt_model *m, const float transform3x4[12]);
int rt_scene_get_num_instances(const rt_scene *s, uint32_t *n);           /* RtScene.h:32 */
/* RtScene::build(ctx, hitGroupCount) (RtScene.cpp:18-52): BLAS per distinct
 * model (RtModel::build, RtModel.cpp:86-118), then the TLAS, all on the GPU. */
int rt_scene_build(rt_scene *s, uint32_t hit_group_count);
int rt_scene_destroy(rt_scene *s);

/* Inspection of the canonical acceleration structures (tests, tools).
 * which = -1: TLAS; which >= 0: BLAS of the model used by instance `which`. */
int rt_scene_bvh_info(const rt_scene *s, int which, uint32_t *n_prims, uint32_t *n_nodes, uint32_t *max_depth);
int rt_scene_bvh_read(const rt_scene *s, int which, rt_bvh_node *nodes, uint64_t *sorted_keys, uint32_t *parents);
/* world_box: the TLAS leaf box of the instance = the exact box of its triangles' transformed vertices (an identity instance: of its BLAS) */
int rt_scene_instance_info(const rt_scene *s, uint32_t instance, float world_box[6], float world_to_object[12]);
/* Inspection of the PRODUCTION traversal layout of the same structures (tests, tools).  rt_wide_layout_info tells which
 * of the two layouts the library was built with.
 * width 4 (the default), 64-B nodes of 16 words:
 *   w0..w2 float origin.xyz of the node's box, w3 float scale.x | w4 lo.x bytes of children 0..3 (child k in bits 8k..),
 *   w5 hi.x, w6 lo.y, w7 hi.y | w8 lo.z, w9 hi.z, w10 float scale.y, w11 float scale.z | w12..w15 int32 child codes;
 *   children packed at the front, larger surface first; plane = fma(byte, scale, origin); a scale of +inf (bytes 0) marks
 *   an axis that is not quantised (non-finite extents): its planes bound nothing.
 * width 8 (build option -DRT_WIDE=8), 128-B records of 32 words:
 *   w0..w2  float origin.xyz
 *   w3      ex | ey << 8 | ez << 16 | valid << 24: biased exponents of the three power-of-two scales (scale = 2^(e - 127);
 *           e = 255: axis not quantised) and the mask of the slots that hold a child
 *   w4,w5   lo.x bytes of slots 0..3, 4..7 (slot k in bits 8 (k mod 4) ...)    w6,w7   hi.x bytes
 *   w8..w11 lo.y, hi.y likewise        w12..w15 lo.z, hi.z likewise
 *   w16..w23 int32 child codes of slots 0..7
 *   w24     index of the first internal child (the internal children of a node are consecutive, in slot order)
 *   w25     mask of the slots that hold an internal child      w26 builder's binary node      w27..w31 zero
 *   Slots: the child nearest the (-,-,-) corner of the node sits in slot 0, nearest (+,+,+) in slot 7 (bit a of the slot =
 *   axis a positive); the traversal visits hit children in the order of slot XOR (sign bits of the ray direction).
 * Child codes: >= 0 node index, INT32_MIN unused, otherwise ~code with code = instance (TLAS) or
 * first_record << 3 | (count - 1) (BLAS).  BLAS records are 48 B: nine floats p0 p1 p2, the uint32 primitive index, 8 B
 * padding.  root_code: 0 = node 0, negative = the whole structure is one leaf. */
int rt_wide_layout_info(uint32_t *width, uint32_t *node_bytes);
int rt_scene_wide_info(const rt_scene *s, int which, uint32_t *n_nodes, int32_t *root_code, uint32_t *n_records);
int rt_scene_wide_read(const rt_scene *s, int which, void *nodes, void *records);
/* test / experiment hook: overwrite the node array with a RENUMBERING of itself (same count; node 0 stays the root and the
 * first nodes the breadth-first top the traversal keeps in LDS).  Results do not depend on node numbers. */
int rt_debug_wide_write(rt_scene *s, int which, const void *nodes, uint32_t n_nodes);
/* Split references (round 5; dxrexperiments_amd/csrc/rt_refs.h, defined in oracle/oracle_bvh.h): a long thin triangle whose AABB is more than
 * eight times its own surface is held by the traversal layout as up to 128 references, each with the box of the part of the triangle inside
 * one slab of its longest axis, and a candidate hit on it is accepted only if one of those boxes passes the slab test over [tmin, t] (an
 * unsplit triangle: its own AABB, the rule of rounds 1 - 4).  rt_scene_refs_info: triangles of instance `which`'s model and its boxes in all
 * (0: no triangle is split); rt_scene_refs_read: ref_off[n_tris + 1], the boxes by primitive (6 floats each) and the box of every record of
 * the traversal layout (rt_scene_wide_read's records; meaningful where a record's third word of c says 1).  No reference counterpart: the
 * Fallback Layer's builder is not in the checkout (libs/DXRFramework/Helpers/BottomLevelASGenerator.cpp:333 only calls it). */
int rt_scene_refs_info(const rt_scene *s, int which, uint32_t *n_tris, uint32_t *n_refs);
int rt_scene_refs_read(const rt_scene *s, int which, uint32_t *ref_off, float *ref_boxes, float *record_boxes);
/* milliseconds the last rt_scene_build spent on the GPU (BLAS + TLAS) */
int rt_scene_build_ms(const rt_scene *s, float *ms);

/* ---- raw TraceRay over a batch (HLSL TraceRay semantics, used by tests and
 *      the traversal benchmark; ProgressiveRaytracing.hlsl:34,53,
 *      RaytracingCommon.hlsli:94) ------------------------------------------- */

#define RT_MEM_HOST   0u
#define RT_MEM_DEVICE 1u
#define RT_TRACE_FAST       0u   /* production kernel */
#define RT_TRACE_CANONICAL  1u   /* reference-order kernel, also fills the counters */

/* origin_tmin / dir_tmax: n x float4.  Outputs (each may be NULL): t (-1 on
 * miss), u, v, prim, inst (RT_NO_HIT on miss), nodes / tris traversal
 * counters (RT_TRACE_CANONICAL only).  mem says where ALL pointers live. */
int rt_trace_batch(rt_context *ctx, const rt_scene *s,
                   const float *origin_tmin, const float *dir_tmax, size_t n,
                   uint32_t ray_flags, uint32_t kernel, uint32_t mem,
                   float *t, float *u, float *v, uint32_t *prim, uint32_t *inst,
                   uint32_t *cnt_nodes, uint32_t *cnt_tris);
/* average GPU milliseconds of the traversal kernel in the last rt_trace_batch */
int rt_trace_last_ms(rt_context *ctx, float *ms);

/* ---- RaytracingPipeline / ProgressiveRaytracingPipeline
 *      (include/RaytracingPipeline.h:8-39, include/ProgressiveRaytracingPipeline.h:19-40,
 *       src/ProgressiveRaytracingPipeline.cpp) ------------------------------- */

#define RT_PIPELINE_PROGRESSIVE 0u
/* RealtimeRaytracingPipeline (include/RealtimeRaytracingPipeline.h:15-74, src/RealtimeRaytracingPipeline.cpp,
 * assets/shaders/RealtimeRaytracing.hlsl): same traversal and lights, one Phong-lobe bounce, no accumulation,
 * two outputs: 0 = direct lighting, 1 = indirect specular (the DenoiseCompositor's inputs) */
#define RT_PIPELINE_REALTIME    1u

int rt_pipeline_create(rt_context *ctx, uint32_t kind, rt_pipeline **out);   /* ::create  ProgressiveRaytracingPipeline.h:19 */
int rt_pipeline_destroy(rt_pipeline *p);
const char *rt_pipeline_get_name(const rt_pipeline *p);                      /* getName   :40 */
int rt_pipeline_set_scene(rt_pipeline *p, rt_scene *s);                      /* setScene  .cpp:93-97 */
int rt_pipeline_add_material(rt_pipeline *p, const rt_material_params *m);   /* addMaterial .h:30; material i <-> instance i */
int rt_pipeline_set_material(rt_pipeline *p, uint32_t index, const rt_material_params *m);
/* loadResources (.cpp:104-125): the environment cube map (t1/space2).  faces =
 * 6 x size x size RGBA float32, D3D face order +X -X +Y -Y +Z -Z. */
int rt_pipeline_set_environment_cube(rt_pipeline *p, const float *faces_rgba32f, uint32_t size);
int rt_pipeline_set_environment_constant(rt_pipeline *p, const float rgb[3]);
/* RT_CUBE_SEAMLESS (default: what TextureCube.SampleLevel does on any D3D12 device, RaytracingCommon.hlsli:152) or
 * RT_CUBE_FACE_CLAMP */
int rt_pipeline_set_environment_filter(rt_pipeline *p, uint32_t filter);
/* DDS cube map, DXGI_FORMAT_R16G16B16A16_FLOAT or R32G32B32A32_FLOAT, mip 0 used */
int rt_pipeline_load_environment_dds(rt_pipeline *p, const char *path);
/* createOutputResource(format, w, h) (.cpp:127-149); accumulation is always fp32,
 * `format` selects what rt_pipeline_read_output converts to. */
int rt_pipeline_create_output(rt_pipeline *p, uint32_t format, uint32_t width, uint32_t height);
/* Render into caller-owned device memory (w*h*4 floats), e.g. a torch tensor. */
int rt_pipeline_bind_output(rt_pipeline *p, void *device_rgba32f, uint32_t width, uint32_t height);
int rt_pipeline_build_acceleration_structures(rt_pipeline *p);               /* .cpp:99-102 */
/* MAX_RADIANCE_RAY_DEPTH / MAX_SHADOW_RAY_DEPTH (RaytracingCommon.hlsli:11-12); defaults 1, 2 (the values the
 * reference compiles in).  The radiance depth may be raised to 4 (specular chains, BASELINE config 5);
 * more -> RT_ERR_UNSUPPORTED.  The shadow depth is unbounded (levels past the radiance depth cast none). */
int rt_pipeline_set_depth_limits(rt_pipeline *p, uint32_t max_radiance_depth, uint32_t max_shadow_depth);
int rt_pipeline_set_accumulation_mode(rt_pipeline *p, uint32_t mode);        /* RT_ACCUM_* */
/* The reference's accumulation STORAGE: its output texture is DXGI_FORMAT_R16G16B16A16_FLOAT (src/DXRExperimentsApp.cpp:28 ->
 * src/ProgressiveRaytracingPipeline.cpp:127-131) and RayGen read-modify-writes it every frame (assets/shaders/ProgressiveRaytracing.hlsl:36-38):
 * the running mean is read as fp16, formed in fp32 and rounded back to fp16 by the store, every frame.
 *   format = RT_FORMAT_R32G32B32A32_FLOAT (default; north_star's "float accumulation buffer"): the mean stays fp32 from frame to frame, and an
 *            output created as RGBA16F is only converted when it is read;
 *   format = RT_FORMAT_R16G16B16A16_FLOAT: every frame's mean is rounded to fp16 (rounding = RT_ROUND_NEAREST_EVEN, or RT_ROUND_TOWARD_ZERO --
 *            the D3D11 functional spec's rule for float -> lower-precision float; hardware differs, parity unpinned) before the next frame
 *            reads it.  The buffer stays 16 B per pixel (values exactly representable in fp16); RT_ACCUM_SUM images are not rounded.
 * Flushes a deferred set; applies from the next frame. */
int rt_pipeline_set_accumulation_storage(rt_pipeline *p, uint32_t format, uint32_t rounding);
/* evaluateDirectionalLight / evaluatePointLight trace their shadow ray even when N.L == 0 (RaytracingCommon.hlsli:126-147) and
 * then multiply the visibility by that zero.  off (default): they are traversed like every other ray, as in the reference.
 * on: such rays are emitted and counted (rt_stats.rays_shadow) but not traversed (rt_stats.rays_shadow_skipped); the image is
 * bit-identical either way.  Measured: a third of the shadow rays of the bench scene, but they end within a few steps inside
 * the closed geometry they start in, so the frame time does not move; -7 % on the single-sided 10 M-triangle terrain. */
int rt_pipeline_set_skip_unlit_shadow_rays(rt_pipeline *p, int on);
/* The shadow cache: a light buffer of occluders for the any-hit (shadow) searches.  TraceRay with
 * ACCEPT_FIRST_HIT_AND_END_SEARCH (assets/shaders/RaytracingCommon.hlsli:84-96) only asks whether ANY triangle lies between the
 * point and the light; the triangle that last answered that for rays at the same place in light space (the cell of the origin
 * projected along the directional light, the cube-map texel of the direction from the point light) is tested first, and the
 * walk only starts if it does not occlude.  The visibility -- and the image -- are the same bit for bit; the time is not
 * (bench scene: shadow stage -19 %).  cells_per_side: -1 automatic (by triangle count; the option shadow_cache_res overrides),
 * 0 off, else 16..8192 (the table takes 10 * cells^2 bytes, 20 * cells^2 for scenes of several instances). */
int rt_pipeline_set_shadow_cache(rt_pipeline *p, int cells_per_side);
/* cells per side the last rendered frame used (0: it ran without the cache -- AO view, or switched off) */
int rt_pipeline_get_shadow_cache(const rt_pipeline *p, int *cells_per_side);
/* The free sphere around the point light: shootShadowRay towards it (RaytracingCommon.hlsli:84-96, :133-147) ends where no
 * geometry can lie any more -- a lower bound of the light's distance from the nearest triangle, found by a device pass the first
 * frame with a new (scene, light position) queues behind itself and later frames pick up when it has landed; the visibility is
 * the same bit for bit.  Returns the radius in use for the light position of the last update() (0: none known yet, or none
 * possible: the light touches geometry).  Never waits.  Env RT_FREE_RADIUS=0 switches the sphere off. */
int rt_pipeline_get_free_sphere(rt_pipeline *p, float *radius);
int rt_pipeline_clear_output(rt_pipeline *p);
/* update(): the 188-byte constant buffer the reference fills each frame (.cpp:177-213) */
int rt_pipeline_update(rt_pipeline *p, const rt_per_frame_constants *constants);
/* render(cmdList, frameIndex, w, h) (.cpp:215-247): one 1-spp progressive frame. */
int rt_pipeline_render(rt_pipeline *p, uint32_t width, uint32_t height);
/* n frames = n x { update(&constants[i]); render(w, h) } with the same bits in the output, rendered through SHARED sets of
 * launches (up to 32 frames each): the sample-batch mode of BASELINE configs[2] (256 spp accumulated).  The reference issues one
 * DispatchRays per frame (src/ProgressiveRaytracingPipeline.cpp:188-195, :244) and RayGen folds each frame into gOutput
 * with that frame's accumCount (assets/shaders/ProgressiveRaytracing.hlsl:36-38); here the rays of the frames of a batch
 * share the ray queues (a slot's frame selects constants, lights and RNG seed), so the persistent traversal launches and
 * their tails are paid once per batch, and the resolve kernel folds a pixel's frames in frame order -- the running mean is
 * the one n single frames leave.  Frames with accumCount >= maxIterations are skipped (:14-16).  Progressive pipeline only;
 * afterwards the pipeline's constants are constants[n - 1].  Queue memory grows with the batch (1080p: at most 0.47 GB per frame;
 * rt_pipeline_set_queue_budget). */
int rt_pipeline_render_batch(rt_pipeline *p, uint32_t width, uint32_t height, const rt_per_frame_constants *constants, uint32_t n);
/* Reserves everything a set of `frames` frames of width x height allocates -- ray / hit / shadow queues, the set's constants, the
 * shadow cache's table and (scene built) the traversal kernels' global stack rows -- now, so that the first set of that size does
 * not allocate (the counterpart of createOutputResource for the per-frame work memory the reference's Fallback Layer keeps inside
 * DispatchRays).  width x height may be the packed rows of a rank's bands.  Optional: all of it also grows on demand. */
int rt_pipeline_reserve_batch(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t frames);
/* same, restricted to pixel rectangle [x0,x1) x [y0,y1) (tile sharding) */
int rt_pipeline_render_tile(rt_pipeline *p, uint32_t width, uint32_t height,
                            uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
/* same, restricted to the interleaved row bands {b : b mod world == rank} of band_rows rows each (rt_tile_bands), all of
 * them in ONE set of launches: the per-frame call of a tile-partitioned multi-GPU run.  band_rows must be a multiple of 8. */
int rt_pipeline_render_bands(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world);
/* rt_pipeline_render_batch restricted to the rank's bands: n frames of a tile-partitioned run through shared sets of launches
 * (BASELINE configs[4]: at 8 ranks a band set of ONE 4K frame is an eighth of a frame per persistent launch; sets of frames give
 * the launches back their length).  Pixels are seeded by their global index (assets/shaders/ProgressiveRaytracing.hlsl:89), so
 * the rows equal the same rows of n whole frames bit for bit. */
int rt_pipeline_render_bands_batch(rt_pipeline *p, uint32_t width, uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world,
                                   const rt_per_frame_constants *constants, uint32_t n);
/* Sets of frames BEHIND the reference's own per-frame calls (src/DXRExperimentsApp.cpp:162-165, :194: one update() and one
 * render() per frame; accumulation order src/ProgressiveRaytracingPipeline.cpp:188-195).  With max_frames > 1, rt_pipeline_render
 * records the frame's constants and returns; the recorded frames go through ONE set of launches (rt_pipeline_render_batch:
 * bit for bit the image of rendering each frame at once) as soon as max_frames (<= 32) have gathered, or when a call reads what
 * they produce (read_output, get_output_device_ptr, checkpoints, statistics, the counting re-walks), changes what they would
 * see (scene, materials, environment, output, depth limits, accumulation mode, clear), or synchronises the context.  Errors of
 * a deferred frame surface at the call that flushes it.  Progressive pipeline only; 0 or 1 = render() renders (the default). */
int rt_pipeline_set_deferred(rt_pipeline *p, uint32_t max_frames);
int rt_pipeline_get_deferred(const rt_pipeline *p, uint32_t *max_frames, uint32_t *pending);      /* either may be NULL */
int rt_pipeline_flush(rt_pipeline *p);
/* Queue memory.  A set of launches reserves the worst case of its ray / hit / shadow queues up front when that fits `bytes`
 * (0: the default, a quarter of the device's memory, or the option queue_budget_mb); above it every radiance level is sized by the
 * count the compaction before it has produced (one 4-byte read-back per level and set).  get: bytes reserved now, and whether
 * the last set of launches sized its levels by count. */
int rt_pipeline_set_queue_budget(rt_pipeline *p, size_t bytes);
int rt_pipeline_get_queue_memory(rt_pipeline *p, size_t *bytes_reserved, uint32_t *sized_by_count);
int rt_pipeline_get_num_outputs(const rt_pipeline *p, int *n);               /* getNumOutputs .h:34 */
int rt_pipeline_get_output_device_ptr(rt_pipeline *p, uint32_t id, void **ptr); /* getOutputResource .h:35 */
/* synchronises; host buffer is w*h*4 floats (RGBA32F) or halfs (RGBA16F) */
int rt_pipeline_read_output(rt_pipeline *p, void *host, size_t bytes);
int rt_pipeline_read_output_n(rt_pipeline *p, uint32_t id, void *host, size_t bytes);   /* output id < getNumOutputs() */
/* the inverse of rt_pipeline_read_output for the fp32 accumulation image (w*h*16 bytes): resume of an accumulation */
int rt_pipeline_write_output(rt_pipeline *p, const void *host_rgba32f, size_t bytes);
/* Accumulation checkpoint (the reference loses its accumulation on exit): the fp32 accumulation
 * image plus the host-side state update() carries from frame to frame (mAccumCount, last camera, options, flags,
 * RNG) in one file.  Loading it into a pipeline with an output of the same size and continuing reproduces the
 * uninterrupted run bit for bit.  `h` may be NULL (image only). */
int rt_pipeline_save_checkpoint(rt_pipeline *p, const rt_progressive_host *h, const char *path);
int rt_pipeline_load_checkpoint(rt_pipeline *p, rt_progressive_host *h, const char *path);
/* synchronises; stage timings need rt_pipeline_enable_timing(p, frames > 0): HIP events
 * are recorded around every stage kernel on the context stream, in a ring that
 * remembers the last `frames` frames */
int rt_pipeline_get_stats(rt_pipeline *p, rt_stats *out);
int rt_pipeline_enable_timing(rt_pipeline *p, int frames);
/* sums over every frame since rt_pipeline_reset_totals (ray counts exact; stage times
 * summed over the frames still in the timing ring, out->frames says how many) */
int rt_pipeline_get_totals(rt_pipeline *p, rt_stats *out);
int rt_pipeline_reset_totals(rt_pipeline *p);
/* Re-traces the ray queues of the LAST rendered frame with the canonical traversal and
 * returns its node / triangle counters per stage: out[RT_STAGE_COUNT].  These are the
 * ALGORITHMIC-bytes inputs of the roofline (32 B per node, 36 B per triangle, 48 B per ray). */
int rt_pipeline_count_work(rt_pipeline *p, rt_stage_work *out);
/* Re-walks the ray queues of the LAST rendered frame with a counting instantiation of the PRODUCTION traversal
 * (same tree, same order, same early exits) and returns what it fetched per stage: out[RT_STAGE_COUNT].  Outputs
 * of the frame are rewritten with identical values.  Input of the L2-request roofline. */
int rt_pipeline_count_walk(rt_pipeline *p, rt_stage_walk *out);
/* per-pixel primary-hit records of the last render (tests): w*h each, may be NULL */
int rt_pipeline_read_primary_hits(rt_pipeline *p, float *t, uint32_t *prim, uint32_t *inst);

/* ---- host-side frame logic (src/ProgressiveRaytracingPipeline.cpp:151-213,
 *      libs/MiniEngine/Camera.cpp:19-36) --------------------------------------- */

/* camera = eye[3] at[3] up[3] vfov aspect(width/height) */
int rt_camera_look(const float eye[3], const float at[3], const float up[3], float forward_out[3], float up_out[3]);
int rt_camera_basis(const float forward[3], const float up[3], float vfov, float aspect,
                    float U[4], float V[4], float W[4]);                    /* calculateCameraVariables :151-168 */
int rt_progressive_host_create(uint32_t rng_seed, rt_progressive_host **out);
int rt_progressive_host_destroy(rt_progressive_host *h);
int rt_progressive_host_options(rt_progressive_host *h, rt_debug_options **options);   /* mShaderDebugOptions :74-84 */
int rt_progressive_host_set_flags(rt_progressive_host *h, int accumulation_enabled, int animation_paused);
int rt_progressive_host_reset(rt_progressive_host *h);                      /* frameDirty :309-311 */
/* serialised host state (see rt_pipeline_save_checkpoint): call with buf = NULL to get the size in *bytes */
int rt_progressive_host_save_state(const rt_progressive_host *h, void *buf, size_t capacity, size_t *bytes);
int rt_progressive_host_load_state(rt_progressive_host *h, const void *buf, size_t bytes);
int rt_progressive_host_update(rt_progressive_host *h, const float camera[11], float elapsed_time,
                               uint32_t elapsed_frames, uint32_t width, uint32_t height,
                               rt_per_frame_constants *out);                /* update :177-213 */

/* RealtimeRaytracingPipeline::update (src/RealtimeRaytracingPipeline.cpp:168-199): as above but accumCount = 0 and
 * options = { environmentStrength 1, rest 0 } */
int rt_realtime_host_update(rt_progressive_host *h, const float camera[11], float elapsed_time,
                            uint32_t elapsed_frames, uint32_t width, uint32_t height, rt_per_frame_constants *out);

/* ---- DenoiseCompositor (include/DenoiseCompositor.h:5-59, src/DenoiseCompositor.cpp,
 *      assets/shaders/BilateralFilter.hlsli, DenoiseCommon.hlsli): separable joint-bilateral filter of the
 *      indirect-specular AOV guided by the direct-lighting AOV, then composite + exposure + Reinhard + gamma ---- */
typedef struct rt_denoiser rt_denoiser;
typedef struct rt_denoiser_params {       /* cbuffer Params, DenoiseCommon.hlsli:18-26 */
    float    exposure;                    /* defaults src/DenoiseCompositor.cpp:44-49: 1.0 */
    float    gamma;                       /* 2.2 */
    uint32_t tonemap;                     /* 1 */
    uint32_t gammaCorrect;                /* 0 */
    int32_t  maxKernelSize;               /* 12; taps on each side, <= 20 (the reference's LDS cache extent) */
    uint32_t debugVisualize;              /* 0 composite, 1 denoised only, 2 input, 3 joint */
} rt_denoiser_params;
int rt_denoiser_create(rt_context *ctx, rt_denoiser **out);                                    /* ::create  DenoiseCompositor.h:10 */
int rt_denoiser_destroy(rt_denoiser *d);
int rt_denoiser_get_params(rt_denoiser *d, rt_denoiser_params **params);                       /* mConstantBuffer */
int rt_denoiser_create_output(rt_denoiser *d, uint32_t format, uint32_t width, uint32_t height); /* createOutputResource .cpp:70-93 */
/* dispatch(cmdList, {directLightingSrv, indirectSpecularSrv}, frameIndex, w, h) (.cpp:109-148): inputs are device
 * RGBA32F images, e.g. outputs 0 and 1 of the realtime pipeline */
int rt_denoiser_dispatch(rt_denoiser *d, const void *direct_lighting, const void *indirect_specular, uint32_t width, uint32_t height);
int rt_denoiser_get_output_device_ptr(rt_denoiser *d, void **ptr);                             /* getOutputResource .h:23 */
int rt_denoiser_read_output(rt_denoiser *d, void *host, size_t bytes);                         /* final (pass V) image */
int rt_denoiser_read_intermediate(rt_denoiser *d, void *host, size_t bytes);                   /* pass H image (tests) */
int rt_denoiser_last_ms(rt_denoiser *d, float *ms);

/* ---- multi-GPU: the reference drives one device (src/DXRExperimentsApp.cpp:107-130); a multi-GPU caller is
 *      one such process PER GPU (scene and acceleration structures replicated) and exactly one collective, issued on the
 *      context's stream through RCCL (librccl.so is opened on first use).  Create the processes before any of them
 *      touches the GPU.  examples/progressive_multi.cpp is the worked example; DESIGN.md section 5 the cost model. ---- */
/* host-side partitions (no device needed) */
/* frames {f < n_frames : f mod world == rank} (partition A: sample batches, rendered with RT_ACCUM_SUM) */
int rt_shard_frame_count(uint32_t rank, uint32_t world, uint32_t n_frames, uint32_t *count);
/* interleaved row bands {b : b mod world == rank}, band b = rows [b*band_rows, min((b+1)*band_rows, height)) (partition B:
 * image tiles for rt_pipeline_render_tile); y0 / y1 may be NULL to count */
int rt_tile_bands(uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t *y0, uint32_t *y1, uint32_t capacity,
                  uint32_t *n_bands);
/* what rt_dist_gather_bands sends per rank: ceil(bands / world) band slots of band_rows x width RGBA32F texels */
int rt_tile_gather_layout(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world, uint32_t *slots_per_rank, size_t *floats_per_rank);
typedef struct rt_dist rt_dist;
int rt_dist_get_unique_id(void *id128);                                   /* ncclGetUniqueId: rank 0 makes it, the launcher hands it round */
/* ncclCommInitRank on ctx's device.  Before that, the ranks of one node compare the PCI bus ids of their devices through
 * files in /dev/shm named after the id (RT_DIST_CHECK_SECONDS, default 20, 0 = off): two ranks on one device -- where RCCL
 * would fail late or hang -- are RT_ERR_INVALID_ARG here.  (The reference runs on one adapter, NodeMask 0:
 * libs/DXRFramework/RtContext.cpp:37.) */
int rt_dist_create(rt_context *ctx, int rank, int world, const void *id128, rt_dist **out);
int rt_dist_destroy(rt_dist *d);
int rt_dist_get_rank(const rt_dist *d, int *rank, int *world);
/* partition A: in-place SUM all-reduce of `count` floats (the RT_ACCUM_SUM image of rt_pipeline_get_output_device_ptr) */
int rt_dist_all_reduce_sum(rt_dist *d, void *device_f32, size_t count);
/* partition B: every rank holds its own bands of the width x height RGBA32F image; afterwards every rank holds all of it */
int rt_dist_gather_bands(rt_dist *d, void *device_rgba32f, uint32_t width, uint32_t height, uint32_t band_rows);
/* HIP-event time of the last collective on the context's stream (all-reduce; gather: pack + all-gather + unpack); synchronises */
int rt_dist_last_collective_ms(rt_dist *d, float *ms);
/* "0000:c1:00.0"-style PCI bus id of the context's device: what a launcher prints per rank */
int rt_dist_device_pci_bus_id(const rt_context *ctx, char *out, size_t capacity);

/* ---- image files for host copies of the outputs (the reference only blits to its window,
 *      src/DXRExperimentsApp.cpp:213-214).  rgba32f = width*height float4, row 0 on top. ---------------- */
/* lossless fp32 RGBA OpenEXR (single-part scan-line file, no compression, FLOAT channels A B G R) */
int rt_image_write_exr(const char *path, const float *rgba32f, uint32_t width, uint32_t height);
/* lossless fp32 RGB portable float map */
int rt_image_write_pfm(const char *path, const float *rgba32f, uint32_t width, uint32_t height);
/* 8-bit RGB PNG for viewing: v*exposure, optional Reinhard v/(1+v), v^(1/gamma) (the compositor's display
 * transform, DenoiseCommon.hlsli:29-41, with host libm: not a parity path) */
int rt_image_write_png(const char *path, const float *rgba32f, uint32_t width, uint32_t height,
                       float exposure, float gamma, int tonemap);

/* ---- device math probes (tests): evaluate the kernels' deterministic
 *      sin/cos/exp/log/pow/sqrt/div and samplers on the GPU ------------------- */
int rt_debug_math(rt_context *ctx, int fn, const float *x, const float *y, float *out, size_t n);
int rt_debug_sample(rt_context *ctx, int kind, const uint32_t *seeds, const float *vec3_in, float exponent,
                    float *vec3_out, float *pdf_brdf, uint32_t *seeds_out, size_t n);
int rt_debug_read_secondary_ray(rt_pipeline *p, uint32_t index, float origin_tmin[4], float dir_tmax[4]);   /* tools/longest_walk.py */
/* Experiment and test knobs of a context, in ONE place (round 5; rounds 1 - 4 read 19 environment variables where they were used).
 * Defaults are the measured best; nothing here changes a result bit -- only which kernels and layouts produce it.  The library
 * reads exactly one environment variable for its behaviour, once, in rt_context_create: RT_DEBUG_OPTIONS="name=value,name=value",
 * applied through this function (an unknown name or a value out of range fails the creation).  (rt_dist also reads the
 * launcher's LOCAL_WORLD_SIZE.)  The reference has no counterpart: its knobs are ImGui widgets (src/ProgressiveRaytracingPipeline.cpp:249-312).
 *   lds_top=0|1              traversal kernels read the top of the tree from LDS (1)
 *   lds_stack_rows=0|6|18    6: the small-stack instantiation of the traversal kernels (tests of the overflow path); 0 / 18: default
 *   persistent_blocks_per_cu=0..16   grid of the persistent launches (0: the occupancy API's answer)
 *   fast_bvh=ploc|lbvh       traversal layout collapsed from the PLOC tree (default) or from the canonical LBVH; set before building
 *   build_batch=0..64        PLOC rounds / collapse levels launched between two looks at the device-side state (0: the builders' estimates)
 *   leaf_max=1..8            triangles per collapsed leaf (2)
 *   wide_sah=0|1, sah_node=x, sah_prim=x   surface-area-optimal collapse and its costs (off; 1.0, 0.5)
 *   verbose=0|1              build phases and structure depths on stderr
 *   shadow_cache_res=-1|0|16..8192   cells per side for pipelines that were not told by rt_pipeline_set_shadow_cache (-1: by triangle count)
 *   shadow_cache_pixels=-1|0|1, primary_persistent=-1|0|1   (-1: on for two-level scenes only)
 *   seven_waves_always=0|1   single frames on the sets' kernels
 *   free_radius=0|1          the free sphere around the point light (1)
 *   batch_max=0..32          frames per set of launches (0: 32)
 *   split_refs=0|1           the traversal layout holds long thin triangles as several references (1; rt_scene_refs_info).  Results do not
 *                            depend on it: the candidate rule follows the references either way
 *   fail_ploc_rounds=0|1     (tests) the PLOC layout of every build is thrown away as if its rounds had made no progress (what non-finite boxes
 *                            cause): the LBVH is collapsed instead
 *   repack=0|1               the shadow stage of single-level scenes on the re-packed engine (rt_debug_repack_stats; 0)
 *   primary_retry_cap=n      entries of the retry list behind the one-tile-per-wave primary launch (0: 2^20; tests make it overflow)
 *   queue_budget_mb=n        worst-case queue bytes a set may reserve up front (0: a quarter of the device's memory)
 *   dist_check_seconds=x     how long rt_dist_create waits for the other ranks' device ids (5)
 * Every value but fast_bvh's must be a number in full ("true", "4x", "" are RT_ERR_INVALID_ARG, not 0); RT_DEBUG_OPTIONS items must be name=value. */
int rt_debug_set_option(rt_context *ctx, const char *name, const char *value);
/* option repack=1 (round 6, csrc/rt_trace_repack.h; measured slower, off by default: profiles/r06/repack.txt): the shadow stage of single-level scenes on the
 * engine whose rays change lanes at every phase switch.  Its tallies since the last call: node steps issued, lanes live in them, leaf passes, lanes live in them,
 * rays through the leaf queue, rays through the node queue, refills, watchdog aborts (must be 0).  Synchronises the context. */
int rt_debug_repack_stats(rt_context *ctx, unsigned long long out[8]);
/* test hook: device allocations of more than `bytes` bytes fail with RT_ERR_OOM as if the device were full (0: no limit) */
int rt_debug_set_alloc_limit(size_t bytes);
int rt_debug_sample_cube(rt_context *ctx, const float *faces_rgba32f, uint32_t size, uint32_t filter, const float *dirs, float *out, size_t n);
/* The DDS cube-map reader behind rt_pipeline_load_environment_dds, without a device (tests run it on the reference's
 * own assets/textures/CathedralRadiance.dds): *size = edge length of mip 0; faces_rgba32f (may be NULL to query the
 * size) receives 6 x size x size x 4 floats when capacity_floats is large enough, else RT_ERR_INVALID_ARG. */
int rt_dds_read_cube(const char *path, float *faces_rgba32f, size_t capacity_floats, uint32_t *size);
/* The OBJ reader behind rt_model_create_from_obj, without a device: counts first (verts / indices NULL), then data. */
int rt_obj_read(const char *path, rt_vertex *verts, uint32_t capacity_verts, uint32_t *indices, uint32_t capacity_tris,
                uint32_t *n_verts, uint32_t *n_tris);
/* The binary-FBX reader behind rt_model_create_from_file, likewise */
int rt_fbx_read(const char *path, rt_vertex *verts, uint32_t capacity_verts, uint32_t *indices, uint32_t capacity_tris,
                uint32_t *n_verts, uint32_t *n_tris);

#ifdef __cplusplus
}
#endif
#endif /* DXR_AMD_H */
