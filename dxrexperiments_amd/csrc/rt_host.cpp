// rt_host.cpp -- per-frame host logic of the progressive pipeline.
//
// Restates src/ProgressiveRaytracingPipeline.cpp:
//   calculateCameraVariables  :151-168   (U/V/W from forward, up, vfov, aspect)
//   hasCameraMoved            :170-175   (here: bitwise compare of the camera inputs)
//   update                    :177-213   (jitter, frame/accum counters, lights, options)
// and libs/MiniEngine/Camera.cpp:19-36 (SetLookDirection) for the forward/up pair
// GetForwardVec()/GetUpVec() hand to calculateCameraVariables.
// Deviations, all documented in DESIGN.md: the host RNG is seeded explicitly
// (reference: wall clock, :86-88) and a uniform float is (u32 >> 8) * 2^-24
// because std::uniform_real_distribution is implementation-defined;
// mAccumCount starts at 0 (reference: uninitialised, .h:69).
#include <math.h>

#include <new>
#include <random>
#include <sstream>
#include <string>

#include "rt_internal.h"

namespace {

struct H3 { float x, y, z; };
inline H3 sub(H3 a, H3 b) { H3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
inline H3 crossp(H3 a, H3 b)
{
    H3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
inline float len(H3 a)
{
    float d = a.x * a.x;
    d = d + a.y * a.y;
    d = d + a.z * a.z;
    return sqrtf(d);
}
inline H3 unit(H3 a)          // XMVector3Normalize: v / |v|
{
    const float l = len(a);
    H3 r = {a.x / l, a.y / l, a.z / l};
    return r;
}

}  // namespace

struct rt_progressive_host {
    std::mt19937 rng;
    uint32_t accum_count = 0;
    bool have_last = false;
    float last_camera[11];
    rt_debug_options options;
    bool accumulation_enabled = true;   // mFrameAccumulationEnabled, ctor :29
    bool animation_paused = true;       // mAnimationPaused, ctor :30
};

extern "C" {

int rt_camera_look(const float eye[3], const float at[3], const float up[3], float forward_out[3], float up_out[3])
{
    RT_REQUIRE(eye && at && up && forward_out && up_out, "null argument");
    const H3 e = {eye[0], eye[1], eye[2]}, a = {at[0], at[1], at[2]}, u = {up[0], up[1], up[2]};
    const H3 f = unit(sub(a, e));
    const H3 r = unit(crossp(f, u));
    const H3 uu = crossp(r, f);
    forward_out[0] = f.x; forward_out[1] = f.y; forward_out[2] = f.z;
    up_out[0] = uu.x; up_out[1] = uu.y; up_out[2] = uu.z;
    return RT_OK;
}

int rt_camera_basis(const float forward[3], const float up[3], float vfov, float aspect, float U[4], float V[4], float W[4])
{
    RT_REQUIRE(forward && up && U && V && W, "null argument");
    const H3 w = {forward[0], forward[1], forward[2]};       // not normalised: implies focal length (:154)
    const float wlen = len(w);
    H3 u = unit(crossp(w, H3{up[0], up[1], up[2]}));
    H3 v = unit(crossp(u, w));
    const float vlen = wlen * tanf(0.5f * vfov);
    const float ulen = vlen * aspect;
    U[0] = u.x * ulen; U[1] = u.y * ulen; U[2] = u.z * ulen; U[3] = 0.0f;
    V[0] = v.x * vlen; V[1] = v.y * vlen; V[2] = v.z * vlen; V[3] = 0.0f;
    W[0] = w.x; W[1] = w.y; W[2] = w.z; W[3] = 0.0f;
    return RT_OK;
}

int rt_progressive_host_create(uint32_t rng_seed, rt_progressive_host **out)
{
    RT_REQUIRE(out, "null argument");
    rt_progressive_host *h = new (std::nothrow) rt_progressive_host();
    if (!h) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    h->rng.seed(rng_seed);
    memset(&h->options, 0, sizeof h->options);
    h->options.maxIterations = 1024;                 // :74
    h->options.cosineHemisphereSampling = 1;         // :75
    h->options.environmentStrength = 1.0f;           // :83
    *out = h;
    return RT_OK;
}

int rt_progressive_host_destroy(rt_progressive_host *h)
{
    delete h;
    return RT_OK;
}

int rt_progressive_host_options(rt_progressive_host *h, rt_debug_options **options)
{
    RT_REQUIRE(h && options, "null argument");
    *options = &h->options;
    return RT_OK;
}

int rt_progressive_host_set_flags(rt_progressive_host *h, int accumulation_enabled, int animation_paused)
{
    RT_REQUIRE(h, "null argument");
    h->accumulation_enabled = accumulation_enabled != 0;
    h->animation_paused = animation_paused != 0;
    return RT_OK;
}

int rt_progressive_host_reset(rt_progressive_host *h)
{
    RT_REQUIRE(h, "null argument");
    h->have_last = false;          // next update sees a "moved" camera, like mLastCameraVPMatrix = Matrix4() (:309-311)
    return RT_OK;
}

// state = fixed header + the mt19937 engine in its standard text form (operator<<)
namespace {
struct HostStateHeader {
    uint32_t magic, accum_count, have_last, accumulation_enabled, animation_paused, rng_bytes;
    float last_camera[11];
    rt_debug_options options;
};
const uint32_t kHostStateMagic = 0x31534844u;      // "DHS1"
}  // namespace

int rt_progressive_host_save_state(const rt_progressive_host *h, void *buf, size_t capacity, size_t *bytes)
{
    RT_REQUIRE(h && bytes, "null argument");
    std::ostringstream os;
    os << h->rng;
    const std::string rng = os.str();
    HostStateHeader hd;
    memset(&hd, 0, sizeof hd);
    hd.magic = kHostStateMagic;
    hd.accum_count = h->accum_count;
    hd.have_last = h->have_last ? 1u : 0u;
    hd.accumulation_enabled = h->accumulation_enabled ? 1u : 0u;
    hd.animation_paused = h->animation_paused ? 1u : 0u;
    hd.rng_bytes = (uint32_t)rng.size();
    memcpy(hd.last_camera, h->last_camera, sizeof hd.last_camera);
    hd.options = h->options;
    *bytes = sizeof hd + rng.size();
    if (!buf) return RT_OK;
    RT_REQUIRE(capacity >= *bytes, "state buffer too small");
    memcpy(buf, &hd, sizeof hd);
    memcpy((char *)buf + sizeof hd, rng.data(), rng.size());
    return RT_OK;
}

int rt_progressive_host_load_state(rt_progressive_host *h, const void *buf, size_t bytes)
{
    RT_REQUIRE(h && buf, "null argument");
    HostStateHeader hd;
    RT_REQUIRE(bytes >= sizeof hd, "host state truncated");
    memcpy(&hd, buf, sizeof hd);
    RT_REQUIRE(hd.magic == kHostStateMagic && bytes == sizeof hd + (size_t)hd.rng_bytes, "not a host state record");
    std::istringstream is(std::string((const char *)buf + sizeof hd, hd.rng_bytes));
    std::mt19937 rng;
    is >> rng;
    RT_REQUIRE(!is.fail(), "host state: bad RNG record");
    h->rng = rng;
    h->accum_count = hd.accum_count;
    h->have_last = hd.have_last != 0;
    h->accumulation_enabled = hd.accumulation_enabled != 0;
    h->animation_paused = hd.animation_paused != 0;
    memcpy(h->last_camera, hd.last_camera, sizeof h->last_camera);
    h->options = hd.options;
    return RT_OK;
}

int rt_progressive_host_update(rt_progressive_host *h, const float camera[11], float elapsed_time, uint32_t elapsed_frames,
                               uint32_t width, uint32_t height, rt_per_frame_constants *out)
{
    RT_REQUIRE(h && camera && out, "null argument");
    RT_REQUIRE(width > 0 && height > 0, "empty image");
    if (h->animation_paused) elapsed_time = 142.0f;                                   // :179-181
    const bool moved = !h->have_last || memcmp(h->last_camera, camera, sizeof h->last_camera) != 0;
    if (moved || !h->accumulation_enabled) {                                          // :183-186
        h->accum_count = 0;
        memcpy(h->last_camera, camera, sizeof h->last_camera);
        h->have_last = true;
    }
    memset(out, 0, sizeof *out);
    float fwd[3], up[3];
    rt_camera_look(camera, camera + 3, camera + 6, fwd, up);
    rt_camera_params &cp = out->cameraParams;                                         // :188-195
    cp.worldEyePos.x = camera[0]; cp.worldEyePos.y = camera[1]; cp.worldEyePos.z = camera[2]; cp.worldEyePos.w = 1.0f;
    rt_camera_basis(fwd, up, camera[9], camera[10], &cp.U.x, &cp.V.x, &cp.W.x);
    const float xi0 = (float)(h->rng() >> 8) * (1.0f / 16777216.0f);
    const float xi1 = (float)(h->rng() >> 8) * (1.0f / 16777216.0f);
    cp.jitters.x = (xi0 - 0.5f) / (float)width;
    cp.jitters.y = (xi1 - 0.5f) / (float)height;
    cp.frameCount = elapsed_frames;
    cp.accumCount = h->accum_count++;

    // (0.3,-0.2,-1,0) * RotationY(sin(0.2 t) * 3.14 * 0.5)                           // :197-201
    const float angle = sinf(elapsed_time * 0.2f) * 3.14f * 0.5f;
    const float s = sinf(angle), c = cosf(angle);
    rt_float4 &fd = out->directionalLight.forwardDir;
    fd.x = 0.3f * c + -1.0f * s;
    fd.y = -0.2f;
    fd.z = 0.3f * (-s) + -1.0f * c;
    fd.w = 0.0f;
    const rt_float4 dir_color = {0.9f, 0.9f, 0.9f, 1.0f};                             // :14
    const rt_float4 point_color = {0.2f, 0.8f, 0.6f, 2.0f};                           // :13
    const rt_float4 point_pos = {0.0f, 0.0f, 0.0f, 1.0f};                             // :206
    out->directionalLight.color = dir_color;
    out->pointLight.worldPos = point_pos;
    out->pointLight.color = point_color;
    out->options = h->options;                                                        // :210
    return RT_OK;
}

int rt_realtime_host_update(rt_progressive_host *h, const float camera[11], float elapsed_time, uint32_t elapsed_frames,
                            uint32_t width, uint32_t height, rt_per_frame_constants *out)
{
    RT_TRY(rt_progressive_host_update(h, camera, elapsed_time, elapsed_frames, width, height, out));
    h->accum_count = 0;
    out->cameraParams.accumCount = 0;                         // RealtimeRaytracingPipeline.cpp:182
    memset(&out->options, 0, sizeof out->options);
    out->options.environmentStrength = 1.0f;                  // :197
    return RT_OK;
}

}  // extern "C"
