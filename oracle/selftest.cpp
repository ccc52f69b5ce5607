// selftest.cpp -- TEST INFRASTRUCTURE ONLY: exercises every part of the CPU oracle in one small
// program so that it can run under the sanitizers ("CPU oracle under ASan/UBSan"):
//   make -C oracle selftest_asan selftest_tsan && oracle/selftest_asan <cornell.obj> && oracle/selftest_tsan <cornell.obj>
// It compiles oracle.cpp into itself (no shared library), loads the Cornell OBJ, builds the LBVH,
// checks BVH traversal == brute force on a ray batch, renders progressive and realtime frames on two
// threads, runs the denoiser and the host update loop.  Exit code 0 = consistent (and no sanitizer report).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "oracle.h"

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "selftest: %s failed (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s cornell.obj\n", argv[0]); return 2; }
    rt_vertex *verts = nullptr;
    uint32_t *idx = nullptr, nv = 0, nt = 0;
    CHECK(orc_obj_load(argv[1], &verts, &nv, &idx, &nt) == 0 && nv > 0 && nt > 0);

    orc_scene *sc = orc_scene_create();
    CHECK(sc != nullptr);
    CHECK(orc_scene_add_model(sc, verts, nv, idx, nt) == 0);                                            // returns the model index
    const float ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const float moved[12] = {0.5f, 0, 0, 2.5f, 0, 0.5f, 0, 0, 0, 0, 0.5f, 0};
    CHECK(orc_scene_add_instance(sc, 0, ident) == 0 && orc_scene_add_instance(sc, 0, moved) == 1);      // returns the instance index
    CHECK(orc_scene_build(sc) == 0);
    uint32_t np = 0, nn = 0, depth = 0;
    CHECK(orc_scene_bvh_info(sc, 0, &np, &nn, &depth) == 0 && np == nt && nn == 2 * nt - 1);

    // rays from a small LCG: BVH traversal (mode 1) must equal brute force (mode 0) bit for bit
    const size_t N = 4096;
    std::vector<float> O(4 * N), D(4 * N), t0(N), t1(N), u(N), v(N);
    std::vector<uint32_t> p0(N), p1(N), i0(N), i1(N), cn(N), ct(N);
    uint32_t s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f; };
    for (size_t k = 0; k < N; k++) {
        O[4 * k] = 3.0f * rnd(); O[4 * k + 1] = 1.5f * rnd(); O[4 * k + 2] = 3.0f * rnd(); O[4 * k + 3] = 0.0f;
        D[4 * k] = rnd(); D[4 * k + 1] = rnd(); D[4 * k + 2] = rnd(); D[4 * k + 3] = 1.0e38f;
    }
    for (uint32_t flags : {0u, RT_RAY_FLAG_CULL_BACK_FACING_TRIANGLES}) {
        CHECK(orc_trace(sc, O.data(), D.data(), N, flags, 0, t0.data(), u.data(), v.data(), p0.data(), i0.data(), nullptr, nullptr, 2) == 0);
        CHECK(orc_trace(sc, O.data(), D.data(), N, flags, 1, t1.data(), u.data(), v.data(), p1.data(), i1.data(), cn.data(), ct.data(), 2) == 0);
        CHECK(std::memcmp(t0.data(), t1.data(), N * 4) == 0 && p0 == p1 && i0 == i1);
    }

    // host update + two progressive frames, two threads, then realtime + denoise
    const uint32_t W = 48, H = 32;
    rt_material_params mats[2];
    std::memset(mats, 0, sizeof mats);
    for (rt_material_params &m : mats) {
        m.albedo.x = 0.95f; m.albedo.y = 0.05f; m.albedo.w = 1.0f;
        m.specular.x = m.specular.y = m.specular.z = 0.58f; m.specular.w = 1.0f;
        m.roughness = 0.5f; m.reflectivity = 0.7f; m.type = 1;
    }
    const float cam[11] = {0, 0, 3.2f, 0, 0, 0, 0, 1, 0, 0.785398f, (float)W / H};
    const float env[3] = {0.5f, 0.5f, 0.5f};
    orc_progressive *host = orc_progressive_create(1234);
    CHECK(host != nullptr);
    std::vector<float> acc((size_t)W * H * 4, 0.0f), direct(acc.size()), indirect(acc.size()), oh(acc.size()), ov(acc.size());
    rt_per_frame_constants pfc;
    orc_render_stats st;
    for (uint32_t f = 0; f < 2; f++) {
        orc_progressive_update(host, cam, 0.0f, f + 1, W, H, &pfc);
        CHECK(pfc.cameraParams.accumCount == f);
        CHECK(orc_render(sc, mats, 2, nullptr, 0, env, &pfc, W, H, 0, 0, W, H, RT_ACCUM_RUNNING_MEAN, 2, 2, 0, acc.data(), 2, &st) == 0);
        CHECK(st.rays_primary == (uint64_t)W * H && st.rays_shadow > 0);
    }
    for (float x : acc) CHECK(std::isfinite(x));
    CHECK(orc_render_realtime(sc, mats, 2, nullptr, 0, env, &pfc, W, H, 1, 2, direct.data(), indirect.data(), 2, &st) == 0);
    struct { float exposure, gamma; uint32_t tonemap, gammaCorrect; int32_t maxKernelSize; uint32_t debugVisualize; } dp = {1.0f, 2.2f, 1, 1, 12, 0};
    CHECK(orc_denoise(direct.data(), indirect.data(), W, H, &dp, oh.data(), ov.data(), 2) == 0);
    for (float x : ov) CHECK(std::isfinite(x));

    orc_progressive_destroy(host);
    orc_scene_destroy(sc);
    orc_free(verts);
    orc_free(idx);
    std::puts("oracle selftest ok");
    return 0;
}
