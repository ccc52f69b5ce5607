// fuzz_parsers.cpp -- mutation fuzz of the product's host-side parsers and host logic under AddressSanitizer + UBSan (round 5,
// VERDICT r4 task 7).  The files rt_obj.cpp, rt_fbx.cpp, rt_dds.cpp, rt_image.cpp and rt_host.cpp need no device: tests/test_sanitized_parsers.py
// compiles exactly those product sources with `g++ -fsanitize=address,undefined -fno-sanitize-recover=undefined` into this driver
// (no copy of them, no stand-in: only rt_set_error, which lives in rt_api.hip beside the HIP calls, is restated in ten lines) and runs
// it on seed files made from the reference's own assets (tests/golden/susanne.obj, cornell.obj, ground.fbx, a DDS written from
// cathedral32.npz) and on files the tests' FBX writer synthesises.
//
//   fuzz_parsers <kind: obj|fbx|dds|host> <seed file or -> <cases> <rng seed> <scratch file>
//
// Every case mutates the seed (bit flips, interesting bytes / words, truncation, chunk duplication and removal, splices of digits and
// separators for the text format, header-field overwrites for the binary ones), writes it to the scratch file and calls the reader
// through the same C entry points the tests and rt_model_create_from_file use: first the counting call, then -- if that succeeded and
// the counts are sane -- the call that fills caller arrays.  A case passes when the reader returns (success or an error code) and, on
// success, every index is inside the vertex array.  Any sanitizer report aborts the process (exit code != 0); the summary line goes to
// stdout.  Import behaviour the readers keep: libs/DXRFramework/RtModel.cpp:24-82 (concatenation, position + normal only); DDS call
// site src/ProgressiveRaytracingPipeline.cpp:114-118.
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/dxr_amd.h"

static thread_local std::string g_err;
void rt_set_error(const char *fmt, ...)
{
    char b[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    g_err = b;
}
size_t &rt_alloc_limit_ref() { static size_t l = ~(size_t)0; return l; }

struct Rng {
    uint64_t s;
    uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); }
    uint32_t below(uint32_t n) { return n ? next() % n : 0; }
};

static const uint32_t WORDS[] = {0u, 1u, 2u, 0x7fu, 0x80u, 0xffu, 0x100u, 0x7fffu, 0x8000u, 0xffffu, 0x10000u, 0x7fffffffu, 0x80000000u,
                                 0xffffffffu, 0xfffffffeu, 0x40000000u, 0x3fffffffu, 0x00100000u, 16384u, 16385u, 124u, 148u};
static const char *TEXT[] = {"/", "//", "-", "-1", "0", "1e39", "-1e-46", "nan", "inf", "4294967296", "2147483648", "-2147483649", " ", "\n", "\r\n", "\\\n",
                             "f ", "v ", "vn ", "vt ", "f 1 2 3 4 5 6 7 8\n", "f -1 -2 -3\n", "v 1 2\n", "f 1/1/1 2/2/2 3//3\n", "#", "\t", "1.", ".", "e", "+"};

static void mutate(std::vector<unsigned char> &d, Rng &r, bool text)
{
    const int n_ops = 1 + (int)r.below(4);
    for (int op = 0; op < n_ops; op++) {
        const uint32_t n = (uint32_t)d.size();
        // binary containers keep most of their structure in the first bytes: aim half of the mutations there
        const uint32_t pos = n == 0 ? 0 : (!text && r.below(2) == 0 ? r.below(n < 512 ? n : 512) : r.below(n));
        switch (r.below(text ? 9 : 8)) {
        case 0: if (n) d[pos] ^= (unsigned char)(1u << r.below(8)); break;
        case 1: if (n) d[pos] = (unsigned char)WORDS[r.below(sizeof WORDS / 4)]; break;
        case 2: if (n >= 4) { const uint32_t w = WORDS[r.below(sizeof WORDS / 4)]; const uint32_t p = pos + 4 <= n ? pos : n - 4; memcpy(&d[p], &w, 4); } break;
        case 3: if (n >= 8) { const uint64_t w = (uint64_t)WORDS[r.below(sizeof WORDS / 4)] | ((uint64_t)(r.below(3) == 0 ? WORDS[r.below(sizeof WORDS / 4)] : 0u) << 32);
                              const uint32_t p = pos + 8 <= n ? pos : n - 8; memcpy(&d[p], &w, 8); } break;
        case 4: d.resize(pos); break;                                                          // truncate
        case 5: if (n) { const uint32_t len = 1 + r.below(n - pos < 64 ? n - pos : 64); d.erase(d.begin() + pos, d.begin() + pos + len); } break;
        case 6: if (n) { const uint32_t len = 1 + r.below(n - pos < 256 ? n - pos : 256); std::vector<unsigned char> c(d.begin() + pos, d.begin() + pos + len);
                         d.insert(d.begin() + r.below(n), c.begin(), c.end()); } break;        // duplicate a chunk somewhere else
        case 7: if (n >= 4) { uint32_t w; const uint32_t p = pos + 4 <= n ? pos : n - 4; memcpy(&w, &d[p], 4); w += (r.below(2) ? 1u : 0xffffffffu) * (1u + r.below(3)); memcpy(&d[p], &w, 4); } break;
        case 8: { const char *t = TEXT[r.below(sizeof TEXT / sizeof TEXT[0])]; d.insert(d.begin() + pos, t, t + strlen(t)); } break;
        }
        if (d.size() > (8u << 20)) d.resize(8u << 20);
    }
}

static bool write_file(const char *path, const std::vector<unsigned char> &d)
{
    FILE *f = fopen(path, "wb");
    if (!f) return false;
    const bool ok = d.empty() || fwrite(d.data(), 1, d.size(), f) == d.size();
    fclose(f);
    return ok;
}

typedef int (*mesh_reader)(const char *, rt_vertex *, uint32_t, uint32_t *, uint32_t, uint32_t *, uint32_t *);

static int run_mesh(mesh_reader read, const char *path, unsigned long long &ok, unsigned long long &refused)
{
    uint32_t nv = 0, nt = 0;
    const int rc = read(path, nullptr, 0, nullptr, 0, &nv, &nt);
    if (rc != RT_OK) { refused++; return 0; }
    if ((uint64_t)nv > (64u << 20) || (uint64_t)nt > (64u << 20)) { fprintf(stderr, "implausible counts %u vertices %u triangles accepted\n", nv, nt); return 1; }
    std::vector<rt_vertex> v(nv ? nv : 1);
    std::vector<uint32_t> idx((size_t)3 * (nt ? nt : 1));
    uint32_t nv2 = 0, nt2 = 0;
    const int rc2 = read(path, v.data(), nv, idx.data(), nt, &nv2, &nt2);
    if (rc2 != RT_OK || nv2 != nv || nt2 != nt) { fprintf(stderr, "second call disagrees with the first: rc %d, %u / %u vertices, %u / %u triangles\n", rc2, nv2, nv, nt2, nt); return 1; }
    for (size_t i = 0; i < (size_t)3 * nt; i++) if (idx[i] >= nv) { fprintf(stderr, "index %u outside %u vertices\n", idx[i], nv); return 1; }
    // a buffer that is one element short must be refused, not overrun
    if (nv > 1) { if (read(path, v.data(), nv - 1, idx.data(), nt, &nv2, &nt2) == RT_OK) { fprintf(stderr, "short vertex buffer accepted\n"); return 1; } }
    ok++;
    return 0;
}

static int run_dds(const char *path, unsigned long long &ok, unsigned long long &refused)
{
    uint32_t size = 0;
    const int rc = rt_dds_read_cube(path, nullptr, 0, &size);
    if (rc != RT_OK) { refused++; return 0; }
    if (size == 0 || size > 16384u) { fprintf(stderr, "cube size %u accepted\n", size); return 1; }
    const size_t n = (size_t)6 * size * size * 4;
    std::vector<float> faces(n);
    uint32_t size2 = 0;
    if (rt_dds_read_cube(path, faces.data(), n, &size2) != RT_OK || size2 != size) { fprintf(stderr, "second DDS call disagrees\n"); return 1; }
    if (rt_dds_read_cube(path, faces.data(), n - 1, &size2) == RT_OK) { fprintf(stderr, "short face buffer accepted\n"); return 1; }
    ok++;
    return 0;
}

// host logic with hostile arguments: camera frames, the per-frame update, the image writers (the partition functions live in rt_dist.hip beside
// the RCCL calls: tests/test_host_and_abi.py::test_shard_helpers covers them through the product library)
static int run_host(Rng &r, const char *scratch, unsigned long long &ok, unsigned long long &refused)
{
    static const float F[] = {0.0f, -0.0f, 1.0f, -1.0f, 1e-30f, 1e30f, 3.4e38f, -3.4e38f, 0.5f, 3.14159265f, 1e-45f, NAN, INFINITY, -INFINITY, 65504.0f, 1e-8f};
    auto f = [&]() { return r.below(3) ? F[r.below(sizeof F / 4)] : (float)((int)r.below(2001) - 1000) * 0.01f; };
    float cam[11];
    for (float &c : cam) c = f();
    float fwd[3], up[3], U[4], V[4], Wv[4];
    if (rt_camera_look(cam, cam + 3, cam + 6, fwd, up) == RT_OK) (void)rt_camera_basis(fwd, up, cam[9], cam[10], U, V, Wv);
    rt_progressive_host *h = nullptr;
    if (rt_progressive_host_create(r.next(), &h) != RT_OK) return 1;
    rt_debug_options *opt = nullptr;
    (void)rt_progressive_host_options(h, &opt);
    if (opt && r.below(2)) { opt->maxIterations = WORDS[r.below(sizeof WORDS / 4)]; opt->cosineHemisphereSampling = r.below(3); }
    (void)rt_progressive_host_set_flags(h, (int)r.below(2), (int)r.below(2));
    rt_per_frame_constants pfc;
    for (int k = 0; k < 3; k++) {
        const uint32_t w = r.below(4) ? 1 + r.below(4096) : WORDS[r.below(sizeof WORDS / 4)], hh = r.below(4) ? 1 + r.below(4096) : WORDS[r.below(sizeof WORDS / 4)];
        const int rc = r.below(2) ? rt_progressive_host_update(h, cam, f(), WORDS[r.below(sizeof WORDS / 4)], w, hh, &pfc) : rt_realtime_host_update(h, cam, f(), r.next(), w, hh, &pfc);
        if (rc == RT_OK) ok++; else refused++;
    }
    rt_progressive_host_destroy(h);
    // image writers: small images of hostile values, hostile sizes with no pixels
    const uint32_t iw = r.below(8) ? 1 + r.below(24) : 0, ih = r.below(8) ? 1 + r.below(24) : 0;
    std::vector<float> img((size_t)iw * ih * 4 + 4);
    for (float &p : img) p = f();
    const int which = (int)r.below(3);
    const int rc = which == 0 ? rt_image_write_pfm(scratch, img.data(), iw, ih) : which == 1 ? rt_image_write_exr(scratch, img.data(), iw, ih)
                                                                                             : rt_image_write_png(scratch, img.data(), iw, ih, f(), f(), (int)r.below(3));
    if (rc == RT_OK) ok++; else refused++;
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 6) { fprintf(stderr, "usage: %s obj|fbx|dds|host seed-file cases rng-seed scratch-file\n", argv[0]); return 2; }
    const std::string kind = argv[1];
    const long cases = atol(argv[3]);
    Rng r = {strtoull(argv[4], nullptr, 10) * 0x9e3779b97f4a7c15ull + 1};
    const char *scratch = argv[5];
    std::vector<unsigned char> seed;
    if (kind != "host") {
        FILE *f = fopen(argv[2], "rb");
        if (!f) { fprintf(stderr, "cannot open %s\n", argv[2]); return 2; }
        unsigned char buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, f)) > 0) seed.insert(seed.end(), buf, buf + n);
        fclose(f);
    }
    unsigned long long ok = 0, refused = 0;
    for (long c = 0; c < cases; c++) {
        if (kind == "host") { if (run_host(r, scratch, ok, refused)) return 1; continue; }
        std::vector<unsigned char> d = seed;
        if (c) mutate(d, r, kind == "obj");             // (case 0: the seed itself must be read)
        if (!write_file(scratch, d)) { fprintf(stderr, "cannot write %s\n", scratch); return 2; }
        const unsigned long long ok_before = ok;
        int bad;
        if (kind == "obj") bad = run_mesh(rt_obj_read, scratch, ok, refused);
        else if (kind == "fbx") bad = run_mesh(rt_fbx_read, scratch, ok, refused);
        else if (kind == "dds") bad = run_dds(scratch, ok, refused);
        else { fprintf(stderr, "unknown kind %s\n", kind.c_str()); return 2; }
        if (bad) { fprintf(stderr, "case %ld of seed %s failed (input left in %s)\n", c, argv[4], scratch); return 1; }
        if (c == 0 && ok == ok_before) { fprintf(stderr, "the unmutated seed %s was refused: %s\n", argv[2], g_err.c_str()); return 1; }
    }
    printf("%s: %ld cases, %llu read, %llu refused, 0 sanitizer reports\n", kind.c_str(), cases, ok, refused);
    return 0;
}
