// rt_dds.cpp -- DDS cube-map ingestion for loadResources().
//
// The reference loads assets/textures/CathedralRadiance.dds through DirectXTK12's
// CreateDDSTextureFromFile (src/ProgressiveRaytracingPipeline.cpp:114-118) and
// samples only mip 0 (RaytracingCommon.hlsli:152).  DirectXTK12 is an empty
// submodule, so this is a minimal reader of the public DDS container: DX10
// extended header or legacy FourCC, six faces in +X -X +Y -Y +Z -Z order, each
// face followed by its mip chain; RGBA16F (DXGI 10 / D3DFMT 113) or RGBA32F
// (DXGI 2 / D3DFMT 116).  Texels are widened to fp32 (exact).
#include "rt_internal.h"

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size);

namespace {

inline float half_to_float(uint16_t h)
{
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {                                  // subnormal half -> normal float
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

inline uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

}  // namespace

int rt_dds_load_cube(const char *path, std::vector<float> &faces, uint32_t &size)
{
    FILE *f = fopen(path, "rb");
    if (!f) { rt_set_error("cannot open DDS file '%s'", path); return RT_ERR_IO; }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> d((size_t)(len > 0 ? len : 0));
    const size_t got = d.empty() ? 0 : fread(d.data(), 1, d.size(), f);
    fclose(f);
    if (got != d.size() || d.size() < 128 || memcmp(d.data(), "DDS ", 4) != 0 || rd32(&d[4]) != 124) {
        rt_set_error("'%s' is not a DDS file", path);
        return RT_ERR_IO;
    }
    const uint32_t height = rd32(&d[12]), width = rd32(&d[16]);
    uint32_t mips = rd32(&d[28]);
    if (mips == 0) mips = 1;
    // header fields are untrusted: bound them before any size arithmetic (a crafted width such as 2^30 + k would wrap
    // the byte counts below and pass the truncation check with a tiny payload)
    if (width > 16384u || height > 16384u || mips > 15u) {
        rt_set_error("'%s': implausible DDS header (%ux%u, %u mips)", path, width, height, mips);
        return RT_ERR_UNSUPPORTED;
    }
    const uint32_t pf_flags = rd32(&d[80]), fourcc = rd32(&d[84]);
    const uint32_t caps2 = rd32(&d[112]);
    size_t off = 128;
    uint32_t bpp = 0;       // bytes per texel
    bool cube = (caps2 & 0x200u) != 0;
    if ((pf_flags & 0x4u) && fourcc == 0x30315844u) {           // "DX10"
        if (d.size() < 148) { rt_set_error("'%s': truncated DX10 header", path); return RT_ERR_IO; }
        const uint32_t fmt = rd32(&d[128]), misc = rd32(&d[136]);
        off = 148;
        cube = cube || (misc & 0x4u);
        if (fmt == 10) bpp = 8; else if (fmt == 2) bpp = 16;
    } else if (pf_flags & 0x4u) {
        if (fourcc == 113) bpp = 8; else if (fourcc == 116) bpp = 16;
    }
    if (!bpp) { rt_set_error("'%s': only RGBA16F / RGBA32F cube maps are supported", path); return RT_ERR_UNSUPPORTED; }
    if (!cube || width != height || width == 0) { rt_set_error("'%s' is not a square cube map", path); return RT_ERR_UNSUPPORTED; }
    size_t face_bytes = 0;
    for (uint32_t m = 0, w = width; m < mips; m++, w = w > 1 ? w / 2 : 1) face_bytes += (size_t)w * w * bpp;
    if (d.size() < off || 6 * face_bytes > d.size() - off) { rt_set_error("'%s': truncated texel data", path); return RT_ERR_IO; }
    size = width;
    faces.resize((size_t)6 * width * width * 4);
    for (int face = 0; face < 6; face++) {
        const unsigned char *src = &d[off + (size_t)face * face_bytes];
        float *dst = &faces[(size_t)face * width * width * 4];
        const size_t n = (size_t)width * width * 4;
        if (bpp == 16) memcpy(dst, src, n * 4);
        else for (size_t i = 0; i < n; i++) dst[i] = half_to_float((uint16_t)(src[2 * i] | (src[2 * i + 1] << 8)));
    }
    return RT_OK;
}

extern "C" int rt_dds_read_cube(const char *path, float *faces_rgba32f, size_t capacity_floats, uint32_t *size)
{
    RT_REQUIRE(path && size, "null argument");
    std::vector<float> faces;
    uint32_t n = 0;
    RT_TRY(rt_dds_load_cube(path, faces, n));
    *size = n;
    if (!faces_rgba32f) return RT_OK;
    RT_REQUIRE(capacity_floats >= faces.size(), "buffer too small for 6 x size x size x 4 floats");
    memcpy(faces_rgba32f, faces.data(), faces.size() * sizeof(float));
    return RT_OK;
}
