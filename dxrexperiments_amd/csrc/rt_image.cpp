// rt_image.cpp -- image files for the outputs ("PNG/EXR/PFM writer").
//
// The reference only ever shows its outputs in a window (CopyResource to the back buffer,
// src/DXRExperimentsApp.cpp:213-214); a headless engine needs files.  Two formats, both
// written without external libraries (round 3: three -- OpenEXR, the format every renderer-side tool reads):
//   EXR  lossless fp32 RGBA: single-part scan-line OpenEXR 2.0, no compression, channels A B G R (the file format's
//        alphabetical order), one scan line per chunk, rows top to bottom;
//   PFM  lossless fp32 RGB, the accumulation image as it is (rows bottom to top, little endian);
//   PNG  8-bit RGB for viewing: the DenoiseCompositor's display transform (exposure, optional
//        Reinhard, gamma; DenoiseCommon.hlsli:29-41 in spirit -- host libm, not a parity path),
//        zlib stream of stored blocks (no compression), CRC-32 / Adler-32 computed here.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "rt_internal.h"

namespace {

struct CrcTable {          // built once by the static initialiser (contexts on different threads may write images)
    uint32_t t[256];
    CrcTable()
    {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[i] = c;
        }
    }
};
const CrcTable crc_table;
uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = crc_table.t[(crc ^ p[i]) & 0xFFu] ^ (crc >> 8);
    return ~crc;
}

void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

bool write_chunk(FILE *f, const char type[4], const std::vector<uint8_t> &data)
{
    std::vector<uint8_t> head;
    put32(head, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    std::vector<uint8_t> tail;
    put32(tail, crc32(0, body.data(), body.size()));
    return fwrite(head.data(), 1, 4, f) == 4 && fwrite(body.data(), 1, body.size(), f) == body.size() && fwrite(tail.data(), 1, 4, f) == 4;
}

}  // namespace

extern "C" {

int rt_image_write_pfm(const char *path, const float *rgba32f, uint32_t width, uint32_t height)
{
    RT_REQUIRE(path && rgba32f, "null argument");
    RT_REQUIRE(width > 0 && height > 0, "empty image");
    FILE *f = fopen(path, "wb");
    if (!f) { rt_set_error("cannot create %s", path); return RT_ERR_IO; }
    bool ok = fprintf(f, "PF\n%u %u\n-1.0\n", width, height) > 0;
    std::vector<float> row((size_t)width * 3);
    for (uint32_t y = height; ok && y-- > 0;) {                       // PFM rows run bottom to top
        const float *src = rgba32f + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++) { row[3 * x] = src[4 * x]; row[3 * x + 1] = src[4 * x + 1]; row[3 * x + 2] = src[4 * x + 2]; }
        ok = fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { rt_set_error("short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

int rt_image_write_exr(const char *path, const float *rgba32f, uint32_t width, uint32_t height)
{
    RT_REQUIRE(path && rgba32f, "null argument");
    RT_REQUIRE(width > 0 && height > 0 && width < (1u << 30) && height < (1u << 30), "bad image size");
    std::vector<uint8_t> h;
    auto bytes = [&](const void *p, size_t n) { h.insert(h.end(), (const uint8_t *)p, (const uint8_t *)p + n); };
    auto str = [&](const char *z) { bytes(z, strlen(z) + 1); };
    auto i32 = [&](int32_t v) { bytes(&v, 4); };            // (little endian host, like everything else here)
    auto f32 = [&](float v) { bytes(&v, 4); };
    auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};        // 20000630, version 2, no flags: single-part scan-line file
    bytes(magic, 8);
    attr("channels", "chlist", 4 * 18 + 1);
    for (const char *c : {"A", "B", "G", "R"}) { str(c); i32(2 /* FLOAT */); const uint8_t lin[4] = {0, 0, 0, 0}; bytes(lin, 4); i32(1); i32(1); }
    h.push_back(0);
    attr("compression", "compression", 1); h.push_back(0);                 // NO_COMPRESSION
    attr("dataWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32((int32_t)width - 1); i32((int32_t)height - 1);
    attr("lineOrder", "lineOrder", 1); h.push_back(0);                     // INCREASING_Y
    attr("pixelAspectRatio", "float", 4); f32(1.0f);
    attr("screenWindowCenter", "v2f", 8); f32(0.0f); f32(0.0f);
    attr("screenWindowWidth", "float", 4); f32(1.0f);
    h.push_back(0);                                                         // end of header
    const uint64_t line_bytes = (uint64_t)width * 16, chunk = 8 + line_bytes;
    const uint64_t first = h.size() + (uint64_t)height * 8;
    for (uint32_t y = 0; y < height; y++) { const uint64_t off = first + y * chunk; bytes(&off, 8); }
    FILE *f = fopen(path, "wb");
    if (!f) { rt_set_error("cannot create %s", path); return RT_ERR_IO; }
    bool ok = fwrite(h.data(), 1, h.size(), f) == h.size();
    std::vector<float> row((size_t)width * 4);
    static const int channel_of[4] = {3, 2, 1, 0};                          // file order A B G R <- memory order R G B A
    for (uint32_t y = 0; ok && y < height; y++) {
        const float *src = rgba32f + (size_t)y * width * 4;
        for (int c = 0; c < 4; c++)
            for (uint32_t x = 0; x < width; x++) row[(size_t)c * width + x] = src[4 * (size_t)x + channel_of[c]];
        const int32_t head[2] = {(int32_t)y, (int32_t)line_bytes};
        ok = fwrite(head, 4, 2, f) == 2 && fwrite(row.data(), sizeof(float), row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    if (!ok) { rt_set_error("short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

int rt_image_write_png(const char *path, const float *rgba32f, uint32_t width, uint32_t height, float exposure, float gamma, int tonemap)
{
    RT_REQUIRE(path && rgba32f, "null argument");
    RT_REQUIRE(width > 0 && height > 0, "empty image");
    RT_REQUIRE(gamma > 0.0f, "gamma must be positive");
    // raw scanlines: filter byte 0 + RGB8
    const size_t stride = (size_t)width * 3 + 1;
    std::vector<uint8_t> raw(stride * height);
    for (uint32_t y = 0; y < height; y++) {
        uint8_t *dst = raw.data() + stride * y;
        *dst++ = 0;
        const float *src = rgba32f + (size_t)y * width * 4;
        for (uint32_t x = 0; x < width; x++)
            for (int c = 0; c < 3; c++) {
                float v = src[4 * x + c] * exposure;
                if (!(v > 0.0f)) v = 0.0f;                              // negatives and NaN -> black
                if (tonemap) v = v > 3.0e38f ? 1.0f : v / (1.0f + v);   // Reinhard; +inf (or an overflowed sum) -> white, not inf/inf
                v = powf(v, 1.0f / gamma);
                if (!(v <= 1.0f)) v = 1.0f;                             // also catches a NaN from powf
                if (!(v >= 0.0f)) v = 0.0f;
                *dst++ = (uint8_t)(v * 255.0f + 0.5f);
            }
    }
    // zlib container around stored (uncompressed) deflate blocks of <= 65535 bytes
    std::vector<uint8_t> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;
    for (size_t off = 0; off < raw.size();) {
        const size_t n = raw.size() - off < 65535 ? raw.size() - off : 65535;
        z.push_back(off + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF)); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF)); z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        for (size_t i = 0; i < n; i++) { a = (a + raw[off + i]) % 65521u; b = (b + a) % 65521u; }
        off += n;
    }
    put32(z, (b << 16) | a);
    FILE *f = fopen(path, "wb");
    if (!f) { rt_set_error("cannot create %s", path); return RT_ERR_IO; }
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put32(ihdr, width); put32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);     // 8-bit, truecolour
    bool ok = fwrite(sig, 1, 8, f) == 8 && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", std::vector<uint8_t>());
    ok = (fclose(f) == 0) && ok;
    if (!ok) { rt_set_error("short write to %s", path); return RT_ERR_IO; }
    return RT_OK;
}

}  // extern "C"
