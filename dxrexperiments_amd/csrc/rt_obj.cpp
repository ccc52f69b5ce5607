// rt_obj.cpp -- Wavefront OBJ ingestion for RtModel::create(ctx, path).
//
// Replaces the reference's Assimp import (libs/DXRFramework/RtModel.cpp:26-27,
// flags Triangulate | GenSmoothNormals | JoinIdenticalVertices |
// PreTransformVertices) and produces what its constructor builds (:33-81): one
// interleaved {position, normal} vertex array and one u32 triangle list.
// Assimp is not available, so ordering is defined here:
//   - primitive id = order of faces in the file; an n-gon (v0..vn-1) becomes the
//     fan (v0,v1,v2), (v0,v2,v3), ...
//   - vertices are joined per distinct (position index, normal index) pair and
//     numbered in first-use order;
//   - a corner without a normal index gets a generated smooth normal: the
//     normalised sum of cross(b-a, c-a) over every triangle using that position.
//   - texture coordinates, materials, groups and objects are ignored (the
//     reference drops them too: it keeps position + normal only, RtModel.cpp:38-44).
#include <stdlib.h>
#include <math.h>

#include <map>
#include <utility>

#include "rt_internal.h"

namespace {

struct P3 { float x, y, z; };
struct Corner { int p, n; };
constexpr int RT_BAD_INDEX = 0x7fffffff;

// one whole line of the file whatever its length (n-gon records and exporter-wrapped lines exceed any fixed buffer)
bool read_line(FILE *f, std::vector<char> &line)
{
    line.clear();
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (c == '\n') break;
        line.push_back((char)c);
    }
    if (c == EOF && line.empty()) return false;
    line.push_back(0);
    return true;
}

inline bool is_blank(char c) { return c == ' ' || c == '\t'; }

bool parse_corner(char *&s, size_t npos, size_t nnrm, Corner &out)
{
    while (is_blank(*s)) s++;
    if (*s == 0 || *s == '\n' || *s == '\r' || *s == '#') return false;
    char *e;
    long pi = strtol(s, &e, 10);
    if (e == s) return false;
    s = e;
    long ni = 0;
    bool has_n = false;
    if (*s == '/') {
        s++;
        if (*s != '/') { (void)strtol(s, &e, 10); s = e; }
        if (*s == '/') {
            s++;
            ni = strtol(s, &e, 10);
            has_n = e != s;
            s = e;
        }
    }
    // 1-based, or negative = relative to the end of the list read so far.  An index that resolves below 0 must not
    // collide with -1 ("no normal"): it becomes RT_BAD_INDEX and fails the range check after parsing.
    const long p = pi < 0 ? (long)npos + pi : pi - 1;
    const long n = has_n ? (ni < 0 ? (long)nnrm + ni : ni - 1) : -1;
    out.p = p < 0 || p > 0x7ffffff0L ? RT_BAD_INDEX : (int)p;
    out.n = !has_n ? -1 : (n < 0 || n > 0x7ffffff0L ? RT_BAD_INDEX : (int)n);
    return true;
}

}  // namespace

int rt_obj_parse(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx)
{
    FILE *f = fopen(path, "rb");
    if (!f) {
        rt_set_error("cannot open OBJ file '%s'", path);
        return RT_ERR_IO;
    }
    std::vector<P3> pos, nrm;
    std::vector<Corner> corners;
    std::vector<Corner> poly;
    std::vector<char> line;
    while (read_line(f, line)) {
        char *s = line.data();
        while (is_blank(*s)) s++;
        if (s[0] == 'v' && is_blank(s[1])) {
            char *e = s + 1;
            P3 p;
            p.x = strtof(e, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
            pos.push_back(p);
        } else if (s[0] == 'v' && s[1] == 'n' && is_blank(s[2])) {
            char *e = s + 2;
            P3 p;
            p.x = strtof(e, &e); p.y = strtof(e, &e); p.z = strtof(e, &e);
            nrm.push_back(p);
        } else if (s[0] == 'f' && is_blank(s[1])) {
            poly.clear();
            char *e = s + 1;
            Corner c;
            while (parse_corner(e, pos.size(), nrm.size(), c)) poly.push_back(c);
            for (size_t k = 1; k + 1 < poly.size(); k++) {
                corners.push_back(poly[0]);
                corners.push_back(poly[k]);
                corners.push_back(poly[k + 1]);
            }
        }
    }
    fclose(f);
    bool need_gen = false;
    for (const Corner &c : corners) {
        if (c.p < 0 || c.p >= (int)pos.size() || c.n >= (int)nrm.size() || c.n < -1) {
            rt_set_error("OBJ file '%s': face index out of range", path);
            return RT_ERR_IO;
        }
        if (c.n < 0) need_gen = true;
    }
    if (corners.empty()) {
        rt_set_error("OBJ file '%s' holds no faces", path);
        return RT_ERR_IO;
    }
    std::vector<P3> gen;
    if (need_gen) {
        P3 zero = {0.0f, 0.0f, 0.0f};
        gen.assign(pos.size(), zero);
        for (size_t t = 0; t + 2 < corners.size(); t += 3) {
            const P3 a = pos[corners[t].p], b = pos[corners[t + 1].p], c = pos[corners[t + 2].p];
            const float e1x = b.x - a.x, e1y = b.y - a.y, e1z = b.z - a.z;
            const float e2x = c.x - a.x, e2y = c.y - a.y, e2z = c.z - a.z;
            const float nx = e1y * e2z - e1z * e2y, ny = e1z * e2x - e1x * e2z, nz = e1x * e2y - e1y * e2x;
            for (int k = 0; k < 3; k++) {
                P3 &g = gen[corners[t + k].p];
                g.x = g.x + nx; g.y = g.y + ny; g.z = g.z + nz;
            }
        }
        for (P3 &g : gen) {
            float d = g.x * g.x;
            d = d + g.y * g.y;
            d = d + g.z * g.z;
            const float l = sqrtf(d);
            if (l > 0.0f) { g.x = g.x / l; g.y = g.y / l; g.z = g.z / l; }
        }
    }
    std::map<std::pair<int, int>, uint32_t> joined;
    verts.clear();
    idx.clear();
    idx.reserve(corners.size());
    for (const Corner &c : corners) {
        const std::pair<int, int> key(c.p, c.n);
        std::map<std::pair<int, int>, uint32_t>::const_iterator it = joined.find(key);
        if (it != joined.end()) {
            idx.push_back(it->second);
            continue;
        }
        const P3 n = c.n >= 0 ? nrm[c.n] : gen[c.p];
        rt_vertex v;
        v.position.x = pos[c.p].x; v.position.y = pos[c.p].y; v.position.z = pos[c.p].z;
        v.normal.x = n.x; v.normal.y = n.y; v.normal.z = n.z;
        const uint32_t id = (uint32_t)verts.size();
        verts.push_back(v);
        joined[key] = id;
        idx.push_back(id);
    }
    return RT_OK;
}

extern "C" int rt_obj_read(const char *path, rt_vertex *verts, uint32_t capacity_verts, uint32_t *indices, uint32_t capacity_tris,
                           uint32_t *n_verts, uint32_t *n_tris)
{
    RT_REQUIRE(path && n_verts && n_tris, "null argument");
    std::vector<rt_vertex> v;
    std::vector<uint32_t> idx;
    RT_TRY(rt_obj_parse(path, v, idx));
    *n_verts = (uint32_t)v.size();
    *n_tris = (uint32_t)(idx.size() / 3);
    if (!verts && !indices) return RT_OK;
    RT_REQUIRE(verts && indices && capacity_verts >= v.size() && capacity_tris >= idx.size() / 3, "buffers too small");
    memcpy(verts, v.data(), v.size() * sizeof(rt_vertex));
    memcpy(indices, idx.data(), idx.size() * sizeof(uint32_t));
    return RT_OK;
}
