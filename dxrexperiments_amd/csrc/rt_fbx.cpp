// rt_fbx.cpp -- minimal binary-FBX mesh ingestion for RtModel::create(ctx, path).
//
// The reference imports every model through Assimp (libs/DXRFramework/RtModel.cpp:24-82, flags Triangulate |
// GenSmoothNormals | FlipUVs | JoinIdenticalVertices | PreTransformVertices) and keeps position + normal only; its app
// loads an FBX scene, and the one FBX in its checkout is assets/models/ground.fbx (binary FBX 7.5, one Geometry: a
// 441-vertex plane of 400 quads with per-polygon-vertex normals).  Assimp is not available, so -- as for OBJ files
// (rt_obj.cpp) -- the ordering is DEFINED here:
//   - the file's Geometry nodes of class "Mesh" are taken in file order and concatenated into one vertex / index array
//     (the reference concatenates Assimp's meshes the same way, :36-56);
//   - primitive id = order of polygons in PolygonVertexIndex; an n-gon (v0..vn-1) becomes the fan (v0,v1,v2), (v0,v2,v3) ...;
//   - normals come from LayerElementNormal (ByPolygonVertex or ByVertice / ByVertex, Direct or IndexToDirect); a geometry
//     without them gets generated smooth normals: the normalised sum of cross(b-a, c-a) over the triangles at a position;
//   - vertices are joined per distinct (position index, normal value) pair and numbered in first-use order;
//   - "PreTransformVertices": the Lcl Translation / Lcl Rotation (Euler XYZ, degrees) / Lcl Scaling of the Model a geometry is
//     connected to are applied (normals by the inverse transpose, renormalised).  Parent chains, pivots and pre / post
//     rotations are not: a file that needs them is outside this reader's scope and says so in the error text.
// Doubles are converted to float with a plain cast.  Arrays may be zlib-compressed (encoding 1).
#include <math.h>
#include <stdlib.h>
#include <zlib.h>

#include <map>
#include <new>
#include <string>

#include "rt_internal.h"

namespace {

struct Reader {
    const unsigned char *d;
    size_t n;
    bool wide;          // FBX >= 7500: 64-bit record fields
};

struct Prop {
    char type = 0;
    std::vector<double> nums;       // scalars and arrays, as doubles ('L' ids exactly up to 2^53: enough to tell them apart)
    std::vector<long long> ints;    // integer scalars / arrays, exactly
    std::string str;
};

struct Node {
    std::string name;
    std::vector<Prop> props;
    std::vector<Node> kids;
    const Node *kid(const char *nm) const
    {
        for (const Node &k : kids) if (k.name == nm) return &k;
        return nullptr;
    }
};

template <class T> bool rd(const Reader &r, size_t off, T &out)
{
    if (off + sizeof(T) > r.n) return false;
    memcpy(&out, r.d + off, sizeof(T));
    return true;
}

bool read_array(const Reader &r, size_t &p, char type, Prop &out)
{
    uint32_t count = 0, enc = 0, clen = 0;
    if (!rd(r, p, count) || !rd(r, p + 4, enc) || !rd(r, p + 8, clen)) return false;
    p += 12;
    if (p + clen > r.n) return false;
    const size_t es = (type == 'd' || type == 'l') ? 8 : (type == 'b' ? 1 : 4);
    // (a count the payload cannot possibly hold -- deflate expands at most ~1032 : 1 -- is a malformed file, not an allocation)
    if ((size_t)count * es > (enc == 1 ? (size_t)clen * 1100 + 64 : (size_t)clen)) return false;
    std::vector<unsigned char> raw((size_t)count * es);
    if (enc == 0) {
        if (clen != raw.size()) return false;
        if (!raw.empty()) memcpy(raw.data(), r.d + p, raw.size());      // (an empty array has no storage: memcpy(nullptr, ., 0) is undefined by the letter -- found by tests/test_sanitized_parsers.py)
    } else if (enc == 1) {
        uLongf dl = (uLongf)raw.size();
        unsigned char none = 0;
        if (uncompress(raw.empty() ? &none : raw.data(), &dl, r.d + p, clen) != Z_OK || dl != raw.size()) return false;
    } else return false;
    p += clen;
    out.nums.resize(count);
    if (type == 'i' || type == 'l' || type == 'b') out.ints.resize(count);
    for (uint32_t i = 0; i < count; i++) {
        if (type == 'd') { double v; memcpy(&v, &raw[(size_t)i * 8], 8); out.nums[i] = v; }
        else if (type == 'f') { float v; memcpy(&v, &raw[(size_t)i * 4], 4); out.nums[i] = v; }
        else if (type == 'i') { int32_t v; memcpy(&v, &raw[(size_t)i * 4], 4); out.ints[i] = v; out.nums[i] = v; }
        else if (type == 'l') { long long v; memcpy(&v, &raw[(size_t)i * 8], 8); out.ints[i] = v; out.nums[i] = (double)v; }
        else { out.ints[i] = raw[i]; out.nums[i] = raw[i]; }
    }
    return true;
}

// one node record at `off`; returns false on a malformed file; *end = 0 for the 13- / 25-byte null record that closes a list
bool read_node(const Reader &r, size_t off, Node &out, size_t *end, int depth)
{
    if (depth > 64) return false;
    unsigned long long e = 0, np = 0, pl = 0;
    if (r.wide) {
        if (!rd(r, off, e) || !rd(r, off + 8, np) || !rd(r, off + 16, pl)) return false;
        off += 24;
    } else {
        uint32_t a = 0, b = 0, c = 0;
        if (!rd(r, off, a) || !rd(r, off + 4, b) || !rd(r, off + 8, c)) return false;
        e = a; np = b; pl = c;
        off += 12;
    }
    unsigned char nl = 0;
    if (!rd(r, off, nl)) return false;
    off += 1;
    if (e == 0) { *end = 0; return true; }
    if (off + nl > r.n || e > r.n || e < off + nl + pl) return false;
    out.name.assign((const char *)r.d + off, nl);
    off += nl;
    size_t p = off;
    for (unsigned long long i = 0; i < np; i++) {
        Prop pr;
        char t = 0;
        if (!rd(r, p, t)) return false;
        p += 1;
        pr.type = t;
        switch (t) {
        case 'Y': { int16_t v; if (!rd(r, p, v)) return false; p += 2; pr.ints.push_back(v); pr.nums.push_back(v); break; }
        case 'C': { unsigned char v; if (!rd(r, p, v)) return false; p += 1; pr.ints.push_back(v); pr.nums.push_back(v); break; }
        case 'I': { int32_t v; if (!rd(r, p, v)) return false; p += 4; pr.ints.push_back(v); pr.nums.push_back(v); break; }
        case 'F': { float v; if (!rd(r, p, v)) return false; p += 4; pr.nums.push_back(v); break; }
        case 'D': { double v; if (!rd(r, p, v)) return false; p += 8; pr.nums.push_back(v); break; }
        case 'L': { long long v; if (!rd(r, p, v)) return false; p += 8; pr.ints.push_back(v); pr.nums.push_back((double)v); break; }
        case 'f': case 'd': case 'l': case 'i': case 'b':
            if (!read_array(r, p, t, pr)) return false;
            break;
        case 'S': case 'R': {
            uint32_t len = 0;
            if (!rd(r, p, len) || p + 4 + len > r.n) return false;
            pr.str.assign((const char *)r.d + p + 4, len);
            p += 4 + (size_t)len;
            break;
        }
        default: return false;
        }
        out.props.push_back(std::move(pr));
    }
    if (p != off + pl) return false;
    size_t at = off + pl;
    while (at < e) {
        Node k;
        size_t ke = 0;
        if (!read_node(r, at, k, &ke, depth + 1)) return false;
        if (ke == 0) break;                              // the null record: end of this node's children
        if (ke <= at) return false;                      // (a record that does not advance: malformed)
        out.kids.push_back(std::move(k));
        at = ke;
    }
    *end = (size_t)e;
    return true;
}

struct M34 { double m[12]; };       // 3x4 row-major affine

M34 identity34() { M34 r = {{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0}}; return r; }

// T * Rz * Ry * Rx * S (FBX's default eEulerXYZ order: X applied first), angles in degrees
M34 model_transform(const Node &model, bool *unsupported)
{
    double t[3] = {0, 0, 0}, rdeg[3] = {0, 0, 0}, s[3] = {1, 1, 1};
    if (const Node *p70 = model.kid("Properties70")) {
        for (const Node &p : p70->kids) {
            if (p.name != "P" || p.props.empty()) continue;
            const std::string &key = p.props[0].str;
            double *dst = key == "Lcl Translation" ? t : key == "Lcl Rotation" ? rdeg : key == "Lcl Scaling" ? s : nullptr;
            if (dst) {
                int k = 0;
                for (size_t i = 4; i < p.props.size() && k < 3; i++)
                    if (!p.props[i].nums.empty()) dst[k++] = p.props[i].nums[0];
            } else if (key == "PreRotation" || key == "PostRotation" || key == "RotationPivot" || key == "ScalingPivot" || key == "RotationOffset" ||
                       key == "ScalingOffset" || key == "GeometricTranslation" || key == "GeometricRotation" || key == "GeometricScaling") {
                for (size_t i = 4; i < p.props.size(); i++)
                    if (!p.props[i].nums.empty() && p.props[i].nums[0] != 0.0 && !(key == "GeometricScaling" && p.props[i].nums[0] == 1.0)) *unsupported = true;
            }
        }
    }
    const double k = 3.14159265358979323846 / 180.0;
    const double cx = cos(rdeg[0] * k), sx = sin(rdeg[0] * k), cy = cos(rdeg[1] * k), sy = sin(rdeg[1] * k), cz = cos(rdeg[2] * k), sz = sin(rdeg[2] * k);
    // R = Rz * Ry * Rx
    const double R[9] = {cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                         sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                         -sy, cy * sx, cy * cx};
    M34 o;
    for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) o.m[4 * r + c] = R[3 * r + c] * s[c];
        o.m[4 * r + 3] = t[r];
    }
    return o;
}

bool is_identity(const M34 &a)
{
    const M34 i = identity34();
    for (int k = 0; k < 12; k++) if (a.m[k] != i.m[k]) return false;
    return true;
}

struct P3 { float x, y, z; };

struct Key {
    int p;
    uint32_t nx, ny, nz;
    bool operator<(const Key &o) const
    {
        if (p != o.p) return p < o.p;
        if (nx != o.nx) return nx < o.nx;
        if (ny != o.ny) return ny < o.ny;
        return nz < o.nz;
    }
};

uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int add_geometry(const Node &geo, const M34 &xf, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx)
{
    const Node *vn = geo.kid("Vertices"), *pn = geo.kid("PolygonVertexIndex");
    if (!vn || !pn || vn->props.empty() || pn->props.empty()) return RT_OK;            // not a mesh with polygons: skipped
    const std::vector<double> &vd = vn->props[0].nums;
    const std::vector<long long> &pv = pn->props[0].ints;
    if (vd.size() % 3 != 0) { rt_set_error("FBX: Vertices array of %zu doubles", vd.size()); return RT_ERR_IO; }
    const size_t npos = vd.size() / 3;
    const bool plain = is_identity(xf);
    std::vector<P3> pos(npos);
    for (size_t i = 0; i < npos; i++) {
        double x = vd[3 * i], y = vd[3 * i + 1], z = vd[3 * i + 2];
        if (!plain) {
            const double a = xf.m[0] * x + xf.m[1] * y + xf.m[2] * z + xf.m[3], b = xf.m[4] * x + xf.m[5] * y + xf.m[6] * z + xf.m[7],
                         c = xf.m[8] * x + xf.m[9] * y + xf.m[10] * z + xf.m[11];
            x = a; y = b; z = c;
        }
        pos[i] = P3{(float)x, (float)y, (float)z};
    }
    // normals
    const Node *ln = geo.kid("LayerElementNormal");
    std::vector<double> nd;
    std::vector<long long> nidx;
    bool by_polygon_vertex = true, indexed = false, have_normals = false;
    if (ln) {
        const Node *arr = ln->kid("Normals"), *map = ln->kid("MappingInformationType"), *ref = ln->kid("ReferenceInformationType"), *ni = ln->kid("NormalsIndex");
        if (arr && !arr->props.empty() && map && !map->props.empty()) {
            const std::string &m = map->props[0].str;
            if (m == "ByPolygonVertex") by_polygon_vertex = true;
            else if (m == "ByVertice" || m == "ByVertex") by_polygon_vertex = false;
            else { rt_set_error("FBX: normals mapped %s (only ByPolygonVertex / ByVertice)", m.c_str()); return RT_ERR_UNSUPPORTED; }
            nd = arr->props[0].nums;
            indexed = ref && !ref->props.empty() && ref->props[0].str == "IndexToDirect" && ni && !ni->props.empty();
            if (indexed) nidx = ni->props[0].ints;
            have_normals = nd.size() >= 3 && nd.size() % 3 == 0;
        }
    }
    // inverse transpose of the linear part, for normals
    double it[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (!plain) {
        const double a = xf.m[0], b = xf.m[1], c = xf.m[2], d = xf.m[4], e = xf.m[5], f = xf.m[6], g = xf.m[8], h = xf.m[9], i = xf.m[10];
        const double det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
        if (det == 0.0 || det != det) { rt_set_error("FBX: singular model transform"); return RT_ERR_UNSUPPORTED; }
        const double inv[9] = {(e * i - f * h) / det, (c * h - b * i) / det, (b * f - c * e) / det, (f * g - d * i) / det, (a * i - c * g) / det,
                               (c * d - a * f) / det, (d * h - e * g) / det, (b * g - a * h) / det, (a * e - b * d) / det};
        for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) it[3 * r + cc] = inv[3 * cc + r];
    }
    // polygons -> corners (position index, polygon-vertex ordinal) in fan order
    struct Corner { int p; size_t pvi; };
    std::vector<Corner> corners;
    std::vector<Corner> poly;
    for (size_t k = 0; k < pv.size(); k++) {
        long long v = pv[k];
        const bool last = v < 0;
        if (last) v = ~v;
        if (v < 0 || (size_t)v >= npos) { rt_set_error("FBX: polygon vertex %lld out of range (%zu positions)", v, npos); return RT_ERR_IO; }
        poly.push_back(Corner{(int)v, k});
        if (last) {
            for (size_t j = 1; j + 1 < poly.size(); j++) { corners.push_back(poly[0]); corners.push_back(poly[j]); corners.push_back(poly[j + 1]); }
            poly.clear();
        }
    }
    // generated smooth normals where the file has none
    std::vector<P3> gen;
    if (!have_normals) {
        gen.assign(npos, P3{0, 0, 0});
        for (size_t t = 0; t + 2 < corners.size(); t += 3) {
            const P3 a = pos[corners[t].p], b = pos[corners[t + 1].p], c = pos[corners[t + 2].p];
            const float ux = b.x - a.x, uy = b.y - a.y, uz = b.z - a.z, vx = c.x - a.x, vy = c.y - a.y, vz = c.z - a.z;
            const float nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
            for (int k = 0; k < 3; k++) { P3 &g = gen[corners[t + k].p]; g.x += nx; g.y += ny; g.z += nz; }
        }
        for (P3 &g : gen) {
            const float l = sqrtf(g.x * g.x + g.y * g.y + g.z * g.z);
            if (l > 0.0f) { g.x /= l; g.y /= l; g.z /= l; }
        }
    }
    const uint32_t base = (uint32_t)verts.size();
    std::map<Key, uint32_t> joined;
    for (const Corner &c : corners) {
        P3 n;
        if (have_normals) {
            size_t e = by_polygon_vertex ? c.pvi : (size_t)c.p;
            if (indexed) {
                if (e >= nidx.size() || nidx[e] < 0) { rt_set_error("FBX: normal index out of range"); return RT_ERR_IO; }
                e = (size_t)nidx[e];
            }
            if (3 * e + 2 >= nd.size()) { rt_set_error("FBX: normal %zu out of range (%zu normals)", e, nd.size() / 3); return RT_ERR_IO; }
            double x = nd[3 * e], y = nd[3 * e + 1], z = nd[3 * e + 2];
            if (!plain) {
                const double a = it[0] * x + it[1] * y + it[2] * z, b = it[3] * x + it[4] * y + it[5] * z, cc = it[6] * x + it[7] * y + it[8] * z;
                const double l = sqrt(a * a + b * b + cc * cc);
                x = l > 0 ? a / l : a; y = l > 0 ? b / l : b; z = l > 0 ? cc / l : cc;
            }
            n = P3{(float)x, (float)y, (float)z};
        } else n = gen[c.p];
        const Key key = {c.p, bits(n.x), bits(n.y), bits(n.z)};
        std::map<Key, uint32_t>::const_iterator f = joined.find(key);
        if (f != joined.end()) { idx.push_back(f->second); continue; }
        rt_vertex v;
        v.position.x = pos[c.p].x; v.position.y = pos[c.p].y; v.position.z = pos[c.p].z;
        v.normal.x = n.x; v.normal.y = n.y; v.normal.z = n.z;
        const uint32_t id = base + (uint32_t)(verts.size() - base);
        verts.push_back(v);
        joined[key] = id;
        idx.push_back(id);
    }
    return RT_OK;
}

}  // namespace

static int fbx_parse_unguarded(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx);

int rt_fbx_parse(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx)
{
    try {                                    // (no exception crosses the C ABI)
        return fbx_parse_unguarded(path, verts, idx);
    } catch (const std::bad_alloc &) {
        rt_set_error("%s: out of host memory", path);
        return RT_ERR_OOM;
    }
}

static int fbx_parse_unguarded(const char *path, std::vector<rt_vertex> &verts, std::vector<uint32_t> &idx)
{
    FILE *f = fopen(path, "rb");
    if (!f) { rt_set_error("cannot open %s", path); return RT_ERR_IO; }
    std::vector<unsigned char> data;
    unsigned char buf[65536];
    size_t got;
    while ((got = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + got);
    fclose(f);
    static const char magic[] = "Kaydara FBX Binary  ";
    if (data.size() < 27 || memcmp(data.data(), magic, 20) != 0) {
        rt_set_error("%s: not a binary FBX file (ASCII FBX is not supported)", path);
        return RT_ERR_UNSUPPORTED;
    }
    uint32_t version = 0;
    memcpy(&version, &data[23], 4);
    Reader r = {data.data(), data.size(), version >= 7500};
    std::vector<Node> top;
    size_t at = 27;
    while (at + (r.wide ? 25 : 13) <= data.size()) {
        Node n;
        size_t e = 0;
        if (!read_node(r, at, n, &e, 0) || (e != 0 && e <= at)) { rt_set_error("%s: malformed FBX record at byte %zu", path, at); return RT_ERR_IO; }
        if (e == 0) break;
        top.push_back(std::move(n));
        at = e;
    }
    const Node *objects = nullptr, *conns = nullptr;
    for (const Node &n : top) { if (n.name == "Objects") objects = &n; else if (n.name == "Connections") conns = &n; }
    if (!objects) { rt_set_error("%s: no Objects section", path); return RT_ERR_IO; }
    // geometry id -> the Model it is connected to ("OO" child, parent)
    std::map<long long, const Node *> models;
    for (const Node &o : objects->kids)
        if (o.name == "Model" && !o.props.empty() && !o.props[0].ints.empty()) models[o.props[0].ints[0]] = &o;
    std::map<long long, long long> parent_of;
    if (conns)
        for (const Node &c : conns->kids)
            if (c.name == "C" && c.props.size() >= 3 && c.props[0].str == "OO" && !c.props[1].ints.empty() && !c.props[2].ints.empty())
                parent_of[c.props[1].ints[0]] = c.props[2].ints[0];
    verts.clear();
    idx.clear();
    bool unsupported = false;
    for (const Node &o : objects->kids) {
        if (o.name != "Geometry" || o.props.size() < 3 || o.props[2].str != "Mesh") continue;
        M34 xf = identity34();
        if (!o.props[0].ints.empty()) {
            std::map<long long, long long>::const_iterator p = parent_of.find(o.props[0].ints[0]);
            if (p != parent_of.end()) {
                std::map<long long, const Node *>::const_iterator m = models.find(p->second);
                if (m != models.end()) {
                    xf = model_transform(*m->second, &unsupported);
                    // a Model whose own parent is another Model (not the scene root, id 0) has an inherited transform
                    std::map<long long, long long>::const_iterator gp = parent_of.find(p->second);
                    if (gp != parent_of.end() && gp->second != 0 && models.count(gp->second)) unsupported = true;
                }
            }
        }
        if (unsupported) {
            rt_set_error("%s: a mesh needs pivots, pre / post rotations, geometric or inherited transforms, which this reader does not apply", path);
            return RT_ERR_UNSUPPORTED;
        }
        RT_TRY(add_geometry(o, xf, verts, idx));
    }
    if (idx.empty()) { rt_set_error("%s: no mesh geometry with polygons", path); return RT_ERR_IO; }
    return RT_OK;
}

extern "C" int rt_fbx_read(const char *path, rt_vertex *verts, uint32_t capacity_verts, uint32_t *indices, uint32_t capacity_tris,
                           uint32_t *n_verts, uint32_t *n_tris)
{
    RT_REQUIRE(path && n_verts && n_tris, "null argument");
    std::vector<rt_vertex> v;
    std::vector<uint32_t> idx;
    RT_TRY(rt_fbx_parse(path, v, idx));
    *n_verts = (uint32_t)v.size();
    *n_tris = (uint32_t)(idx.size() / 3);
    if (!verts && !indices) return RT_OK;
    RT_REQUIRE(verts && indices && capacity_verts >= v.size() && capacity_tris >= idx.size() / 3, "buffers too small");
    memcpy(verts, v.data(), v.size() * sizeof(rt_vertex));
    memcpy(indices, idx.data(), idx.size() * sizeof(uint32_t));
    return RT_OK;
}
