// rt355_host.hpp -- C++ host side above the C ABI (include/rt355.h).
//
// Mirrors the scene-building interface of liz425/RayTracer so that a main() written for the
// reference (image_synthesis/main.cpp:526-642) reads the same here: Plane / Sphere / Mesh
// with the reference's constructor argument order, light_src, Shader, Scene with the public
// camera fields, and setPixels(scene, shader, alias).  These classes only HOLD data; all
// rendering happens in librt355.so on the GPU.  Errors surface as rt355::Error (the reference
// prints to cout and carries on; a GPU library should not).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rt355.h"

namespace rt355 {

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

struct Vector3D { double x = 0, y = 0, z = 0; Vector3D() {} Vector3D(double a, double b, double c) : x(a), y(b), z(c) {} };
using Point3D = Vector3D;
struct Vector2D { double x = 0, y = 0; Vector2D() {} Vector2D(double a, double b) : x(a), y(b) {} };

// Normalize of vector3D.h:79-88 (same operation order: the light direction is stored normalised)
inline Vector3D Normalize(Vector3D v) {
    double l = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    if (l != 0) return Vector3D(v.x / l, v.y / l, v.z / l);
    return v;
}

struct Color { // color.h:16-31
    unsigned char r = 0, g = 0, b = 0;
    Color() {}
    Color(int r_, int g_, int b_) : r((unsigned char)r_), g((unsigned char)g_), b((unsigned char)b_) {}
    explicit Color(unsigned int c) : r((c & 0xff0000) >> 16), g((c & 0xff00) >> 8), b(c & 0xff) {}
};

struct Texture { // a loaded texture.h Texture: texels in memory order, scale factors
    std::vector<unsigned char> rgb; int width = 0, height = 0; double sx = 1, sy = 1;
    // the projector of a solid texture (texture.h:32-35); p00 and p_projector are derived from it by the library
    Vector3D n0{1, 0, 0}, n1{0, 1, 0}, n2{0, 0, 1};
    Point3D p_center{0, 0, 0};
    Texture() {}
    // Texture(string filename, double sx, double sy) of texture.h:42-79: a 54-byte header with the width at
    // byte 18 and the height at byte 22, then 3 * width * height bytes taken as they come (no row padding,
    // no bit-depth check, bottom row first), B and R swapped in place.  A file that cannot be opened is an
    // error here as it is there (`throw "Argument Exception"`, texture.h:52); a file shorter than its header
    // promises leaves the missing texels 0 (the reference leaves them uninitialised).
    explicit Texture(const std::string &filename, double sx_in = 1, double sy_in = 1, Vector3D n0_in = {1, 0, 0},
                     Vector3D n1_in = {0, 1, 0}, Vector3D n2_in = {0, 0, 1}, Point3D p_center_in = {0, 0, 0})
        : sx(sx_in), sy(sy_in), n0(n0_in), n1(n1_in), n2(n2_in), p_center(p_center_in) {
        FILE *fp = std::fopen(filename.c_str(), "rb");
        if (!fp) throw Error("cannot open texture " + filename);
        unsigned char info[54];
        std::memset(info, 0, sizeof info);
        size_t got = std::fread(info, 1, 54, fp);
        (void)got;
        std::memcpy(&width, info + 18, 4);
        std::memcpy(&height, info + 22, 4);
        if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 28)) {
            std::fclose(fp);
            throw Error("texture " + filename + ": bad size in the header");
        }
        rgb.assign((size_t)3 * width * height, 0);
        got = std::fread(rgb.data(), 1, rgb.size(), fp);
        std::fclose(fp);
        for (size_t i = 0; i + 2 < rgb.size(); i += 3) std::swap(rgb[i], rgb[i + 2]);
    }
};

class AnyObject { // anyobject.h:17-36
public:
    double IOR = 1, fresnel = 1, ks = 0;
    int texture_type = 0;
    std::vector<std::shared_ptr<Texture>> textures;
    virtual ~AnyObject() {}
    virtual const std::string GetObjectType() const = 0;
    void AddTexture(std::shared_ptr<Texture> t) { textures.push_back(std::move(t)); }
};

class Plane : public AnyObject { // anyobject.h:40-93
public:
    Color clr; Point3D pi; Vector3D n0, ni;
    Plane(Point3D pi_in, Vector3D ni_in, Vector3D n0_in, Color clr_in, double ks_in = 0)
        : clr(clr_in), pi(pi_in), n0(n0_in), ni(ni_in) { ks = ks_in; }
    const std::string GetObjectType() const override { return "Plane"; }
};

class Sphere : public AnyObject { // anyobject.h:97-164
public:
    Color clr; Point3D pCenter; double radius; Vector3D n0{1, 0, 0}, n1{0, 1, 0}, n2{0, 0, 1};
    Sphere(Point3D c, double r, Color clr_in, Vector3D n0_in = {1, 0, 0}, Vector3D n1_in = {0, 1, 0},
           Vector3D n2_in = {0, 0, 1}, double ks_in = 0)
        : clr(clr_in), pCenter(c), radius(r), n0(n0_in), n1(n1_in), n2(n2_in) { ks = ks_in; }
    const std::string GetObjectType() const override { return "Sphere"; }
};

class Environment : public Sphere { // anyobject.h:171-225: the constructor normalises the axes (:185-187)
public:
    Environment(Point3D c, double r, Color clr_in, Vector3D n0_in, Vector3D n1_in, Vector3D n2_in, double ks_in = 0)
        : Sphere(c, r, clr_in, Normalize(n0_in), Normalize(n1_in), Normalize(n2_in), ks_in) {}
    const std::string GetObjectType() const override { return "Environment"; }
};

class Quadric : public AnyObject { // anyobject.h:231-350: sum of aCoor[i] * ((nCoor[i] . (p - pCenter)) / sCoor[i])^2 ...
public:
    Point3D pCenter; int aCoor[5] = {0, 0, 0, 1, 0}; // a02, a12, a22, a21, a00; the default is a plane (:242-249)
    Vector3D nCoor[3]; double sCoor[3] = {0, 0, 0}; Color clr;
    Quadric() {}
    Quadric(int a0, int a1, int a2, int a3, int a4) { aCoor[0] = a0; aCoor[1] = a1; aCoor[2] = a2; aCoor[3] = a3; aCoor[4] = a4; }
    void SetN(Vector3D n0, Vector3D n1, Vector3D n2) { nCoor[0] = n0; nCoor[1] = n1; nCoor[2] = n2; }
    void SetS(float s0, float s1, float s2) { sCoor[0] = s0; sCoor[1] = s1; sCoor[2] = s2; } // floats, as :265
    void SetPCenter(Point3D pc) { pCenter = pc; }
    void SetColor(Color c) { clr = c; }
    void SetKs(double k) { ks = k; }
    const std::string GetObjectType() const override { return "Quadric"; }
};
// anyobject.h:355-442: the subclasses set aCoor and their name
class Cone : public Quadric { public: Cone() : Quadric(1, 1, -1, 0, 0) {} const std::string GetObjectType() const override { return "Cone"; } };
class Ellipsoid : public Quadric { public: Ellipsoid() : Quadric(1, 1, 1, 0, -1) {} const std::string GetObjectType() const override { return "Ellipsoid"; } };
class Cylinder : public Quadric { public: Cylinder() : Quadric(1, 1, 0, 0, -1) {} const std::string GetObjectType() const override { return "Cylinder"; } };
class EllipticParaboloid : public Quadric { public: EllipticParaboloid() : Quadric(1, 1, 0, -1, 0) {} const std::string GetObjectType() const override { return "EllipticParaboloid"; } };
class HyperbolicParaboloid : public Quadric { public: HyperbolicParaboloid() : Quadric(1, -1, 0, -1, 0) {} const std::string GetObjectType() const override { return "HyperbolicParaboloid"; } };

struct Triangle { // anyobject.h:444-532 (data members only)
    Point3D p0, p1, p2; Vector2D UV_p0, UV_p1, UV_p2; // (0, 0) until SetUV, as the reference's members are
    Triangle(Point3D a, Point3D b, Point3D c) : p0(a), p1(b), p2(c) {}
    void SetUV(Vector2D t0, Vector2D t1, Vector2D t2) { UV_p0 = t0; UV_p1 = t1; UV_p2 = t2; }
};

// ---- OBJ files, read the way Mesh::Mesh reads them (anyobject.h:546-644) ------------------------------------
// What the reference's reader does, as rules: a line is told by its first two characters ("v ", "vt", "vn", "f ",
// anything else is skipped); numbers are taken from a line as operator>> takes them (the leading ones that parse,
// 0 for the rest); a face line is matched against one of three layouts with sscanf and "%i" (so 0x.. and 0..
// prefixes count), and only its first three corners are used; the layout in use starts as the simplest and moves
// on to the next when a line does not fit -- for good; a face line that fits no layout ends the reading with what
// was read so far; a file that cannot be opened reads as empty.
namespace objfile {
struct Corner { int vertex = 1, uv = 1, normal = 1; }; // 1-based; 1 where the layout has no such field (:597-599)
struct FaceLayout { const char *pattern; int per_corner; };
static const FaceLayout kLayouts[3] = {
    {"f %i %i %i", 1},                       // anyobject.h:603
    {"f %i/%i %i/%i %i/%i", 2},              // :610
    {"f %i/%i/%i %i/%i/%i %i/%i/%i", 3},     // :617
};
inline bool scan_face(const std::string &line, const FaceLayout &lay, Corner corner[3]) {
    int field[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
    const int got = std::sscanf(line.c_str(), lay.pattern, &field[0], &field[1], &field[2], &field[3], &field[4], &field[5],
                                &field[6], &field[7], &field[8]);
    if (got != 3 * lay.per_corner) return false;
    for (int c = 0; c < 3; c++) {
        const int *f = field + c * lay.per_corner;
        corner[c] = Corner();
        corner[c].vertex = f[0];
        if (lay.per_corner >= 2) corner[c].uv = f[1];
        if (lay.per_corner >= 3) corner[c].normal = f[2];
    }
    return true;
}
inline void leading_numbers(const std::string &text, double *out, int count) {
    for (int i = 0; i < count; i++) out[i] = 0;
    std::istringstream in(text);
    for (int i = 0; i < count && (in >> out[i]); i++) {}
}
// Calls on_line(kind, rest) for every line of the file, kind = the line's first two characters; stops when
// on_line returns false.  Returns false if the file cannot be opened.
template <class OnLine>
inline bool for_each_line(const std::string &path, OnLine on_line) {
    std::ifstream in(path.c_str());
    if (in.fail()) return false;
    std::string line;
    while (in.peek() != EOF) {
        std::getline(in, line);
        if (!on_line(line.substr(0, 2), line)) break;
    }
    return true;
}
} // namespace objfile

class Mesh : public AnyObject { // anyobject.h:534-675
public:
    Color clr{150, 200, 110};
    std::vector<Triangle> triangles;
    explicit Mesh(std::vector<Triangle> tris = {}, double ks_in = 0) : triangles(std::move(tris)) { ks = ks_in; }
    // Mesh(string filePath, Point3D mesh_origin, double scale, double ks) of anyobject.h:546-644: vertices become
    // mesh_origin + (p - 0) * scale (:575); a face takes its texture coordinates only if a `vt` line came before it
    // (:631), vertex normals are counted and used nowhere on the path.  An index outside the lists read so far is
    // undefined behaviour in the reference and an Error here.
    explicit Mesh(const std::string &filePath, Point3D mesh_origin = Point3D(0, 0, 0), double scale = 1, double ks_in = 0) {
        ks = ks_in;
        std::vector<Point3D> vertices;
        std::vector<Vector2D> uvs;
        size_t n_normals = 0;
        int layout = 0; // index into objfile::kLayouts, only ever grows
        auto listed = [](int index, size_t size) { return index >= 1 && (size_t)index <= size; };
        objfile::for_each_line(filePath, [&](const std::string &kind, const std::string &line) {
            double num[3];
            if (kind == "v ") {
                objfile::leading_numbers(line.substr(2), num, 3);
                vertices.push_back(Point3D(mesh_origin.x + (num[0] - 0) * scale, mesh_origin.y + (num[1] - 0) * scale,
                                           mesh_origin.z + (num[2] - 0) * scale));
            } else if (kind == "vt") {
                objfile::leading_numbers(line.substr(2), num, 2);
                uvs.push_back(Vector2D(num[0], num[1]));
            } else if (kind == "vn") {
                n_normals++;
            } else if (kind == "f ") {
                objfile::Corner c[3];
                while (layout < 3 && !objfile::scan_face(line, objfile::kLayouts[layout], c)) layout++;
                if (layout == 3) return false; // fits no layout: the reading ends here (:618-621)
                for (int k = 0; k < 3; k++) {
                    if (!listed(c[k].vertex, vertices.size()))
                        throw Error(filePath + ": face refers to a vertex that was not read: " + line);
                    if (n_normals && !listed(c[k].normal, n_normals))
                        throw Error(filePath + ": face refers to a normal that was not read: " + line);
                    if (!uvs.empty() && !listed(c[k].uv, uvs.size()))
                        throw Error(filePath + ": face refers to a texture coordinate that was not read: " + line);
                }
                Triangle tri(vertices[c[0].vertex - 1], vertices[c[1].vertex - 1], vertices[c[2].vertex - 1]);
                if (!uvs.empty()) tri.SetUV(uvs[c[0].uv - 1], uvs[c[1].uv - 1], uvs[c[2].uv - 1]);
                triangles.push_back(tri);
            }
            return true;
        });
    }
    const std::string GetObjectType() const override { return "Mesh"; }
};

struct light_src { // light.h:12-56
    int type; Point3D position; Vector3D direction; Vector3D color;
    Vector3D n0{1, 0, 0}, n1{0, 1, 0}, n2{0, 0, 1}; // the area of a type 3 light (light.h:19-23)
    double sx = 1, sy = 1;
    double spot_angle = 20;                        // (read by the hard spot light only, which is compiled out: shader.h:21)
    light_src(int type_in, Point3D position_in, Vector3D direction_in, Vector3D color_in)
        : type(type_in), position(position_in), direction(Normalize(direction_in)), color(color_in) {}
    light_src(int type_in, Point3D position_in, Vector3D direction_in, Vector3D color_in, double spot_angle_in)
        : type(type_in), position(position_in), direction(Normalize(direction_in)), color(color_in), spot_angle(spot_angle_in) {}
    light_src(int type_in, Point3D position_in, Vector3D direction_in, Vector3D color_in, Vector3D n0_in, Vector3D n1_in,
              Vector3D n2_in, double sx_in, double sy_in)
        : type(type_in), position(position_in), direction(Normalize(direction_in)), color(color_in), n0(n0_in), n1(n1_in),
          n2(n2_in), sx(sx_in), sy(sy_in) {}
};

struct Shader { // shader.h:25-37
    std::vector<light_src *> light;
    Color ambient_color, specular_color, border_color;
    Shader(std::vector<light_src *> l, Color a, Color s, Color b)
        : light(std::move(l)), ambient_color(a), specular_color(s), border_color(b) {}
};

struct Scene { // main.cpp:215-265 (camera fields; SetCamera is done by the library)
    Vector3D v_up{0, 0, 1}, v_view{0, 1, 0}; Point3D p_eye{0, -50, 0};
    double dist = 5, s_x = 10, s_y = 10;
    std::vector<AnyObject *> objs;
};

namespace detail {
inline void put(double *d, const Vector3D &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline void put(uint8_t *d, const Color &c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; }
inline void check(int rc, const char *what) {
    if (rc != RT355_OK) throw Error(std::string(what) + ": " + rt355_last_error());
}
} // namespace detail

// An uploaded scene.  Build once, render many frames (camera, lights and shader colours may change).
class DeviceScene {
public:
    explicit DeviceScene(const std::vector<AnyObject *> &objs) {
        std::vector<rt355_object> o; std::vector<rt355_triangle> t; std::vector<rt355_texture> x;
        for (const AnyObject *a : objs) {
            rt355_object q; std::memset(&q, 0, sizeof q);
            q.ks = a->ks; q.fresnel = a->fresnel; q.ior = a->IOR; q.texture_type = a->texture_type;
            q.tex0 = q.tex1 = -1;
            if (auto p = dynamic_cast<const Plane *>(a)) {
                q.type = RT355_OBJ_PLANE; detail::put(q.p, p->pi); detail::put(q.n, p->ni); detail::put(q.n0, p->n0);
                detail::put(q.rgb, p->clr);
            } else if (auto s = dynamic_cast<const Sphere *>(a)) {
                q.type = dynamic_cast<const Environment *>(a) ? RT355_OBJ_ENVIRONMENT : RT355_OBJ_SPHERE;
                detail::put(q.p, s->pCenter); q.radius = s->radius;
                detail::put(q.n0, s->n0); detail::put(q.n1, s->n1); detail::put(q.n2, s->n2); detail::put(q.rgb, s->clr);
            } else if (auto k = dynamic_cast<const Quadric *>(a)) {
                q.type = RT355_OBJ_QUADRIC; detail::put(q.p, k->pCenter); detail::put(q.rgb, k->clr);
                detail::put(q.n0, k->nCoor[0]); detail::put(q.n1, k->nCoor[1]); detail::put(q.n2, k->nCoor[2]);
                for (int i = 0; i < 3; i++) q.s[i] = k->sCoor[i];
                for (int i = 0; i < 5; i++) q.a[i] = k->aCoor[i];
            } else if (auto m = dynamic_cast<const Mesh *>(a)) {
                q.type = RT355_OBJ_MESH; detail::put(q.rgb, m->clr);
                q.tri_begin = (int32_t)t.size(); q.tri_count = (int32_t)m->triangles.size();
                for (const Triangle &tr : m->triangles) {
                    rt355_triangle r;
                    detail::put(r.p0, tr.p0); detail::put(r.p1, tr.p1); detail::put(r.p2, tr.p2);
                    r.uv0[0] = tr.UV_p0.x; r.uv0[1] = tr.UV_p0.y; r.uv1[0] = tr.UV_p1.x; r.uv1[1] = tr.UV_p1.y;
                    r.uv2[0] = tr.UV_p2.x; r.uv2[1] = tr.UV_p2.y;
                    t.push_back(r);
                }
            } else {
                throw Error("object type '" + a->GetObjectType() + "' is not on the accelerated path");
            }
            for (size_t k = 0; k < a->textures.size() && k < 2; k++) {
                const Texture &tx = *a->textures[k];
                rt355_texture r; r.rgb = tx.rgb.data(); r.sx = tx.sx; r.sy = tx.sy; r.width = tx.width; r.height = tx.height;
                detail::put(r.n0, tx.n0); detail::put(r.n1, tx.n1); detail::put(r.n2, tx.n2); detail::put(r.p_center, tx.p_center);
                (k == 0 ? q.tex0 : q.tex1) = (int32_t)x.size();
                x.push_back(r);
            }
            o.push_back(q);
        }
        rt355_scene_desc d; std::memset(&d, 0, sizeof d);
        d.objects = o.data(); d.n_objects = (int32_t)o.size();
        d.triangles = t.data(); d.n_triangles = (int32_t)t.size();
        d.textures = x.data(); d.n_textures = (int32_t)x.size();
        detail::check(rt355_scene_create(&d, &h_), "rt355_scene_create");
    }
    ~DeviceScene() { rt355_scene_destroy(h_); }
    DeviceScene(const DeviceScene &) = delete;
    DeviceScene &operator=(const DeviceScene &) = delete;
    rt355_scene *handle() const { return h_; }
    // the global `geo` (main.cpp:36, filled at main.cpp:639): the direction set of the gather shaders (shading2 / shading3)
    void set_directions(const std::vector<Vector3D> &geo) {
        std::vector<double> xyz;
        for (const Vector3D &g : geo) { xyz.push_back(g.x); xyz.push_back(g.y); xyz.push_back(g.z); }
        detail::check(rt355_scene_set_directions(h_, xyz.data(), (int32_t)geo.size()), "rt355_scene_set_directions");
    }
private:
    rt355_scene *h_ = nullptr;
};

// Geodesic(string filePath) of main.cpp:408-450: the `v x y z` lines of an OBJ file, moved so that their mean is
// the origin and normalised (:443-447, in that operation order).  A file that cannot be opened gives an empty set,
// as there (every hit then renders black).
inline std::vector<Vector3D> Geodesic(const std::string &filePath) {
    std::vector<Vector3D> geo;
    double total[3] = {0, 0, 0};
    objfile::for_each_line(filePath, [&](const std::string &kind, const std::string &line) {
        if (kind != "v ") return true;
        double num[3];
        objfile::leading_numbers(line.substr(2), num, 3);
        geo.push_back(Vector3D(num[0], num[1], num[2]));
        for (int a = 0; a < 3; a++) total[a] = total[a] + num[a];
        return true;
    });
    const double count = (double)(int)geo.size();
    const Vector3D mean(total[0] / count, total[1] / count, total[2] / count);
    const Vector3D shift(0 - mean.x, 0 - mean.y, 0 - mean.z); // Point3D(0, 0, 0) - sum, main.cpp:446
    for (Vector3D &g : geo) g = Normalize(Vector3D(g.x + shift.x, g.y + shift.y, g.z + shift.z));
    return geo;
}

// setPixels(Scene sce, Shader shd, int alias) of main.cpp:339-376; width, height and pixmap are
// globals there (main.cpp:33-34) and arguments here.  `shading` picks the renderer: RT355_SHADING_LIGHTS is
// CalcSubPixel with Shader::shading, RT355_SHADING_BLEED what HEAD of the reference calls (CalcSubPixel2 with
// shading3, main.cpp:313,331), on the directions given to DeviceScene::set_directions.  Returns the ray statistics.
inline rt355_stats setPixels(DeviceScene &dev, const Scene &sce, const Shader &shd, int alias, int width, int height,
                             unsigned char *pixmap, int row_begin = 0, int row_end = 0,
                             int shading = RT355_SHADING_LIGHTS, uint32_t seed = 0) {
    rt355_camera c;
    detail::put(c.v_up, sce.v_up); detail::put(c.v_view, sce.v_view); detail::put(c.p_eye, sce.p_eye);
    c.dist = sce.dist; c.s_x = sce.s_x; c.s_y = sce.s_y;
    std::vector<rt355_light> ls;
    for (const light_src *l : shd.light) {
        rt355_light q; std::memset(&q, 0, sizeof q);
        q.type = l->type; detail::put(q.position, l->position); detail::put(q.direction, l->direction);
        detail::put(q.color, l->color);
        detail::put(q.n0, l->n0); detail::put(q.n1, l->n1); q.sx = l->sx; q.sy = l->sy;
        ls.push_back(q);
    }
    rt355_shader s; std::memset(&s, 0, sizeof s);
    s.lights = ls.data(); s.n_lights = (int32_t)ls.size();
    detail::put(s.ambient, shd.ambient_color); detail::put(s.specular, shd.specular_color);
    detail::put(s.border, shd.border_color);
    rt355_opts o; std::memset(&o, 0, sizeof o);
    o.width = width; o.height = height; o.alias = alias; o.row_begin = row_begin; o.row_end = row_end;
    o.mode = RT355_MODE_PACKET; o.shading = shading; o.seed = seed;
    rt355_stats st; std::memset(&st, 0, sizeof st);
    detail::check(rt355_render(dev.handle(), &c, &s, &o, pixmap, &st), "rt355_render");
    return st;
}

// 24-bit BMP, bottom-up rows, B,G,R -- what SaveImage (main.cpp:456-517) produces, with a
// correct 3-bytes-per-pixel size field and rows padded to 4 bytes.
inline void SaveImage(const char *file, int width, int height, const unsigned char *pixmap) {
    FILE *f = std::fopen(file, "wb");
    if (!f) throw Error(std::string("cannot open ") + file);
    const int row = (width * 3 + 3) & ~3;
    const uint32_t data = (uint32_t)row * height, size = 54 + data;
    unsigned char h[54]; std::memset(h, 0, sizeof h);
    h[0] = 'B'; h[1] = 'M'; std::memcpy(h + 2, &size, 4); h[10] = 54; h[14] = 40;
    std::memcpy(h + 18, &width, 4); std::memcpy(h + 22, &height, 4); h[26] = 1; h[28] = 24; std::memcpy(h + 34, &data, 4);
    std::fwrite(h, 1, 54, f);
    std::vector<unsigned char> line(row, 0);
    for (int y = 0; y < height; y++) {
        for (int x = 0; x < width; x++) {
            const unsigned char *p = pixmap + ((size_t)y * width + x) * 3;
            line[3 * x] = p[2]; line[3 * x + 1] = p[1]; line[3 * x + 2] = p[0];
        }
        std::fwrite(line.data(), 1, row, f);
    }
    std::fclose(f);
}

} // namespace rt355
