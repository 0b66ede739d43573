// ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Built only where
// /root/reference is present (this container); output goes to oracle/_ref/ only.
//
// Compiles the reference's OWN headers (vector3D.h, color.h, texture.h, light.h,
// view.h, anyobject.h, shader.h) where they lie, and drives their real
// CalcIntersect / Shader::shading / Texture::GetColor* code from rt355 scene
// descriptors, so that the C restatement (rt_oracle.c) and the HIP library can be
// checked against what the reference itself computes.
//
// image_synthesis/main.cpp is NOT compiled: it needs <GLUT/GLUT.h>, which this image
// does not have, and no stand-in is written for it.  The ~25 lines of main.cpp that sit
// on the path (Scene::SetCamera 255-262, CalcView 269-276, the closest-hit loop of
// CalcSubPixel 42-91, the alias==0 branch of CalcPixel 326-332 and setPixels 355-370)
// are therefore re-expressed below with the reference's own vector/colour types.  That
// part is pinned by 18 pixel values recorded from a run of main.cpp itself (tests/golden/meta.json, baseline_md_samples).
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

// Shader::shading2 / shading3 draw their two rotation angles with rand() (shader.h:106-107,149-150).  Inside
// the reference's headers that name is pointed at the harness's own source of numbers, which hands out the
// counter-based values of rt355_rotation for the sub-sample being shaded -- the one change of input that
// lets frames be compared pixel by pixel.  The standard headers are included above, so only the
// reference's code sees the macro.
// Their sin() and cos() (shader.h:115,119,160,164; nothing else on the path calls them) are pointed at
// correctly rounded evaluations: the C library's last bit differs between libraries, and a gather ray that
// starts on a sphere's surface hits or misses that sphere by the sign of a rounding error.
#include <quadmath.h>
static int rt355_harness_rand();
// (the reference asks for the same two angles once per direction: the last two are remembered)
static double rt355_harness_trig(double x, int want_cos) {
    static thread_local struct { double x, s, c; bool set; } memo[2];
    static thread_local int next = 0;
    for (int i = 0; i < 2; i++)
        if (memo[i].set && memo[i].x == x) return want_cos ? memo[i].c : memo[i].s;
    memo[next].x = x; memo[next].s = (double)sinq((__float128)x); memo[next].c = (double)cosq((__float128)x);
    memo[next].set = true;
    const double r = want_cos ? memo[next].c : memo[next].s;
    next ^= 1;
    return r;
}
static double rt355_harness_sin(double x) { return rt355_harness_trig(x, 0); }
static double rt355_harness_cos(double x) { return rt355_harness_trig(x, 1); }
#define rand rt355_harness_rand
#define sin rt355_harness_sin
#define cos rt355_harness_cos

// same order as main.cpp:17-21
#include "vector3D.h"
#include "color.h"
#include "texture.h"
#include "shader.h"
#include "anyobject.h"

#undef rand
#undef sin
#undef cos

#include "../include/rt355.h"
#include "../include/rt355_reference_adaptor.hpp"

// the sub-sample whose hit is being shaded, and how many numbers it has drawn
static thread_local struct { uint32_t seed; int x, y, sample, drawn; } g_draw;
static int rt355_harness_rand() {
    // rt355_rotation returns value / RAND_MAX with value a 31-bit integer; hand that integer back
    const double v = rt355x_rotation(g_draw.seed, g_draw.x, g_draw.y, g_draw.sample, g_draw.drawn++);
    return (int)(v * 2147483647.0 + 0.5);
}

namespace {

Vector3D vec(const double *p) { return Vector3D(p[0], p[1], p[2]); }
Point3D pnt(const double *p) { return Point3D(p[0], p[1], p[2]); }

// Write a 24-bit BMP the way Texture::Texture expects to read it (texture.h:54-78):
// 54-byte header with width at 18 and height at 22, then width*height*3 bytes, B,G,R.
std::string write_bmp(const rt355_texture &t, const std::string &dir, int id) {
    std::string path = dir + "/rt355_tex_" + std::to_string(id) + ".bmp";
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return "";
    unsigned char hdr[54];
    memset(hdr, 0, sizeof hdr);
    hdr[0] = 'B'; hdr[1] = 'M'; hdr[10] = 54; hdr[14] = 40; hdr[26] = 1; hdr[28] = 24;
    memcpy(hdr + 18, &t.width, 4);
    memcpy(hdr + 22, &t.height, 4);
    fwrite(hdr, 1, 54, f);
    size_t n = (size_t)t.width * t.height;
    std::vector<unsigned char> bgr(n * 3);
    for (size_t i = 0; i < n; i++) {
        bgr[3 * i] = t.rgb[3 * i + 2];
        bgr[3 * i + 1] = t.rgb[3 * i + 1];
        bgr[3 * i + 2] = t.rgb[3 * i];
    }
    fwrite(bgr.data(), 1, bgr.size(), f);
    fclose(f);
    return path;
}

struct RefScene {
    std::vector<AnyObject *> objs;
    std::vector<int> kinds;
    ~RefScene() {
        for (AnyObject *o : objs) {
            for (Texture *t : o->textures) delete t; // shared pointers are not used below
            delete o;
        }
    }
};

// Build reference objects from a descriptor.  Returns false on I/O failure.
bool build(const rt355_scene_desc *d, const char *tmpdir, RefScene &rs) {
    std::vector<std::string> tex_paths(d->n_textures);
    for (int i = 0; i < d->n_textures; i++) {
        tex_paths[i] = write_bmp(d->textures[i], tmpdir ? tmpdir : "/tmp", i);
        if (tex_paths[i].empty()) return false;
    }
    // Mesh's ctor reports a missing OBJ file on cout; keep the test output clean.
    std::streambuf *old = std::cout.rdbuf();
    std::ostringstream sink;
    for (int i = 0; i < d->n_objects; i++) {
        const rt355_object &o = d->objects[i];
        Color clr(o.rgb[0], o.rgb[1], o.rgb[2]);
        AnyObject *obj = nullptr;
        if (o.type == RT355_OBJ_PLANE) {
            obj = new Plane(pnt(o.p), vec(o.n), vec(o.n0), clr, o.ks);
        } else if (o.type == RT355_OBJ_SPHERE) {
            obj = new Sphere(pnt(o.p), o.radius, clr, vec(o.n0), vec(o.n1), vec(o.n2), o.ks);
        } else if (o.type == RT355_OBJ_ENVIRONMENT) {
            Environment *env = new Environment(pnt(o.p), o.radius, clr, vec(o.n0), vec(o.n1), vec(o.n2), o.ks);
            env->n0 = vec(o.n0); env->n1 = vec(o.n1); env->n2 = vec(o.n2); // the descriptor holds the members as stored
            obj = env;
        } else if (o.type == RT355_OBJ_QUADRIC) {
            // a subclass where the coefficients are one's (anyobject.h:355-442), the general constructor otherwise
            Quadric *q = nullptr;
            auto is = [&](int a0, int a1, int a2, int a3, int a4) {
                return o.a[0] == a0 && o.a[1] == a1 && o.a[2] == a2 && o.a[3] == a3 && o.a[4] == a4;
            };
            if (is(1, 1, -1, 0, 0)) q = new Cone();
            else if (is(1, 1, 1, 0, -1)) q = new Ellipsoid();
            else if (is(1, 1, 0, 0, -1)) q = new Cylinder();
            else if (is(1, 1, 0, -1, 0)) q = new EllipticParaboloid();
            else if (is(1, -1, 0, -1, 0)) q = new HyperbolicParaboloid();
            else if (is(0, 0, 0, 1, 0)) q = new Quadric();
            else q = new Quadric(o.a[0], o.a[1], o.a[2], o.a[3], o.a[4]);
            q->SetPCenter(pnt(o.p));
            q->SetN(vec(o.n0), vec(o.n1), vec(o.n2));
            q->SetS(o.s[0], o.s[1], o.s[2]); // floats there (anyobject.h:265): the descriptor holds float values
            q->SetColor(clr);
            q->SetKs(o.ks);
            obj = q;
        } else {
            std::cout.rdbuf(sink.rdbuf());
            Mesh *m = new Mesh("/nonexistent/rt355.obj", Point3D(0, 0, 0), 1, o.ks);
            std::cout.rdbuf(old);
            m->clr = clr;
            for (int k = 0; k < o.tri_count; k++) {
                const rt355_triangle &t = d->triangles[o.tri_begin + k];
                // what the OBJ face branch does, anyobject.h:625-636
                Triangle tri(pnt(t.p0), pnt(t.p1), pnt(t.p2), o.ks);
                tri.SetUV(Vector2D(t.uv0[0], t.uv0[1]), Vector2D(t.uv1[0], t.uv1[1]),
                          Vector2D(t.uv2[0], t.uv2[1]));
                tri.clr = m->clr;
                m->triangles.push_back(tri);
            }
            obj = m;
        }
        obj->fresnel = o.fresnel;
        obj->IOR = o.ior;
        obj->texture_type = o.texture_type;
        for (int id : {o.tex0, o.tex1}) {
            if (id < 0) continue;
            const rt355_texture &t = d->textures[id];
            static const double unit[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, zero[3] = {0, 0, 0};
            const bool plain = !memcmp(t.n0, unit, 24) && !memcmp(t.n1, unit + 3, 24) && !memcmp(t.n2, unit + 6, 24) &&
                               !memcmp(t.p_center, zero, 24);
            if (plain) obj->AddTexture(tex_paths[id], t.sx, t.sy); // anyobject.h:32-35
            else obj->textures.push_back(new Texture(tex_paths[id], t.sx, t.sy, vec(t.n0), vec(t.n1), vec(t.n2), pnt(t.p_center)));
        }
        rs.objs.push_back(obj);
        rs.kinds.push_back(o.type);
    }
    return true;
}

struct Cam { Point3D p_eye, p00; Vector3D n0, n1; double s_x, s_y; };

// Scene::SetCamera (main.cpp:255-262), re-expressed
Cam set_camera(const rt355_camera *c) {
    Cam k;
    Vector3D v_view = vec(c->v_view), v_up = vec(c->v_up);
    k.p_eye = pnt(c->p_eye);
    Vector3D n_view = Normalize(v_view);
    Point3D p_center = k.p_eye + n_view * c->dist;
    Vector3D v0 = CrossProduct(v_view, v_up);
    k.n0 = Normalize(v0);
    k.n1 = CrossProduct(k.n0, n_view);
    k.p00 = p_center - k.n0 * (c->s_x / 2) - k.n1 * (c->s_y / 2);
    k.s_x = c->s_x; k.s_y = c->s_y;
    return k;
}

// CalcView (main.cpp:269-276), re-expressed
View calc_view(const Cam &k, double x_per, double y_per) {
    Point3D pp = k.p00 + k.n0 * (x_per * k.s_x) + k.n1 * (y_per * k.s_y);
    Vector3D npe = Normalize(pp - k.p_eye);
    return View(k.p_eye, npe);
}

// CalcSubPixel (main.cpp:42-158), re-expressed line by line with the reference's classes: the call on the
// object the ray is inside of (:53-63), closest-hit loop and shading call (:65-91), reflection recursion
// (:93-112, thresholds :27-28), refraction branches (:116-154).  The recursion has no bound of its own; `depth`
// stops a chain that does not end before the stack does and reports it through g_too_deep.
thread_local int g_too_deep = 0;
// Soft shadows: Shader::shading (shader.h:42-96) with the call it keeps commented out at :91, CalcSoftShadow, in
// place of CalcShadow at :90.  That one line cannot be switched from outside the class, so the function's few
// lines are re-expressed here; everything they call -- TextureMapping, SolidTexture, Get3DFunction, CalcDiffuse,
// CalcSpecular, CalcBorder and CalcSoftShadow itself -- is the reference's own code.  `light` is the vector the
// Shader was constructed with (its own copy is private, shader.h:26).
std::vector<light_src *> *g_soft_lights = nullptr; // non-null: render with soft shadows
Color shading_soft(Shader &shd, std::vector<light_src *> &light, Color cm0, Color cm1, Point3D ph, Vector3D &nh, View eye,
                   bool no_border, std::vector<AnyObject *> &objs, int object_index, Vector2D UV, double m, double n) {
    int light_num = (int)light.size();
    Color clr_sum = Color(0, 0, 0);
    Color clr = cm0;
    if (objs[object_index]->texture_type == 1) {
        tuple<Color, Color, Vector3D> r = shd.TextureMapping(ph, nh, objs[object_index], false, UV);
        cm0 = get<0>(r); cm1 = get<1>(r); nh = get<2>(r);
    } else if (objs[object_index]->texture_type == 2) {
        tuple<Color, Color> r = shd.SolidTexture(ph, nh, objs[object_index], cm0);
        cm0 = get<0>(r); cm1 = get<1>(r);
    } else if (objs[object_index]->texture_type == 3) {
        cm0 = objs[object_index]->textures[0]->Get3DFunction(ph, Point3D(0, 0, 20));
        cm1 = shd.ambient_color;
    }
    if (objs[object_index]->GetObjectType() == "Environment") return cm0;
    for (int light_index = 0; light_index < light_num; light_index++) {
        Vector3D nhl;
        if (light[light_index]->type == 0) nhl = light[light_index]->direction;
        else nhl = Normalize(ph - light[light_index]->position);
        clr = shd.CalcDiffuse(cm0, cm1, ph, nh, eye, nhl, light_index);
        clr = shd.CalcSpecular(clr, ph, nh, eye, nhl, light_index);
        if (!no_border) clr = shd.CalcBorder(clr, ph, nh, eye, nhl);
        clr = shd.CalcSoftShadow(clr, ph, nh, nhl, light_index, objs, m, n);
        clr_sum = clr_sum / (light_index + 1.0) * light_index + clr / (light_index + 1.0);
    }
    return clr_sum;
}
// m, n: the last two parameters of CalcSubPixel (main.cpp:42), which it hands to Shader::shading (:88); the calls of
// the recursion leave them at their default 0 (:109,126,138,148)
Color sub_pixel(View eye, std::vector<AnyObject *> &objs, Shader &shd, int reflection_times = 0,
                double reflection_ktotal = 1.0, int outside = 1, int object_index = 0, int depth = 0, double m = 0, double n = 0) {
    double t_min = FLT_MAX;
    Color clr = Color(0, 0, 0);
    Color cm0(0, 0, 0);
    Point3D ph(0, 0, 0);
    Vector3D nh(0, 0, 0);
    bool no_border = true;
    Vector2D UV(0, 0);
    if (depth > 2000) { g_too_deep = 1; return clr; }
    if (objs.empty()) return clr;
    if (outside <= -1 && object_index >= 0) {
        tuple<Color, double, Point3D, Vector3D, int> r = objs[object_index]->CalcIntersect(eye, outside);
        t_min = get<1>(r); cm0 = get<0>(r); ph = get<2>(r); nh = get<3>(r);
        std::string kind = objs[object_index]->GetObjectType();
        no_border = (kind == "Plane" || kind == "Mesh");
        if (kind == "Mesh") UV = ((Mesh *)objs[object_index])->GetUV();
    } else {
        for (int i = 0; i < (int)objs.size(); i++) {
            tuple<Color, double, Point3D, Vector3D, int> r = objs[i]->CalcIntersect(eye);
            double t = get<1>(r);
            if (t >= 0 && t < t_min) {
                t_min = t; object_index = i;
                cm0 = get<0>(r); ph = get<2>(r); nh = get<3>(r);
                std::string kind = objs[i]->GetObjectType();
                no_border = (kind == "Plane" || kind == "Mesh");
                if (kind == "Mesh") UV = ((Mesh *)objs[i])->GetUV();
            }
        }
    }
    if (t_min != FLT_MAX)
        clr = g_soft_lights ? shading_soft(shd, *g_soft_lights, cm0, shd.ambient_color, ph, nh, eye, no_border, objs, object_index, UV, m, n)
                            : shd.shading(cm0, shd.ambient_color, ph, nh, eye, no_border, objs, object_index, UV, m, n);
    double ks = objs[object_index]->ks;
    reflection_ktotal *= ks;
    if (outside >= 1) reflection_times++;
    Color clr_reflection;
    Color clr_refraction;
    if (outside >= 1 && reflection_times <= 3 && reflection_ktotal >= 0.05) {
        Vector3D reflect_npe = eye.npe - nh * 2 * DotProduct(nh, eye.npe);
        View reflect_view = View(ph, reflect_npe);
        clr_reflection = clr * (1 - ks) + sub_pixel(reflect_view, objs, shd, reflection_times, reflection_ktotal, 1, 0, depth + 1) * ks;
    } else {
        clr_reflection = clr;
    }
    double fresnel = objs[object_index]->fresnel;
    if (fresnel < 1) {
        if (outside >= 1) {
            Vector3D V = eye.npe * (-1);
            double c = DotProduct(nh, V);
            double IOR = 1 / objs[object_index]->IOR;
            Vector3D T = V * (-IOR) + nh * (IOR * c - sqrt(pow(IOR * c, 2) - IOR * IOR + 1));
            T = T.normalize();
            View refract_view = View(ph, T);
            clr_refraction = sub_pixel(refract_view, objs, shd, reflection_times, reflection_ktotal, -1, object_index, depth + 1);
            clr = clr_reflection * fresnel + clr_refraction * (1 - fresnel);
        } else {
            Vector3D V = eye.npe * (-1);
            double c = DotProduct(nh, V);
            double IOR = objs[object_index]->IOR;
            double delta = (pow(IOR * c, 2) - IOR * IOR + 1);
            if (delta > 0) {
                Vector3D T = V * (-IOR) + nh * (IOR * c - sqrt(delta));
                T = T.normalize();
                View refract_view = View(ph, T);
                clr_refraction = sub_pixel(refract_view, objs, shd, reflection_times, reflection_ktotal, 1, object_index, depth + 1);
            } else {
                Vector3D reflect_npe;
                reflect_npe = eye.npe - nh * 2 * DotProduct(nh, eye.npe);
                View reflect_view = View(ph, reflect_npe);
                if (outside-- <= -5) outside = 1;
                clr_refraction = sub_pixel(reflect_view, objs, shd, reflection_times, reflection_ktotal, outside, object_index, depth + 1);
            }
            clr = clr_refraction;
        }
    } else {
        clr = clr_reflection;
    }
    return clr;
}

// CalcSubPixel2 as CalcPixel calls it (main.cpp:163-209 with outside = 1), re-expressed; the shader call is the
// reference's own Shader::shading3 (main.cpp:205) or Shader::shading2.
Color sub_pixel2(View eye, std::vector<AnyObject *> &objs, Shader &shd, std::vector<Vector3D> &geo, int shading) {
    double t_min = FLT_MAX;
    Color clr = Color(0, 0, 0);
    Color cm0(0, 0, 0);
    Point3D ph(0, 0, 0);
    Vector3D nh(0, 0, 0);
    for (int i = 0; i < (int)objs.size(); i++) {
        tuple<Color, double, Point3D, Vector3D, int> r = objs[i]->CalcIntersect(eye);
        double t = get<1>(r);
        if (t >= 0 && t < t_min) { t_min = t; cm0 = get<0>(r); ph = get<2>(r); nh = get<3>(r); }
    }
    if (t_min != FLT_MAX)
        clr = (shading == RT355_SHADING_BLEED) ? shd.shading3(cm0, shd.ambient_color, ph, nh, objs, geo)
                                               : shd.shading2(cm0, shd.ambient_color, ph, nh, objs, geo);
    return clr;
}

std::vector<light_src *> make_lights(const rt355_shader *sh) {
    std::vector<light_src *> ls;
    for (int i = 0; i < sh->n_lights; i++) {
        const rt355_light &l = sh->lights[i];
        light_src *p = new light_src(l.type, pnt(l.position), vec(l.direction), vec(l.color));
        p->direction = vec(l.direction); // the descriptor holds the member as stored
        p->n0 = vec(l.n0); p->n1 = vec(l.n1); p->sx = l.sx; p->sy = l.sy; // the area (light.h:19-23,49-53)
        ls.push_back(p);
    }
    return ls;
}

} // namespace

extern "C" {

int ref_render_gather(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
                      const rt355_opts *opts, const double *dirs, int n_dirs, unsigned char *pixmap,
                      const char *tmpdir);
int ref_render(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
               const rt355_opts *opts, unsigned char *pixmap, const char *tmpdir) {
    return ref_render_gather(d, cam, sh, opts, nullptr, 0, pixmap, tmpdir);
}
int ref_render_gather(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
                      const rt355_opts *opts, const double *dirs, int n_dirs, unsigned char *pixmap,
                      const char *tmpdir) {
    std::vector<Vector3D> geo;
    for (int i = 0; i < n_dirs; i++) geo.push_back(Vector3D(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]));
    const int shading = opts->shading;
    const int alias_ = opts->alias;
    auto shade = [&](View eye, std::vector<AnyObject *> &objs, Shader &shd, int x, int y, int sample) {
        if (shading == RT355_SHADING_LIGHTS) return sub_pixel(eye, objs, shd);
        if (shading == RT355_SHADING_SOFT) { // the call of main.cpp:312 / :330, see rt355x_soft_args
            int32_t object_index = 0; double m = 0, n = 0;
            if (alias_ > 0) rt355x_soft_args(opts->seed, x, y, sample % alias_, sample / alias_, alias_, &object_index, &m, &n);
            if (object_index >= (int)objs.size()) object_index = 0;
            return sub_pixel(eye, objs, shd, 0, 1.0, 1, object_index, 0, m, n);
        }
        g_draw.seed = opts->seed; g_draw.x = x; g_draw.y = y; g_draw.sample = sample; g_draw.drawn = 0;
        return sub_pixel2(eye, objs, shd, geo, shading);
    };
    RefScene rs;
    if (!build(d, tmpdir, rs)) return -1;
    std::vector<light_src *> ls = make_lights(sh);
    Shader shd(ls, Color(sh->ambient[0], sh->ambient[1], sh->ambient[2]),
               Color(sh->specular[0], sh->specular[1], sh->specular[2]),
               Color(sh->border[0], sh->border[1], sh->border[2]));
    g_soft_lights = (shading == RT355_SHADING_SOFT) ? &ls : nullptr;
    Cam k = set_camera(cam);
    int W = opts->width, H = opts->height;
    int y0 = opts->row_begin, y1 = opts->row_end;
    if (y0 == 0 && y1 == 0) y1 = H;
    const int alias = opts->alias;
    for (int y = y0; y < y1; y++)
        for (int x = 0; x < W; x++) {
            Color c;
            if (alias > 0) { // CalcPixel's sub-sample loop with the reference's Color; jitter from rt355_jitter, not rand()
                double red = 0, green = 0, blue = 0;
                for (int q = 0; q < alias; q++)
                    for (int p = 0; p < alias; p++) {
                        double pointX = (rt355x_jitter(opts->seed, x, y, q * alias + p, 0) + p) / alias + x;
                        double pointY = (rt355x_jitter(opts->seed, x, y, q * alias + p, 1) + q) / alias + y;
                        View eye = calc_view(k, pointX / W, pointY / H);
                        Color clr_sub = shade(eye, rs.objs, shd, x, y, q * alias + p);
                        red += clr_sub.r; green += clr_sub.g; blue += clr_sub.b;
                    }
                red = red / (alias * alias); green = green / (alias * alias); blue = blue / (alias * alias);
                c = Color(red, green, blue);
            } else {
                View eye = calc_view(k, (double)x / W, (double)y / H);
                c = shade(eye, rs.objs, shd, x, y, 0);
            }
            size_t i = ((size_t)y * W + x) * 3;
            pixmap[i] = c.r; pixmap[i + 1] = c.g; pixmap[i + 2] = c.b;
        }
    g_soft_lights = nullptr;
    for (light_src *l : ls) delete l;
    if (g_too_deep) { g_too_deep = 0; return -2; } // a refraction chain that did not end
    return 0;
}

// objs[obj]->CalcIntersect(View(pe,npe)); out = t, ph[3], nh[3], uv[2]
int ref_intersect(const rt355_scene_desc *d, int obj, const double *pe, const double *npe,
                  double *out, const char *tmpdir) {
    RefScene rs;
    if (!build(d, tmpdir, rs)) return -1;
    View eye(pnt(pe), vec(npe));
    tuple<Color, double, Point3D, Vector3D, int> r = rs.objs[obj]->CalcIntersect(eye);
    out[0] = get<1>(r);
    Point3D ph = get<2>(r);
    Vector3D nh = get<3>(r);
    out[1] = ph.x; out[2] = ph.y; out[3] = ph.z;
    out[4] = nh.x; out[5] = nh.y; out[6] = nh.z;
    out[7] = out[8] = 0;
    if (rs.kinds[obj] == RT355_OBJ_MESH && out[0] >= 0) {
        Vector2D uv = ((Mesh *)rs.objs[obj])->GetUV();
        out[7] = uv.x; out[8] = uv.y;
    }
    return 0;
}

// Many rays against one object without rebuilding it: rays = n x 6, out = n x 9
int ref_intersect_many(const rt355_scene_desc *d, int obj, const double *rays, int n,
                       double *out, const char *tmpdir) {
    RefScene rs;
    if (!build(d, tmpdir, rs)) return -1;
    for (int i = 0; i < n; i++) {
        const double *q = rays + 6 * i;
        double *o = out + 9 * i;
        View eye(pnt(q), vec(q + 3));
        tuple<Color, double, Point3D, Vector3D, int> r = rs.objs[obj]->CalcIntersect(eye);
        o[0] = get<1>(r);
        Point3D ph = get<2>(r);
        Vector3D nh = get<3>(r);
        o[1] = ph.x; o[2] = ph.y; o[3] = ph.z;
        o[4] = nh.x; o[5] = nh.y; o[6] = nh.z;
        o[7] = o[8] = 0;
        if (rs.kinds[obj] == RT355_OBJ_MESH && o[0] >= 0) {
            Vector2D uv = ((Mesh *)rs.objs[obj])->GetUV();
            o[7] = uv.x; o[8] = uv.y;
        }
    }
    return 0;
}

// Color operators (color.h:33-51): op 0 `c*v`, 1 `c/v`, 2 `c+c2`, 3 `c*Vector3D`
void ref_color_op(int op, const unsigned char *c, const unsigned char *c2, double v, double v2,
                  double v3, unsigned char *out) {
    Color a(c[0], c[1], c[2]), r = a;
    if (op == 0) r = a * v;
    else if (op == 1) r = a / v;
    else if (op == 2) r = a + Color(c2[0], c2[1], c2[2]);
    else if (op == 3) r = a * Vector3D(v, v2, v3);
    out[0] = r.r; out[1] = r.g; out[2] = r.b;
}

// Texture::GetColorPlane (kind 0) / GetColorSphere (kind 1), texture.h:81-122
int ref_texture(const rt355_texture *t, int kind, const double *ab, int n, unsigned char *out,
                const char *tmpdir) {
    std::string path = write_bmp(*t, tmpdir ? tmpdir : "/tmp", 9999);
    if (path.empty()) return -1;
    Texture tex(path, t->sx, t->sy);
    for (int i = 0; i < n; i++) {
        Color r = kind ? tex.GetColorSphere(ab[2 * i], ab[2 * i + 1]) : tex.GetColorPlane(ab[2 * i], ab[2 * i + 1]);
        out[3 * i] = r.r; out[3 * i + 1] = r.g; out[3 * i + 2] = r.b;
    }
    return 0;
}

// The reference's own OBJ reader: Mesh::Mesh(filePath, mesh_origin, scale) (anyobject.h:546-644).  out receives up
// to cap triangles, 15 doubles each (p0 p1 p2 UV_p0 UV_p1 UV_p2); returns the number of triangles read.
int ref_load_mesh(const char *path, const double *origin, double scale, double *out, int cap) {
    std::streambuf *old = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf());
    Mesh m(path, pnt(origin), scale, 0);
    std::cout.rdbuf(old);
    int n = (int)m.triangles.size();
    for (int i = 0; i < n && i < cap; i++) {
        const Triangle &t = m.triangles[i];
        double *o = out + 15 * i;
        o[0] = t.p0.x; o[1] = t.p0.y; o[2] = t.p0.z; o[3] = t.p1.x; o[4] = t.p1.y; o[5] = t.p1.z;
        o[6] = t.p2.x; o[7] = t.p2.y; o[8] = t.p2.z;
        o[9] = t.UV_p0.x; o[10] = t.UV_p0.y; o[11] = t.UV_p1.x; o[12] = t.UV_p1.y; o[13] = t.UV_p2.x; o[14] = t.UV_p2.y;
    }
    return n;
}

// The reference's own BMP reader: Texture::Texture(filename) (texture.h:42-79).  Its texel array is private, so
// the texels are read back through GetColorPlane at texel corners (exact when width and height are powers of
// two: the bilinear weights are then 1, 0, 0, 0).  out receives width * height * 3 bytes in memory order.
int ref_load_texture(const char *path, int *width, int *height, unsigned char *out, int cap) {
    try {
        Texture t(path, 1, 1);
        *width = t.width; *height = t.height;
        if ((long long)t.width * t.height * 3 > cap) return -2;
        for (int iy = 0; iy < t.height; iy++)
            for (int ix = 0; ix < t.width; ix++) {
                Color c = t.GetColorPlane((double)ix / t.width, (double)iy / t.height);
                unsigned char *o = out + ((size_t)iy * t.width + ix) * 3;
                o[0] = c.r; o[1] = c.g; o[2] = c.b;
            }
        return 0;
    } catch (const char *) {
        return -1;
    }
}

// Build reference objects from a descriptor, run the maintainer-side adaptor
// (include/rt355_reference_adaptor.hpp) over them and compare what it produces with the
// descriptor we started from.  Returns 0 when identical, else a positive code.
int ref_adaptor_roundtrip(const rt355_scene_desc *d, const rt355_shader *sh, const char *tmpdir) {
    RefScene rs;
    if (!build(d, tmpdir, rs)) return -1;
    // Texture::pixmap is private: hand the adaptor the bytes the BMPs were written from
    std::vector<const Texture *> order;
    std::vector<const unsigned char *> bytes;
    for (size_t i = 0; i < rs.objs.size(); i++) {
        const rt355_object &o = d->objects[i];
        int ids[2] = {o.tex0, o.tex1};
        for (size_t k = 0; k < rs.objs[i]->textures.size(); k++) {
            order.push_back(rs.objs[i]->textures[k]);
            bytes.push_back(d->textures[ids[k]].rgb);
        }
    }
    auto texels = [&](const Texture *t) -> const unsigned char * {
        for (size_t i = 0; i < order.size(); i++) if (order[i] == t) return bytes[i];
        return nullptr;
    };
    Rt355Flattened f = rt355_flatten(rs.objs, texels);
    if ((int)f.objects.size() != d->n_objects) return 1;
    if ((int)f.triangles.size() != d->n_triangles) return 2;
    for (int i = 0; i < d->n_objects; i++) {
        const rt355_object &a = d->objects[i], &b = f.objects[i];
        if (a.type != b.type || a.texture_type != b.texture_type || memcmp(a.rgb, b.rgb, 3)) return 10 + i;
        if (a.ks != b.ks || a.fresnel != b.fresnel || a.ior != b.ior) return 10 + i;
        if (a.type == RT355_OBJ_PLANE && (memcmp(a.p, b.p, 24) || memcmp(a.n, b.n, 24) || memcmp(a.n0, b.n0, 24))) return 10 + i;
        if ((a.type == RT355_OBJ_SPHERE || a.type == RT355_OBJ_ENVIRONMENT) && (memcmp(a.p, b.p, 24) || a.radius != b.radius || memcmp(a.n0, b.n0, 72))) return 10 + i;
        if (a.type == RT355_OBJ_QUADRIC && (memcmp(a.p, b.p, 24) || memcmp(a.n0, b.n0, 72) || memcmp(a.s, b.s, 24) || memcmp(a.a, b.a, 20))) return 10 + i;
        if (a.type == RT355_OBJ_MESH && (a.tri_begin != b.tri_begin || a.tri_count != b.tri_count)) return 10 + i;
        if ((a.tex0 >= 0) != (b.tex0 >= 0) || (a.tex1 >= 0) != (b.tex1 >= 0)) return 10 + i;
        if (a.tex0 >= 0) {
            const rt355_texture &ta = d->textures[a.tex0], &tb = f.textures[b.tex0];
            if (ta.width != tb.width || ta.height != tb.height || ta.sx != tb.sx || ta.sy != tb.sy || ta.rgb != tb.rgb) return 10 + i;
            if (memcmp(ta.n0, tb.n0, 96)) return 10 + i; // n0, n1, n2, p_center
        }
    }
    if (d->n_triangles && memcmp(d->triangles, f.triangles.data(), sizeof(rt355_triangle) * d->n_triangles)) return 3;
    std::vector<light_src *> ls = make_lights(sh);
    std::vector<rt355_light> back = rt355_lights(ls);
    int rc = 0;
    for (int i = 0; i < sh->n_lights; i++)
        if (back[i].type != sh->lights[i].type || memcmp(back[i].position, sh->lights[i].position, 72)) rc = 4;
    Shader shd(ls, Color(sh->ambient[0], sh->ambient[1], sh->ambient[2]),
               Color(sh->specular[0], sh->specular[1], sh->specular[2]),
               Color(sh->border[0], sh->border[1], sh->border[2]));
    rt355_shader s2 = rt355_shader_of(shd, back);
    if (memcmp(s2.ambient, sh->ambient, 3) || memcmp(s2.specular, sh->specular, 3) || memcmp(s2.border, sh->border, 3)) rc = 5;
    for (light_src *l : ls) delete l;
    return rc;
}

} // extern "C"
