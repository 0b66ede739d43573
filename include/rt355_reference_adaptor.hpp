// rt355_reference_adaptor.hpp -- glue a maintainer of liz425/RayTracer adds next to main.cpp.
//
// Include AFTER the reference's own headers (vector3D.h, color.h, texture.h, shader.h,
// anyobject.h): it walks a `vector<AnyObject*>` (Scene::objs, main.cpp:229), down-casts by
// GetObjectType() ("Plane" anyobject.h:89, "Sphere" :161, "Environment" :223, "Quadric" and its subclasses
// :348-440, "Mesh" :672) and fills the plain-C
// descriptors of rt355.h.  It holds no reference code, only member reads.
//
// Two members of the reference are private and cannot be read from outside:
//   * Texture::pixmap (texture.h:24)  -> the caller supplies a texel accessor;
//   * Shader::light   (shader.h:26)   -> the caller passes its own vector<light_src*>
//                                        (main.cpp:622), as it passed it to the Shader ctor.
#pragma once
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt355.h"

struct Rt355Flattened {
    std::vector<rt355_object> objects;
    std::vector<rt355_triangle> triangles;
    std::vector<rt355_texture> textures;
    rt355_scene_desc desc() const {
        rt355_scene_desc d;
        std::memset(&d, 0, sizeof d);
        d.objects = objects.data(); d.n_objects = (int32_t)objects.size();
        d.triangles = triangles.data(); d.n_triangles = (int32_t)triangles.size();
        d.textures = textures.data(); d.n_textures = (int32_t)textures.size();
        return d;
    }
};

namespace rt355_adaptor_detail {
inline void put(double *d, const Vector3D &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline void put(double *d, const Point3D &v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
inline void put(uint8_t *d, const Color &c) { d[0] = c.r; d[1] = c.g; d[2] = c.b; }
} // namespace rt355_adaptor_detail

// texels(t) must return t's width*height*3 bytes in the order Texture keeps them in memory
// (after the B<->R swap of texture.h:73-78), valid until rt355_scene_create returns.
inline Rt355Flattened rt355_flatten(const std::vector<AnyObject *> &objs,
                                    const std::function<const unsigned char *(const Texture *)> &texels) {
    using namespace rt355_adaptor_detail;
    Rt355Flattened f;
    for (AnyObject *a : objs) {
        rt355_object o;
        std::memset(&o, 0, sizeof o);
        o.ks = a->ks; o.fresnel = a->fresnel; o.ior = a->IOR;
        o.texture_type = a->texture_type;
        o.tex0 = o.tex1 = -1;
        const std::string kind = a->GetObjectType();
        if (kind == "Plane") {
            Plane *p = (Plane *)a;
            o.type = RT355_OBJ_PLANE;
            put(o.p, p->pi); put(o.n, p->ni); put(o.n0, p->n0); put(o.n1, p->n1); put(o.rgb, p->clr);
        } else if (kind == "Sphere" || kind == "Environment") { // Environment derives from Sphere (anyobject.h:171)
            Sphere *s = (Sphere *)a;
            o.type = kind == "Sphere" ? RT355_OBJ_SPHERE : RT355_OBJ_ENVIRONMENT;
            put(o.p, s->pCenter); o.radius = s->radius;
            put(o.n0, s->n0); put(o.n1, s->n1); put(o.n2, s->n2); put(o.rgb, s->clr);
        } else if (kind == "Quadric" || kind == "Cone" || kind == "Ellipsoid" || kind == "Cylinder" ||
                   kind == "EllipticParaboloid" || kind == "HyperbolicParaboloid") { // anyobject.h:348,367,386,404,422,440
            Quadric *q = (Quadric *)a;
            o.type = RT355_OBJ_QUADRIC;
            put(o.p, q->pCenter); put(o.n0, q->nCoor[0]); put(o.n1, q->nCoor[1]); put(o.n2, q->nCoor[2]);
            for (int k = 0; k < 3; k++) o.s[k] = q->sCoor[k];
            for (int k = 0; k < 5; k++) o.a[k] = q->aCoor[k];
            put(o.rgb, q->clr); // Quadric::clr (:240), the one CalcIntersect returns
        } else if (kind == "Mesh") {
            Mesh *m = (Mesh *)a;
            o.type = RT355_OBJ_MESH;
            put(o.rgb, m->clr);
            o.tri_begin = (int32_t)f.triangles.size();
            o.tri_count = (int32_t)m->triangles.size();
            for (const Triangle &t : m->triangles) {
                rt355_triangle q;
                put(q.p0, t.p0); put(q.p1, t.p1); put(q.p2, t.p2);
                q.uv0[0] = t.UV_p0.x; q.uv0[1] = t.UV_p0.y;
                q.uv1[0] = t.UV_p1.x; q.uv1[1] = t.UV_p1.y;
                q.uv2[0] = t.UV_p2.x; q.uv2[1] = t.UV_p2.y;
                f.triangles.push_back(q);
            }
        } else {
            throw std::runtime_error("rt355: object type '" + kind + "' is not on the accelerated path");
        }
        for (size_t k = 0; k < a->textures.size() && k < 2; k++) {
            const Texture *t = a->textures[k];
            rt355_texture q;
            q.rgb = texels(t); q.sx = t->sx; q.sy = t->sy; q.width = t->width; q.height = t->height;
            put(q.n0, t->n0); put(q.n1, t->n1); put(q.n2, t->n2); put(q.p_center, t->p_center); // texture.h:32-35
            (k == 0 ? o.tex0 : o.tex1) = (int32_t)f.textures.size();
            f.textures.push_back(q);
        }
        f.objects.push_back(o);
    }
    return f;
}

inline std::vector<rt355_light> rt355_lights(const std::vector<light_src *> &lights) {
    using namespace rt355_adaptor_detail;
    std::vector<rt355_light> out;
    for (const light_src *l : lights) {
        rt355_light q;
        std::memset(&q, 0, sizeof q);
        q.type = l->type;
        put(q.position, l->position); put(q.direction, l->direction); put(q.color, l->color);
        put(q.n0, l->n0); put(q.n1, l->n1); q.sx = l->sx; q.sy = l->sy; // the area CalcSoftShadow aims at (light.h:19-23)
        out.push_back(q);
    }
    return out;
}

inline rt355_shader rt355_shader_of(const Shader &shd, const std::vector<rt355_light> &lights) {
    rt355_shader s;
    std::memset(&s, 0, sizeof s);
    s.lights = lights.data(); s.n_lights = (int32_t)lights.size();
    rt355_adaptor_detail::put(s.ambient, shd.ambient_color);
    rt355_adaptor_detail::put(s.specular, shd.specular_color);
    rt355_adaptor_detail::put(s.border, shd.border_color);
    return s;
}

// The direction set of the gather shaders: the global `geo` that main.cpp:639 fills with Geodesic().
inline int rt355_set_directions(rt355_scene *scene, const std::vector<Vector3D> &geo) {
    std::vector<double> xyz;
    xyz.reserve(geo.size() * 3);
    for (const Vector3D &g : geo) { xyz.push_back(g.x); xyz.push_back(g.y); xyz.push_back(g.z); }
    return rt355_scene_set_directions(scene, xyz.data(), (int32_t)geo.size());
}

// SceneT = the reference's Scene (main.cpp:215-265): only its public camera fields are read.
template <class SceneT>
inline rt355_camera rt355_camera_of(const SceneT &sce) {
    rt355_camera c;
    rt355_adaptor_detail::put(c.v_up, sce.v_up); rt355_adaptor_detail::put(c.v_view, sce.v_view);
    rt355_adaptor_detail::put(c.p_eye, sce.p_eye);
    c.dist = sce.dist; c.s_x = sce.s_x; c.s_y = sce.s_y;
    return c;
}
