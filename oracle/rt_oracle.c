/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, double-precision CPU restatement of the per-pixel render path of
 * liz425/RayTracer (image_synthesis/main.cpp + headers), written from the algorithm,
 * operation by operation, so that its frames are byte-identical to the reference's.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * HIP library never links or calls anything in this file.
 *
 * Parity pin: oracle/ref_harness.cpp compiles the reference's own headers and compares
 * (tests/golden/ holds the derived vectors: frames, CalcIntersect values, the Color operator
 * table, texture lookups); 18 pixel values recorded from a run of the reference's main.cpp and
 * the hashes of the default scene's 800x600 and 3840x2160 frames (b5fabdf4953d0714,
 * 6647a8b414e21d93, b7bc333b9d9ae34d: what a build of main.cpp itself was measured to render)
 * are checked in tests/test_oracle_golden.py.
 *
 * Every function cites the reference lines (relative to image_synthesis/) it follows.
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (no -ffast-math: the
 * operation order and roundings below are the contract).
 */
#include <float.h>
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rt355.h"

/* texture.h:17 defines PI first; shader.h:15-17 and main.cpp:23-25 are #ifndef guarded,
 * so 3.1415927 is the value everywhere. */
#define REF_PI 3.1415927

typedef struct { double x, y, z; } v3;
typedef struct { unsigned char r, g, b; } col;

/* ---- vector3D.h ------------------------------------------------------------------------- */
static v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }   /* vector3D.h:33,118 */
static v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :38,113,123 */
static v3 vmul(v3 a, double s) { return V(a.x * s, a.y * s, a.z * s); }     /* :46 */
static v3 vdiv(v3 a, double s) { return V(a.x / s, a.y / s, a.z / s); }     /* :42 */
static double vlen(v3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } /* :50 */
static double dot(v3 u, v3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }  /* :90 */
static v3 cross(v3 u, v3 v) {                                                /* :94 */
    return V(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
static v3 normalize(v3 v) {                                                  /* :79-88, :55-63 */
    double l = vlen(v);
    if (l != 0) return vdiv(v, l);
    return v;
}
static v3 from3(const double *p) { return V(p[0], p[1], p[2]); }

/* ---- color.h ------------------------------------------------------------------------------ */
/* Color(char,char,char) (color.h:25) receives double or int expressions.  On x86-64 the
 * double->char conversion is cvttsd2si to 32 bits followed by a byte truncation: toward
 * zero, wrap mod 256, out-of-range/NaN -> 0x80000000 -> 0. */
static unsigned char q8(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return 0;
    return (unsigned char)(int32_t)x;
}
static col C(unsigned char r, unsigned char g, unsigned char b) { col c = {r, g, b}; return c; }
static col cmul(col c, double v) { return C(q8(c.r * v), q8(c.g * v), q8(c.b * v)); }          /* :33 */
static col cmulv(col c, v3 v) { return C(q8(c.r * v.x), q8(c.g * v.y), q8(c.b * v.z)); }       /* :37 */
static col cdivd(col c, double v) { return C(q8(c.r / v), q8(c.g / v), q8(c.b / v)); }         /* :41 */
static col cadd(col a, col b) {                                                                 /* :45 */
    return C((unsigned char)(a.r + b.r), (unsigned char)(a.g + b.g), (unsigned char)(a.b + b.b));
}

/* x86 (int)double as the reference's `(int)x` casts compile (texture.h:92-97, shader.h:253). */
static int trunc_i32(double x) {
    if (!(x > -2147483649.0 && x < 2147483648.0)) return INT32_MIN;
    return (int32_t)x;
}

/* ---- flattened scene ------------------------------------------------------------------------ */
typedef struct {
    v3 p0, p1, p2, normal, p_center;
    double uv0[2], uv1[2], uv2[2];
} otri;

typedef struct {
    const rt355_scene_desc *d;
    otri *tris;
} oscene;

/* Triangle ctor, anyobject.h:456-461 */
static void tri_setup(const rt355_triangle *t, otri *o) {
    o->p0 = from3(t->p0); o->p1 = from3(t->p1); o->p2 = from3(t->p2);
    o->normal = normalize(cross(vsub(o->p1, o->p0), vsub(o->p2, o->p0)));
    o->p_center = V((o->p0.x + o->p1.x + o->p2.x) / 3.0, (o->p0.y + o->p1.y + o->p2.y) / 3.0,
                    (o->p0.z + o->p1.z + o->p2.z) / 3.0);
    memcpy(o->uv0, t->uv0, sizeof o->uv0);
    memcpy(o->uv1, t->uv1, sizeof o->uv1);
    memcpy(o->uv2, t->uv2, sizeof o->uv2);
}

typedef struct { v3 pe, npe; } ray;
/* View(Point3D, Vector3D) normalises its direction (view.h:23) */
static ray mkview(v3 pe, v3 npe) { ray r; r.pe = pe; r.npe = normalize(npe); return r; }

typedef struct {
    double t;     /* -1 = miss (anyobject.h:81,151,500,520,664) */
    v3 ph, nh;
    double uvx, uvy; /* Mesh::UV_hitpoint (anyobject.h:655) */
} hit;

/* Plane::CalcIntersect, anyobject.h:69-87 (outside == 0 on every call of this path) */
static hit plane_isect(const rt355_object *o, ray e) {
    hit h; memset(&h, 0, sizeof h);
    v3 ni = from3(o->n), pi = from3(o->p);
    int outside = (dot(ni, vsub(e.pe, pi)) <= 0) ? -1 : 1;
    if (dot(ni, e.npe) >= 0) { h.t = -1; return h; }
    h.t = -dot(ni, vsub(e.pe, pi)) / dot(ni, e.npe);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(ni, outside);
    return h;
}

/* Sphere::CalcIntersect, anyobject.h:128-158 */
static hit sphere_isect(const rt355_object *o, ray e) {
    hit h; memset(&h, 0, sizeof h);
    v3 c = from3(o->p);
    double r = o->radius;
    v3 ec = vsub(e.pe, c);
    int outside = (dot(ec, ec) - r * r <= -0.00001) ? -1 : 1;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    if (delta < 0) { h.t = -1; return h; }
    h.t = B - outside * sqrt(delta);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(normalize(vsub(h.ph, c)), outside);
    return h;
}

/* Environment::CalcIntersect, anyobject.h:199-220.  Every call on the path goes through an AnyObject* (so the
 * default argument is the base class's 0, anyobject.h:23) or passes 0 itself (shader.h:128,174): `inside` is 0 with
 * the origin inside the sphere, -1 outside; t is the far root. */
static hit environment_isect(const rt355_object *o, ray e) {
    hit h; memset(&h, 0, sizeof h);
    v3 c = from3(o->p);
    double r = o->radius;
    v3 ec = vsub(e.pe, c);
    int inside = 0;
    if (dot(ec, ec) - r * r >= 0.00001) inside = -1;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    if (delta < 0) { h.t = -1; return h; }
    h.t = B + sqrt(delta);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(normalize(vsub(c, h.ph)), inside);
    return h;
}

/* Quadric::CalcIntersect, anyobject.h:283-345, outside == 0 on entry.  The formulas are kept as written there
 * (the divisions and powers repeated per call); pow(x, 2) is x * x.  In the A == 0 branch the reference's loop
 * over the axes starts from an uninitialised index (`for (int i; ...`, :316): it is taken to run from 0 like its
 * twin at :333, which is what the harness build (g++ -O2) does too. */
static hit quadric_isect_k(const rt355_object *o, ray e, int outside_in) {
    hit h; memset(&h, 0, sizeof h);
    v3 pc = from3(o->p);
    v3 n[3] = {from3(o->n0), from3(o->n1), from3(o->n2)};
    const int32_t *a = o->a;
    const double *s = o->s;
    double Fp = a[3] / s[2] * dot(n[2], vsub(e.pe, pc)) + a[4];
    for (int i = 0; i < 3; i++) { double v = dot(n[i], vsub(e.pe, pc)) / s[i]; Fp += a[i] * (v * v); }
    int outside = outside_in ? outside_in : ((Fp < 0) ? -1 : 1); /* :289-296 */
    double A = 0;
    double B = a[3] / s[2] * dot(n[2], e.npe);
    double C = a[3] / s[2] * dot(n[2], vsub(e.pe, pc)) + a[4];
    for (int i = 0; i < 3; i++) {
        double u = dot(n[i], e.npe) / s[i];
        A += a[i] * (u * u);
        B += 2 * a[i] / (s[i] * s[i]) * dot(n[i], e.npe) * dot(n[i], vsub(e.pe, pc));
        double v = dot(n[i], vsub(e.pe, pc)) / s[i];
        C += a[i] * (v * v);
    }
    if (A == 0) {
        if (B != 0) {
            h.t = -(C / B);
            h.ph = vadd(e.pe, vmul(e.npe, h.t));
            v3 nh = vmul(n[2], a[3] / s[2]);
            for (int i = 0; i < 3; i++) nh = vadd(nh, vmul(n[i], 2 * a[i] / (s[i] * s[i]) * dot(n[i], vsub(h.ph, pc))));
            h.nh = vmul(normalize(nh), outside);
            return h;
        }
        h.t = -1;
        return h;
    }
    double delta = B * B - 4 * A * C;
    if (delta >= 0) {
        h.t = (-B - sqrt(delta)) / (2 * A);
        h.ph = vadd(e.pe, vmul(e.npe, h.t));
        v3 nh = vmul(n[2], a[3] / s[2]);
        for (int i = 0; i < 3; i++) nh = vadd(nh, vmul(n[i], 2 * a[i] / (s[i] * s[i]) * dot(n[i], vsub(h.ph, pc))));
        h.nh = vmul(normalize(nh), outside);
        return h;
    }
    h.t = -1;
    return h;
}

static hit quadric_isect(const rt355_object *o, ray e) { return quadric_isect_k(o, e, 0); }

/* Triangle::CalcIntersect, anyobject.h:475-522; u,v returned for GetUV (:524-526) */
static hit tri_isect(const otri *t, ray e, double *u_out, double *v_out) {
    hit h; memset(&h, 0, sizeof h);
    int outside = (dot(t->normal, vsub(e.pe, t->p_center)) <= 0) ? -1 : 1;
    if (dot(t->normal, e.npe) >= 0) { h.t = -1; return h; }
    double tt = -dot(t->normal, vsub(e.pe, t->p_center)) / dot(t->normal, e.npe);
    if ((outside < 0 && tt > 0) || (tt < 0)) { h.t = -1; return h; }
    v3 ph = vadd(e.pe, vmul(e.npe, tt));
    v3 AA = cross(vsub(t->p1, t->p0), vsub(t->p2, t->p0));
    v3 A1 = cross(vsub(t->p2, ph), vsub(t->p0, ph));
    v3 A2 = cross(vsub(t->p0, ph), vsub(t->p1, ph));
    double u = (AA.x != 0) ? A1.x / AA.x : ((AA.y != 0) ? A1.y / AA.y : A1.z / AA.z);
    double v = (AA.x != 0) ? A2.x / AA.x : ((AA.y != 0) ? A2.y / AA.y : A2.z / AA.z);
    if (u >= 0 && v >= 0 && (u + v) <= 1) {
        h.t = tt; h.ph = ph; h.nh = vmul(t->normal, outside);
        *u_out = u; *v_out = v;
        return h;
    }
    h.t = -1;
    return h;
}

/* Mesh::CalcIntersect, anyobject.h:646-666 */
static hit mesh_isect(const oscene *s, const rt355_object *o, ray e) {
    double t_min = FLT_MAX;
    hit best; memset(&best, 0, sizeof best);
    for (int i = 0; i < o->tri_count; i++) {
        const otri *t = &s->tris[o->tri_begin + i];
        double u = 0, v = 0;
        hit h = tri_isect(t, e, &u, &v);
        if (h.t >= 0 && h.t < t_min) {
            t_min = h.t;
            best = h;
            /* Triangle::GetUV, :524-526: UV_p0*(1-u-v) + UV_p1*u + UV_p2*v */
            double w = 1 - u - v;
            best.uvx = t->uv0[0] * w + t->uv1[0] * u + t->uv2[0] * v;
            best.uvy = t->uv0[1] * w + t->uv1[1] * u + t->uv2[1] * v;
        }
    }
    if (t_min > 0 && t_min != FLT_MAX) return best;
    memset(&best, 0, sizeof best);
    best.t = -1;
    return best;
}

/* CalcIntersect(eye, outside) with outside != 0, as the refraction branches pass it (main.cpp:55 with the
 * -1 .. -5 of :126,144-148).  Plane and Quadric use the value as given (anyobject.h:86,320,339), Sphere and
 * Triangle its sign (:144,486), Environment takes it for its `inside` (:199-205).  A miss returns the tuple
 * of the class's miss branch: t = -1, the point (0,0,0) and whatever normal that branch builds. */
static hit plane_isect_k(const rt355_object *o, ray e, int outside) {
    hit h; memset(&h, 0, sizeof h);
    v3 ni = from3(o->n), pi = from3(o->p);
    if (dot(ni, e.npe) >= 0) { h.t = -1; h.nh = vmul(normalize(ni), outside); return h; } /* :80-82 */
    h.t = -dot(ni, vsub(e.pe, pi)) / dot(ni, e.npe);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(ni, outside);
    return h;
}
static hit sphere_isect_k(const rt355_object *o, ray e, int outside) {
    hit h; memset(&h, 0, sizeof h);
    v3 c = from3(o->p);
    double r = o->radius;
    v3 ec = vsub(e.pe, c);
    outside = (outside > 0) ? 1 : -1;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    if (delta < 0) { h.t = -1; h.nh = vmul(V(0, 0, 0), outside); return h; }
    h.t = B - outside * sqrt(delta);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(normalize(vsub(h.ph, c)), outside);
    return h;
}
static hit environment_isect_k(const rt355_object *o, ray e, int inside) {
    hit h; memset(&h, 0, sizeof h);
    v3 c = from3(o->p);
    double r = o->radius;
    v3 ec = vsub(e.pe, c);
    if (dot(ec, ec) - r * r >= 0.00001) inside = -1;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    if (delta < 0) { h.t = -1; h.nh = vmul(V(0, 0, 0), inside); return h; }
    h.t = B + sqrt(delta);
    h.ph = vadd(e.pe, vmul(e.npe, h.t));
    h.nh = vmul(normalize(vsub(c, h.ph)), inside);
    return h;
}
static hit tri_isect_k(const otri *t, ray e, int outside, double *u_out, double *v_out) {
    hit h; memset(&h, 0, sizeof h);
    outside = (outside > 0) ? 1 : -1;
    h.t = -1;
    h.nh = vmul(normalize(t->normal), outside); /* the three miss branches, :491,500,520 */
    if (dot(t->normal, e.npe) >= 0) return h;
    double tt = -dot(t->normal, vsub(e.pe, t->p_center)) / dot(t->normal, e.npe);
    if ((outside < 0 && tt > 0) || (tt < 0)) return h;
    v3 ph = vadd(e.pe, vmul(e.npe, tt));
    v3 AA = cross(vsub(t->p1, t->p0), vsub(t->p2, t->p0));
    v3 A1 = cross(vsub(t->p2, ph), vsub(t->p0, ph));
    v3 A2 = cross(vsub(t->p0, ph), vsub(t->p1, ph));
    double u = (AA.x != 0) ? A1.x / AA.x : ((AA.y != 0) ? A1.y / AA.y : A1.z / AA.z);
    double v = (AA.x != 0) ? A2.x / AA.x : ((AA.y != 0) ? A2.y / AA.y : A2.z / AA.z);
    if (u >= 0 && v >= 0 && (u + v) <= 1) {
        h.t = tt; h.ph = ph; h.nh = vmul(t->normal, outside);
        *u_out = u; *v_out = v;
    }
    return h;
}
static hit mesh_isect_k(const oscene *s, const rt355_object *o, ray e, int outside) { /* anyobject.h:646-666 */
    double t_min = FLT_MAX;
    hit best; memset(&best, 0, sizeof best);
    for (int i = 0; i < o->tri_count; i++) {
        double u = 0, v = 0;
        hit h = tri_isect_k(&s->tris[o->tri_begin + i], e, outside, &u, &v);
        if (h.t >= 0 && h.t < t_min) { t_min = h.t; best = h; }
    }
    if (t_min > 0 && t_min != FLT_MAX) return best;
    memset(&best, 0, sizeof best); /* Normalize(Vector3D(0,0,0)) is (0,0,0), :664 */
    best.t = -1;
    return best;
}
static hit obj_isect_k(const oscene *s, int i, ray e, int k) {
    const rt355_object *o = &s->d->objects[i];
    if (o->type == RT355_OBJ_PLANE) return plane_isect_k(o, e, k);
    if (o->type == RT355_OBJ_SPHERE) return sphere_isect_k(o, e, k);
    if (o->type == RT355_OBJ_ENVIRONMENT) return environment_isect_k(o, e, k);
    if (o->type == RT355_OBJ_QUADRIC) return quadric_isect_k(o, e, k);
    return mesh_isect_k(s, o, e, k);
}

static hit obj_isect(const oscene *s, int i, ray e) {
    const rt355_object *o = &s->d->objects[i];
    if (o->type == RT355_OBJ_PLANE) return plane_isect(o, e);
    if (o->type == RT355_OBJ_SPHERE) return sphere_isect(o, e);
    if (o->type == RT355_OBJ_ENVIRONMENT) return environment_isect(o, e);
    if (o->type == RT355_OBJ_QUADRIC) return quadric_isect(o, e);
    return mesh_isect(s, o, e);
}

/* ---- texture.h ------------------------------------------------------------------------------ */
static col texel(const rt355_texture *t, int idx) {
    return C(t->rgb[3 * idx], t->rgb[3 * idx + 1], t->rgb[3 * idx + 2]);
}
/* shared tail of GetColorSphere / GetColorPlane, texture.h:92-103 / 110-121 */
static col bilinear(const rt355_texture *t, double x, double y) {
    int ix = trunc_i32(x), iy = trunc_i32(y);
    double u = x - ix, v = y - iy;
    int W = t->width, H = t->height;
    int pre00 = iy % H * W + ix % W;
    int pre01 = (iy + 1) % H * W + ix % W;
    int pre11 = (iy + 1) % H * W + (ix + 1) % W;
    int pre10 = iy % H * W + (ix + 1) % W;
    col c00 = texel(t, pre00), c01 = texel(t, pre01), c11 = texel(t, pre11), c10 = texel(t, pre10);
    col a = cmul(cmul(c00, 1 - u), 1 - v);
    col b = cmul(cmul(c01, 1 - u), v);
    col c = cmul(cmul(c11, u), v);
    col d = cmul(cmul(c10, u), 1 - v);
    return cadd(cadd(cadd(a, b), c), d);
}
static col tex_sphere(const rt355_texture *t, double theta, double phi) { /* texture.h:81-105 */
    double x = theta / (2 * REF_PI) * t->width / t->sx;
    double y = phi / REF_PI * t->height / t->sy;
    return bilinear(t, x, y);
}
static col tex_plane(const rt355_texture *t, double x, double y) {        /* texture.h:107-122 */
    return bilinear(t, x * t->width, y * t->height);
}

/* Shader::TextureMapping, shader.h:196-288 (normal map off, shader.h:23) */
static void texture_mapping(const oscene *s, const rt355_object *o, v3 ph, double uvx, double uvy,
                            col *cm0, col *cm1) {
    if (o->type == RT355_OBJ_QUADRIC) { *cm0 = C(0, 0, 0); *cm1 = C(0, 0, 0); return; } /* no branch of shader.h:198-285: Color() */
    const rt355_texture *t0 = &s->d->textures[o->tex0], *t1 = &s->d->textures[o->tex1];
    if (o->type == RT355_OBJ_SPHERE || o->type == RT355_OBJ_ENVIRONMENT) { /* shader.h:198 */
        v3 c = from3(o->p);
        double x = dot(from3(o->n0), vsub(ph, c)) / o->radius;
        double y = dot(from3(o->n1), vsub(ph, c)) / o->radius;
        double z = dot(from3(o->n2), vsub(ph, c)) / o->radius;
        double phi = REF_PI - acos(z);
        double theta = 0;
        if (z != 1 && z != -1) {
            /* `long double sss`, and acos(long double) is acosl under <cmath> (shader.h:216-222) */
            long double sss = y / sqrt(1 - z * z);
            if (sss > -1 && sss < 1) theta = (double)acosl(sss);
            else if (sss == 1) theta = 0;
            else if (sss == -1) theta = REF_PI;
            else theta = 0;
        }
        if (x < 0) theta = 2 * REF_PI - theta;
        *cm0 = tex_sphere(t0, theta, phi);
        *cm1 = tex_sphere(t1, theta, phi);
    } else if (o->type == RT355_OBJ_PLANE) {
        v3 pi = from3(o->p), n0 = from3(o->n0);
        v3 n1 = cross(n0, from3(o->n)); /* Plane ctor, anyobject.h:57 */
        double x = dot(n0, vsub(ph, pi)) / t0->sx;
        x = x - trunc_i32(x);
        double y = dot(n1, vsub(ph, pi)) / t0->sy;
        y = y - trunc_i32(y);
        x = (x < 0) ? x + 1 : x;
        y = (y < 0) ? y + 1 : y;
        *cm0 = tex_plane(t0, x, y);
        *cm1 = tex_plane(t1, x, y);
    } else {
        *cm0 = tex_plane(t0, uvx, uvy);
        *cm1 = tex_plane(t1, uvx, uvy);
    }
}

/* Shader::SolidTexture, shader.h:296-319; p00 as the Texture ctor derives it, texture.h:45 */
static void solid_texture(const oscene *s, const rt355_object *o, v3 ph, col ambient, col *cm0, col *cm1) {
    const rt355_texture *t0 = &s->d->textures[o->tex0], *t1 = &s->d->textures[o->tex1];
    v3 n0 = from3(t0->n0), n1 = from3(t0->n1), n2 = from3(t0->n2), pc = from3(t0->p_center);
    v3 p00 = vsub(vsub(pc, vmul(n0, t0->sx / 2)), vmul(n1, t0->sy / 2));
    double t = dot(vsub(ph, pc), n2);
    v3 pp = vsub(ph, vmul(n2, t));
    double x = dot(vsub(pp, p00), n0) / t0->sx;
    double y = dot(vsub(pp, p00), n1) / t0->sy;
    if (x > 0 && x < 1 && y > 0 && y < 1) {
        *cm0 = tex_plane(t0, x, y);
        *cm1 = tex_plane(t1, x, y);
    } else {
        *cm1 = ambient;
    }
}
/* Texture::Get3DFunction, texture.h:164-177, with the centre shader.h:66 passes */
static col function_3d(v3 ph) {
    v3 v = vsub(ph, V(0, 0, 20));
    v3 u = V(v.x, 0, v.z);
    double f = vlen(cross(u, V(1, 0, 0))) + dot(u, V(1, 0, 0));
    f = sin(f * REF_PI * 0.3); /* only the sign is used: any libm will do */
    double T = 0;
    if (f - T > 0) return C(250, 235, 190);
    return C(90, 190, 250);
}

/* ---- shader.h --------------------------------------------------------------------------------- */
typedef struct {
    const oscene *s;
    const rt355_shader *sh;
    uint64_t shadow_rays;
    uint64_t secondary_rays;
    const double *dirs; /* geo, main.cpp:639 */
    int n_dirs;
    uint64_t gather_rays;
    int too_deep; /* a refraction chain that does not end (see calc_subpixel_rec) */
    int soft;     /* Shader::shading calls CalcSoftShadow (shader.h:91) in place of CalcShadow (:90) */
} octx;

static col calc_diffuse(const rt355_light *L, col cm0, col cm1, v3 nh, v3 nhl) { /* shader.h:324-338 */
    double t = (dot(nh, vmul(nhl, -1)) + 1) / 2.0;
    if (L->type == 2) t = t * ((dot(from3(L->direction), nhl) + 1) / 2.0);
    v3 lc = vmul(from3(L->color), t);
    return cadd(cmulv(cm0, lc), cmulv(cm1, vsub(V(1, 1, 1), lc)));
}
static col calc_specular(const rt355_shader *sh, const rt355_light *L, col clr, v3 nh, ray eye,
                         v3 nhl) {                                                /* shader.h:340-376 */
    v3 reh = normalize(vsub(eye.npe, vmul(nh, 2 * dot(nh, eye.npe))));
    double s = dot(vmul(nhl, -1), reh);
    s = (s > 0) ? s : 0;
    s = pow(s, 25);
    if (L->type == 2) s = s * ((dot(from3(L->direction), nhl) + 1) / 2.0);
    return cadd(cmul(clr, 1 - s), cmul(C(sh->specular[0], sh->specular[1], sh->specular[2]), s));
}
static col calc_border(const rt355_shader *sh, col clr, v3 ph, v3 nh, ray eye) {   /* shader.h:378-387 */
    double b = 1 - dot(normalize(vsub(eye.pe, ph)), nh);
    double mn = 0.8, mx = 0.9;
    b = (b - mn) / (mx - mn);
    b = (b > 1) ? 1 : ((b < 0) ? 0 : b);
    return cadd(cmul(clr, 1 - b), cmul(C(sh->border[0], sh->border[1], sh->border[2]), b));
}
static col calc_shadow(octx *cx, const rt355_light *L, col clr, v3 ph, v3 nh, v3 nhl) { /* shader.h:393-433 */
    double shd = 1;
    v3 nlh = vmul(nhl, -1);
    if (dot(nh, nlh) <= 0) {
        shd = 0.5;
    } else {
        int n = cx->s->d->n_objects;
        double t_min = FLT_MAX;
        ray sr = mkview(ph, nlh);
        cx->shadow_rays++;
        for (int i = 0; i < n; i++) {
            hit h = obj_isect(cx->s, i, sr);
            if (h.t >= 0 && h.t < t_min) t_min = h.t;
        }
        double Ld = vlen(vsub(from3(L->position), ph));
        if (t_min > 0 && t_min < Ld) shd = t_min / Ld;
    }
    if (L->type == 2) shd = shd * ((dot(from3(L->direction), nhl) + 1) / 2.0);
    const rt355_shader *sh = cx->sh;
    return cadd(cmul(clr, shd), cmul(C(sh->ambient[0], sh->ambient[1], sh->ambient[2]), 1 - shd));
}

/* CalcSoftShadow, shader.h:439-483: the shadow ray aims at a point of lights[0]'s area (whichever light i is
 * being shaded, :441-443), nlh is not normalised before the facing test (:445-446; the View constructor
 * normalises the ray's direction), L is the distance to light i's position (:462). */
static col calc_soft_shadow(octx *cx, const rt355_light *L, col clr, v3 ph, v3 nh, v3 nhl, double m, double n) {
    const rt355_shader *sh = cx->sh;
    const rt355_light *L0 = &sh->lights[0];
    double shd = 1;
    v3 pl = from3(L0->position);
    pl = vadd(vadd(pl, vmul(vmul(from3(L0->n0), L0->sx), m)), vmul(vmul(from3(L0->n1), L0->sy), n)); /* :443 */
    v3 nlh = vsub(pl, ph);                                                                          /* :445 */
    if (dot(nh, nlh) <= 0) {
        shd = 0.5;
    } else {
        int no = cx->s->d->n_objects;
        double t_min = FLT_MAX;
        ray sr = mkview(ph, nlh);
        cx->shadow_rays++;
        for (int i = 0; i < no; i++) {
            hit h = obj_isect(cx->s, i, sr);
            if (h.t >= 0 && h.t < t_min) t_min = h.t;
        }
        double Ld = vlen(vsub(from3(L->position), ph));
        if (t_min > 0 && t_min < Ld) shd = t_min / Ld;
    }
    if (L->type == 2) shd = shd * ((dot(from3(L->direction), nhl) + 1) / 2.0);
    return cadd(cmul(clr, shd), cmul(C(sh->ambient[0], sh->ambient[1], sh->ambient[2]), 1 - shd));
}

/* Shader::shading, shader.h:42-96; m, n: its last two parameters, which only CalcSoftShadow reads */
static col shading(octx *cx, col cm0, col cm1, v3 ph, v3 nh, ray eye, int no_border, int obj,
                   double uvx, double uvy, double m, double n) {
    const rt355_object *o = &cx->s->d->objects[obj];
    const rt355_shader *sh = cx->sh;
    col clr_sum = C(0, 0, 0);
    if (o->texture_type == 1) texture_mapping(cx->s, o, ph, uvx, uvy, &cm0, &cm1);
    else if (o->texture_type == 2) solid_texture(cx->s, o, ph, cm1, &cm0, &cm1); /* shader.h:59-63 */
    else if (o->texture_type == 3) cm0 = function_3d(ph);                        /* :64-68; cm1 is the ambient colour already */
    if (o->type == RT355_OBJ_ENVIRONMENT) return cm0; /* shader.h:70-72 */
    for (int k = 0; k < sh->n_lights; k++) {
        const rt355_light *L = &sh->lights[k];
        v3 nhl;
        if (L->type == 0) nhl = from3(L->direction);
        else nhl = normalize(vsub(ph, from3(L->position)));
        col clr = calc_diffuse(L, cm0, cm1, nh, nhl);
        clr = calc_specular(sh, L, clr, nh, eye, nhl);
        if (!no_border) clr = calc_border(sh, clr, ph, nh, eye);
        if (cx->soft) clr = calc_soft_shadow(cx, L, clr, ph, nh, nhl, m, n);   /* shader.h:91 */
        else clr = calc_shadow(cx, L, clr, ph, nh, nhl);                       /* shader.h:90 */
        clr_sum = cadd(cmul(cdivd(clr_sum, k + 1.0), k), cdivd(clr, k + 1.0));
    }
    return clr_sum;
}

/* CalcSubPixel, main.cpp:42-158: closest hit and shading (:65-91), the reflection recursion (:93-112) with
 * REFLECTION_TIMES_THRES 3 and REFLECTION_KTOTAL_THRES 0.05 (:27-28), the refraction branches (:116-154).  On a
 * miss object_index keeps the value passed in (0 for camera and mirror rays), ph and nh stay zero, and the
 * reference goes on with that object's ks and fresnel (:94,116); so does this.
 * A call with outside <= -1 (a ray inside objs[object_index]) intersects that object alone, with `outside` as
 * given (:53-63); its shading (:88) is evaluated by the reference and then overwritten (:150), so it is left
 * out here, and with it its shadow rays.  Nothing bounds the refraction recursion (the thresholds of :103
 * guard the mirror ray only): a chain that does not end -- a glass object 0 in an open scene -- overflows the
 * reference's stack; here it sets too_deep. */
#define ORACLE_MAX_DEPTH 2000
static col calc_subpixel_rec(octx *cx, ray eye, int reflection_times, double reflection_ktotal, int outside,
                             int object_index, int depth, double soft_m, double soft_n) {
    int n = cx->s->d->n_objects;
    double t_min = FLT_MAX;
    hit best; memset(&best, 0, sizeof best);
    int idx = object_index, no_border = 1;
    col clr = C(0, 0, 0);
    if (depth > ORACLE_MAX_DEPTH) { cx->too_deep = 1; return clr; }
    if (outside <= -1 && object_index >= 0 && n > 0) {                         /* :53-63 */
        best = obj_isect_k(cx->s, object_index, eye, outside);
        t_min = best.t;
    } else {
        for (int i = 0; i < n; i++) {
            hit h = obj_isect(cx->s, i, eye);
            if (h.t >= 0 && h.t < t_min) {
                t_min = h.t; idx = i; best = h;
                no_border = (cx->s->d->objects[i].type == RT355_OBJ_PLANE || cx->s->d->objects[i].type == RT355_OBJ_MESH);
            }
        }
        if (t_min != FLT_MAX) {
            const rt355_object *o = &cx->s->d->objects[idx];
            const rt355_shader *sh = cx->sh;
            clr = shading(cx, C(o->rgb[0], o->rgb[1], o->rgb[2]),
                          C(sh->ambient[0], sh->ambient[1], sh->ambient[2]), best.ph, best.nh, eye,
                          no_border, idx, best.uvx, best.uvy, soft_m, soft_n);
        }
    }
    if (n == 0) return clr; /* the reference would read objs[0] of an empty vector here */
    const rt355_object *o = &cx->s->d->objects[idx];
    double ks = o->ks;                                                         /* :94 */
    reflection_ktotal *= ks;                                                   /* :97 */
    if (outside >= 1) reflection_times++;                                      /* :98-100 */
    col clr_reflection = clr;
    if (outside >= 1 && reflection_times <= 3 && reflection_ktotal >= 0.05) {  /* :103 */
        v3 reflect_npe = vsub(eye.npe, vmul(vmul(best.nh, 2), dot(best.nh, eye.npe))); /* :104 */
        ray reflect_view = mkview(best.ph, reflect_npe);                      /* :108 */
        cx->secondary_rays++;
        col sub = calc_subpixel_rec(cx, reflect_view, reflection_times, reflection_ktotal, 1, 0, depth + 1, 0, 0);
        clr_reflection = cadd(cmul(clr, 1 - ks), cmul(sub, ks));               /* :109 */
    }
    double fresnel = o->fresnel;                                               /* :116 */
    if (!(fresnel < 1)) return clr_reflection;                                 /* :153 */
    v3 Vv = vmul(eye.npe, -1);                                                 /* :119,129 */
    double c = dot(best.nh, Vv);
    if (outside >= 1) {
        double IOR = 1 / o->ior;                                               /* :121 */
        v3 T = vadd(vmul(Vv, -IOR), vmul(best.nh, IOR * c - sqrt((IOR * c) * (IOR * c) - IOR * IOR + 1))); /* :123 */
        T = normalize(T);                                                      /* :124 */
        cx->secondary_rays++;
        col clr_refraction = calc_subpixel_rec(cx, mkview(best.ph, T), reflection_times, reflection_ktotal, -1, idx, depth + 1, 0, 0);
        return cadd(cmul(clr_reflection, fresnel), cmul(clr_refraction, 1 - fresnel)); /* :127 */
    }
    double IOR = o->ior;                                                       /* :131 */
    double delta = (IOR * c) * (IOR * c) - IOR * IOR + 1;                      /* :133 */
    cx->secondary_rays++;
    if (delta > 0) {
        v3 T = normalize(vadd(vmul(Vv, -IOR), vmul(best.nh, IOR * c - sqrt(delta)))); /* :135-136 */
        return calc_subpixel_rec(cx, mkview(best.ph, T), reflection_times, reflection_ktotal, 1, idx, depth + 1, 0, 0); /* :138,150 */
    }
    v3 reflect_npe = vsub(eye.npe, vmul(vmul(best.nh, 2), dot(best.nh, eye.npe))); /* :142 */
    if (outside-- <= -5) outside = 1;                                          /* :144-146 */
    return calc_subpixel_rec(cx, mkview(best.ph, reflect_npe), reflection_times, reflection_ktotal, outside, idx, depth + 1, 0, 0); /* :148,150 */
}
static col calc_subpixel(octx *cx, ray eye) { return calc_subpixel_rec(cx, eye, 0, 1.0, 1, 0, 0, 0, 0); }
/* the call of main.cpp:312 for sub-sample (p, q) of pixel (x, y): see rt355x_soft_args */
static col calc_subpixel_soft(octx *cx, ray eye, uint32_t seed, int x, int y, int p, int q, int alias) {
    int32_t object_index; double m, n;
    rt355x_soft_args(seed, x, y, p, q, alias, &object_index, &m, &n);
    if (object_index >= cx->s->d->n_objects) object_index = 0; /* (the reference would index past its vector) */
    return calc_subpixel_rec(cx, eye, 0, 1.0, 1, object_index, 0, m, n);
}

/* sin and cos of the gather rotation, correctly rounded (evaluated in binary128 by libquadmath, rounded
 * once).  The reference calls its C library's sin and cos; those differ between libraries in the last bit
 * (glibc's are within 0.55 ulp, not correctly rounded), and a gather ray that starts on a sphere's surface
 * hits or misses that sphere by the sign of a rounding error, so a last-bit change of the rotation does
 * change pixels.  The path is therefore defined on the correctly rounded functions, which every
 * implementation can reproduce. */
static double sin_cr(double x) { return (double)sinq((__float128)x); }
static double cos_cr(double x) { return (double)cosq((__float128)x); }

/* Shader::shading2 (bleed == 0, shader.h:100-140) and Shader::shading3 (bleed == 1, shader.h:143-191).
 * random1 / random2 are the two `(double)rand() / RAND_MAX` draws, taken from rt355_rotation by the caller. */
static col shading_gather(octx *cx, int bleed, col cm0, col cm1, v3 ph, v3 nh, double random1, double random2) {
    int len = cx->n_dirs;
    int cnt = 0;
    double theta1 = random1 * 2 * REF_PI;
    double theta2 = random2 * 2 * REF_PI;
    col clr_tmp = C(0, 0, 0);
    v3 clr_sum = V(0, 0, 0);
    int obj_nmb = cx->s->d->n_objects;
    /* the reference evaluates these inside the loop; they do not change there */
    const double s1 = sin_cr(theta1), c1 = cos_cr(theta1), s2 = sin_cr(theta2), c2 = cos_cr(theta2);
    for (int i = 0; i < len; i++) {
        double x = cx->dirs[3 * i], y = cx->dirs[3 * i + 1], z = cx->dirs[3 * i + 2];
        v3 r = V(c1 * x - s1 * y, s1 * x + c1 * y, z);          /* shader.h:115,160 */
        x = r.x; y = r.y; z = r.z;
        r = V(x, c2 * y - s2 * z, s2 * y + c2 * z);             /* :119,164 */
        if (dot(r, nh) <= 0) continue;
        cx->gather_rays++;
        ray vw = mkview(ph, r);
        if (!bleed) {
            for (int j = 0; j < obj_nmb; j++) {
                hit h = obj_isect(cx->s, j, vw);
                if (h.t > 0) { cnt += 2; break; }
            }
            continue;
        }
        double t_min = FLT_MAX;
        for (int j = 0; j < obj_nmb; j++) {
            hit h = obj_isect(cx->s, j, vw);
            if (h.t > 0 && h.t < t_min) {
                t_min = h.t;
                const rt355_object *o = &cx->s->d->objects[j]; /* get<0>: the object's clr (anyobject.h:86,156,517) */
                clr_tmp = C(o->rgb[0], o->rgb[1], o->rgb[2]);
            }
        }
        if (t_min < 100) {
            clr_sum = V(clr_sum.x + clr_tmp.r, clr_sum.y + clr_tmp.g, clr_sum.z + clr_tmp.b);
            cnt += 2;
        }
    }
    if (bleed) cm1 = C(q8(clr_sum.x / cnt), q8(clr_sum.y / cnt), q8(clr_sum.z / cnt)); /* :187 */
    return cadd(cmul(cm0, 1 - cnt * 1.0 / len), cmul(cm1, cnt * 1.0 / len));           /* :137,188 */
}

/* CalcSubPixel2, main.cpp:163-209, as CalcPixel calls it (outside = 1, so the branch at :174 is not taken);
 * `shading` is 2 for the shading3 call of :205, 1 for shading2 in its place. */
static col calc_subpixel2(octx *cx, ray eye, int shading, uint32_t seed, int px, int py, int sample) {
    int n = cx->s->d->n_objects;
    double t_min = FLT_MAX;
    hit best; memset(&best, 0, sizeof best);
    int idx = 0;
    col clr = C(0, 0, 0);
    for (int i = 0; i < n; i++) {
        hit h = obj_isect(cx->s, i, eye);
        if (h.t >= 0 && h.t < t_min) { t_min = h.t; idx = i; best = h; }
    }
    if (t_min != FLT_MAX) {
        const rt355_object *o = &cx->s->d->objects[idx];
        const rt355_shader *sh = cx->sh;
        clr = shading_gather(cx, shading == RT355_SHADING_BLEED, C(o->rgb[0], o->rgb[1], o->rgb[2]),
                             C(sh->ambient[0], sh->ambient[1], sh->ambient[2]), best.ph, best.nh,
                             rt355x_rotation(seed, px, py, sample, 0), rt355x_rotation(seed, px, py, sample, 1));
    }
    return clr;
}

typedef struct { v3 p_eye, n0, n1, p00; double s_x, s_y; } ocam;
/* Scene::SetCamera, main.cpp:255-262 */
static ocam set_camera(const rt355_camera *c) {
    ocam o;
    v3 v_view = from3(c->v_view), v_up = from3(c->v_up);
    o.p_eye = from3(c->p_eye);
    v3 n_view = normalize(v_view);
    v3 p_center = vadd(o.p_eye, vmul(n_view, c->dist));
    v3 v0 = cross(v_view, v_up);
    o.n0 = normalize(v0);
    o.n1 = cross(o.n0, n_view);
    o.p00 = vsub(vsub(p_center, vmul(o.n0, c->s_x / 2)), vmul(o.n1, c->s_y / 2));
    o.s_x = c->s_x; o.s_y = c->s_y;
    return o;
}
/* CalcView, main.cpp:269-276 (Normalize at :272, again in the View ctor at view.h:23) */
static ray calc_view(const ocam *c, double x_per, double y_per) {
    v3 pp = vadd(vadd(c->p00, vmul(c->n0, x_per * c->s_x)), vmul(c->n1, y_per * c->s_y));
    v3 npe = normalize(vsub(pp, c->p_eye));
    return mkview(c->p_eye, npe);
}

static int oscene_init(oscene *s, const rt355_scene_desc *d) {
    s->d = d;
    s->tris = NULL;
    if (d->n_triangles > 0) {
        s->tris = (otri *)malloc(sizeof(otri) * (size_t)d->n_triangles);
        if (!s->tris) return -1;
        for (int i = 0; i < d->n_triangles; i++) tri_setup(&d->triangles[i], &s->tris[i]);
    }
    return 0;
}

/* ---- exported entry points ------------------------------------------------------------------ */

/* setPixels + CalcPixel, main.cpp:339-376, 281-334: alias == 0 is the one ray of lines 326-332, alias > 0
 * the alias x alias sub-samples of lines 289-325 with the jitter taken from rt355_jitter (see rt355.h) in
 * place of rand(); the soft-shadow parameters m, n (lines 307-308) are not computed because
 * Shader::shading does not use them (shader.h:90-91 calls CalcShadow).
 * `stride` > 1 renders only pixels with x % stride == 0 and y % stride == 0 (the rays
 * are the same as the full frame's; other pixels are left untouched) -- used to time a
 * bounded sample.  n_threads <= 0 = all cores; rows are interleaved over threads.  With opts->tile_count > 1
 * only the rows of the call's share of the row tiles are rendered (the multi-GPU partition, rt355.h). */
int rt_oracle_render_gather(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
                            const rt355_opts *opts, const double *dirs, int n_dirs, unsigned char *pixmap,
                            rt355_stats *stats, int n_threads, int stride);
int rt_oracle_render(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
                     const rt355_opts *opts, unsigned char *pixmap, rt355_stats *stats,
                     int n_threads, int stride) {
    return rt_oracle_render_gather(d, cam, sh, opts, NULL, 0, pixmap, stats, n_threads, stride);
}
/* The same with the direction set `geo` of the gather shaders (opts->shading != RT355_SHADING_LIGHTS). */
int rt_oracle_render_gather(const rt355_scene_desc *d, const rt355_camera *cam, const rt355_shader *sh,
                            const rt355_opts *opts, const double *dirs, int n_dirs, unsigned char *pixmap,
                            rt355_stats *stats, int n_threads, int stride) {
    if (!d || !cam || !sh || !opts || !pixmap) return RT355_ERR_ARG;
    if (opts->alias < 0) return RT355_ERR_ARG;
    oscene s;
    if (oscene_init(&s, d)) return RT355_ERR_NOMEM;
    ocam c = set_camera(cam);
    int W = opts->width, H = opts->height;
    int y0 = opts->row_begin, y1 = opts->row_end;
    if (y0 == 0 && y1 == 0) y1 = H;
    if (stride < 1) stride = 1;
    uint64_t prim = 0, shad = 0, sec = 0, gat = 0, deep = 0;
    const int soft = opts->shading == RT355_SHADING_SOFT;
    const int shading = soft ? 0 : opts->shading; /* != 0: CalcSubPixel2 with a gather shader */
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)n_threads;
    const int y_first = (y0 + stride - 1) / stride * stride; /* sampled rows: multiples of stride; dealt out one by one */
    const int n_sampled = y_first < y1 ? (y1 - y_first + stride - 1) / stride : 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : prim, shad, sec, gat, deep)
    for (int j = 0; j < n_sampled; j++) {
        const int y = y_first + j * stride;
        /* interleaved row tiles (rt355_opts.tile_*): only the rows of this call's share */
        if (opts->tile_count > 1 && opts->tile_rows > 0 && ((y - y0) / opts->tile_rows) % opts->tile_count != opts->tile_phase) continue;
        octx cx; cx.s = &s; cx.sh = sh; cx.shadow_rays = 0; cx.secondary_rays = 0;
        cx.dirs = dirs; cx.n_dirs = n_dirs; cx.gather_rays = 0; cx.too_deep = 0; cx.soft = soft;
        const int alias = opts->alias;
        for (int x = 0; x < W; x += stride) {
            col clr;
            if (alias > 0) {
                double red = 0, green = 0, blue = 0;
                for (int q = 0; q < alias; q++)
                    for (int p = 0; p < alias; p++) {
                        double pointX = (rt355x_jitter(opts->seed, x, y, q * alias + p, 0) + p) / alias + x; /* :292 */
                        double pointY = (rt355x_jitter(opts->seed, x, y, q * alias + p, 1) + q) / alias + y; /* :293 */
                        ray eye = calc_view(&c, pointX / W, pointY / H);
                        col sub = shading ? calc_subpixel2(&cx, eye, shading, opts->seed, x, y, q * alias + p)
                                          : (soft ? calc_subpixel_soft(&cx, eye, opts->seed, x, y, p, q, alias) : calc_subpixel(&cx, eye));
                        red += sub.r; green += sub.g; blue += sub.b;                                         /* :315-317 */
                        prim++;
                    }
                red = red / (alias * alias); green = green / (alias * alias); blue = blue / (alias * alias);  /* :322-324 */
                clr = C(q8(red), q8(green), q8(blue));                                                        /* :325 */
            } else {
                ray eye = calc_view(&c, (double)x / W, (double)y / H);
                clr = shading ? calc_subpixel2(&cx, eye, shading, opts->seed, x, y, 0) : calc_subpixel(&cx, eye);
                prim++;
            }
            size_t i = ((size_t)y * W + x) * 3;
            pixmap[i] = clr.r; pixmap[i + 1] = clr.g; pixmap[i + 2] = clr.b;
        }
        shad += cx.shadow_rays;
        sec += cx.secondary_rays;
        gat += cx.gather_rays;
        deep += cx.too_deep;
    }
    free(s.tris);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->primary_rays = prim;
        stats->shadow_rays = shad;
        stats->secondary_rays = sec;
        stats->gather_rays = gat;
    }
    if (deep) return RT355_ERR_UNSUPPORTED;
    return RT355_OK;
}

/* Known-answer probes (one call = one reference function). */

double rt_oracle_strata(uint32_t seed, int32_t x, int32_t y, int32_t which) { return rt355x_strata(seed, x, y, which); }
double rt_oracle_jitter(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t dim) {
    return rt355x_jitter(seed, x, y, sample, dim);
}
/* the C library's sin and cos, as the reference's shading2 / shading3 call them, and the correctly rounded
 * ones the gather path is defined on */
void rt_oracle_sincos(double x, double *s, double *c) { *s = sin(x); *c = cos(x); }
void rt_oracle_sincos_rounded(double x, double *s, double *c) { *s = sin_cr(x); *c = cos_cr(x); }


/* Camera ray of pixel (x,y): out[0..2] = pe, out[3..5] = npe. */
void rt_oracle_view(const rt355_camera *cam, int W, int H, int x, int y, double *out) {
    ocam c = set_camera(cam);
    ray e = calc_view(&c, (double)x / W, (double)y / H);
    out[0] = e.pe.x; out[1] = e.pe.y; out[2] = e.pe.z;
    out[3] = e.npe.x; out[4] = e.npe.y; out[5] = e.npe.z;
}

/* objs[obj]->CalcIntersect(View(pe,npe)): out = t, ph[3], nh[3], uv[2] */
int rt_oracle_intersect(const rt355_scene_desc *d, int obj, const double *pe, const double *npe,
                        double *out) {
    oscene s;
    if (oscene_init(&s, d)) return RT355_ERR_NOMEM;
    hit h = obj_isect(&s, obj, mkview(from3(pe), from3(npe)));
    out[0] = h.t; out[1] = h.ph.x; out[2] = h.ph.y; out[3] = h.ph.z;
    out[4] = h.nh.x; out[5] = h.nh.y; out[6] = h.nh.z; out[7] = h.uvx; out[8] = h.uvy;
    free(s.tris);
    return RT355_OK;
}

/* Color operators: op 0 `c*v`, 1 `c/v`, 2 `c+c2`, 3 `c*Vector3D(v,v2,v3)` */
void rt_oracle_color_op(int op, const unsigned char *c, const unsigned char *c2, double v, double v2,
                        double v3_, unsigned char *out) {
    col a = C(c[0], c[1], c[2]), r = a;
    if (op == 0) r = cmul(a, v);
    else if (op == 1) r = cdivd(a, v);
    else if (op == 2) r = cadd(a, C(c2[0], c2[1], c2[2]));
    else if (op == 3) r = cmulv(a, V(v, v2, v3_));
    out[0] = r.r; out[1] = r.g; out[2] = r.b;
}

/* Texture::GetColorPlane (kind 0) / GetColorSphere (kind 1) */
void rt_oracle_texture(const rt355_texture *t, int kind, double a, double b, unsigned char *out) {
    col r = kind ? tex_sphere(t, a, b) : tex_plane(t, a, b);
    out[0] = r.r; out[1] = r.g; out[2] = r.b;
}

/* FNV-1a 64 over a byte buffer: the frame hash quoted for reference frames. */
uint64_t rt_oracle_fnv1a64(const unsigned char *p, uint64_t n) {
    uint64_t h = 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < n; i++) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}
