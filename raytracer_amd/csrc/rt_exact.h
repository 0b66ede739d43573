// rt_exact.h -- fp64 device functions that evaluate the reference's formulas operation by
// operation (same order, same roundings; the translation unit is built with
// -ffp-contract=off so no multiply-add is fused here).  They run only on the few
// ray/primitive pairs that survive the fp32 candidate search, and on the one accepted hit
// of each ray, so their cost is O(1) per ray.
//
// Citations are to liz425/RayTracer, image_synthesis/<file>:<line>.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "rt_sincos_tables.h"

#define RT_FLT_MAX 3.4028234663852886e38 /* FLT_MAX held in a double: main.cpp:45, shader.h:401 */
#define RT_PI 3.1415927                  /* texture.h:17 (shader.h:15 and main.cpp:23 are #ifndef) */

#define RT_OBJ_PLANE 0
#define RT_OBJ_SPHERE 1
#define RT_OBJ_MESH 2
#define RT_OBJ_ENVIRONMENT 3
#define RT_OBJ_QUADRIC 4

struct d3 { double x, y, z; };

// ---- vector3D.h -------------------------------------------------------------------------
__device__ __forceinline__ d3 D3(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 vadd(d3 a, d3 b) { return D3(a.x + b.x, a.y + b.y, a.z + b.z); } // :33,118
__device__ __forceinline__ d3 vsub(d3 a, d3 b) { return D3(a.x - b.x, a.y - b.y, a.z - b.z); } // :38,113
__device__ __forceinline__ d3 vmul(d3 a, double s) { return D3(a.x * s, a.y * s, a.z * s); }   // :46
__device__ __forceinline__ d3 vdiv(d3 a, double s) { return D3(a.x / s, a.y / s, a.z / s); }   // :42
__device__ __forceinline__ double vlen(d3 a) { return sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } // :50
__device__ __forceinline__ double dot(d3 u, d3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }  // :90
__device__ __forceinline__ d3 cross(d3 u, d3 v) {                                                // :94
    return D3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
__device__ __forceinline__ d3 normalize(d3 v) { // :79-88
    double l = vlen(v);
    if (l != 0) return vdiv(v, l);
    return v;
}

// ---- color.h ------------------------------------------------------------------------------
// Color(char,char,char) (color.h:25) fed with a double: x86-64 converts with cvttsd2si to 32
// bits and keeps the low byte -- toward zero, wrap mod 256, out of range or NaN -> 0.
// v_cvt_i32_f64 saturates instead, hence the explicit range test.
// v_cvt_i32_f64 truncates toward zero, saturates out-of-range values and turns NaN into 0.  Saturated at the negative
// end it gives 0x80000000, x86's own answer for everything it cannot represent (low byte 0), so only the positive
// end needs a test: from 2^31 up (and for NaN, where the comparison fails) the byte is 0, not the 0xff of INT_MAX.
// (The instruction is written out because a C cast of an out-of-range double is undefined to the compiler.)
__device__ __forceinline__ int32_t cvt_i32_saturating(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t i;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(x));
    return i;
#else
    return (x > -2147483649.0 && x < 2147483648.0) ? (int32_t)x : (x < 0 ? INT32_MIN : (x == x ? INT32_MAX : 0));
#endif
}
__device__ __forceinline__ uint32_t q8(double x) {
    const uint32_t low = (uint32_t)cvt_i32_saturating(x) & 0xffu;
    return (x < 2147483648.0) ? low : 0u;
}
// colours travel as 0x00BBGGRR in one register
__device__ __forceinline__ uint32_t RGB(uint32_t r, uint32_t g, uint32_t b) { return r | (g << 8) | (b << 16); }
__device__ __forceinline__ uint32_t cr(uint32_t c) { return c & 0xffu; }
__device__ __forceinline__ uint32_t cg(uint32_t c) { return (c >> 8) & 0xffu; }
__device__ __forceinline__ uint32_t cb(uint32_t c) { return (c >> 16) & 0xffu; }
__device__ __forceinline__ uint32_t cmul(uint32_t c, double v) { // :33
    return RGB(q8((double)(int)cr(c) * v), q8((double)(int)cg(c) * v), q8((double)(int)cb(c) * v));
}
__device__ __forceinline__ uint32_t cmulv(uint32_t c, d3 v) { // :37
    return RGB(q8((double)(int)cr(c) * v.x), q8((double)(int)cg(c) * v.y), q8((double)(int)cb(c) * v.z));
}
__device__ __forceinline__ uint32_t cdivd(uint32_t c, double v) { // :41
    return RGB(q8((double)(int)cr(c) / v), q8((double)(int)cg(c) / v), q8((double)(int)cb(c) / v));
}
__device__ __forceinline__ uint32_t cadd(uint32_t a, uint32_t b) { // :45, each channel wraps mod 256
    return RGB((cr(a) + cr(b)) & 0xffu, (cg(a) + cg(b)) & 0xffu, (cb(a) + cb(b)) & 0xffu);
}
__device__ __forceinline__ int trunc_i32(double x) { // the reference's (int)x casts, texture.h:92-97
    const int32_t i = cvt_i32_saturating(x);
    return (x < 2147483648.0) ? i : (int)0x80000000;
}

// ---- device scene records --------------------------------------------------------------------
struct DObject { // one entry of Scene::objs, as uploaded
    double p[3], n[3], n0[3], n1[3], n2[3];
    double radius;
    int32_t type, texture_type, tex0, tex1, tri_begin, tri_count;
    uint32_t rgb; // 0x00BBGGRR
    int32_t pad_;
    double ks;    // reflection coefficient (anyobject.h:22)
    double fresnel, ior; // anyobject.h:20-21 (refraction, main.cpp:116-154)
};
struct DQuadric { // Quadric (anyobject.h:231-350): its members, and the constants its formulas divide out every call
    double pc[3], n0[3], n1[3], n2[3]; // pCenter, nCoor[0..2]
    double s[3];                       // sCoor
    double a[5];                       // aCoor (ints, held as the doubles the formulas convert them to)
    double k;                          // aCoor[3] / sCoor[2]
    double c2[3];                      // 2 * aCoor[i] / pow(sCoor[i], 2)
};
struct DTriangle { // Triangle with its ctor-derived members (anyobject.h:456-461)
    double p0[3], p1[3], p2[3];
    double normal[3], p_center[3];
    double uv0[2], uv1[2], uv2[2];
};
struct DTexture {
    uint64_t offset; // byte offset into the texel pool
    double sx, sy;
    int32_t width, height;
    double n0[3], n1[3], n2[3], p_center[3], p00[3]; // the solid texture's projector (texture.h:32-37,44-45)
};
struct DLight {
    double position[3], direction[3], color[3];
    int32_t type, pad_;
    // directional light: the shadow rays' direction Normalize(-direction) (shader.h:395, view.h:23), the same for
    // every hit, evaluated once on the host (IEEE sqrt and division, the operations the device would repeat)
    double shadow_dir[3];
    double n0[3], n1[3], sx, sy; // the light's area (light.h:19-23), read by CalcSoftShadow of lights[0] (shader.h:441-443)
};
struct DCamera { // Scene::SetCamera's results (main.cpp:255-262), computed on the host in double
    double p_eye[3], p00[3], n0[3], n1[3];
    double s_x, s_y;
};

__device__ __forceinline__ d3 ld3(const double *p) { return D3(p[0], p[1], p[2]); }

struct Ray { d3 pe, npe; };
// View(Point3D, Vector3D) normalises its direction (view.h:23)
__device__ __forceinline__ Ray mkview(d3 pe, d3 npe) { Ray r; r.pe = pe; r.npe = normalize(npe); return r; }

// CalcView, main.cpp:269-276
__device__ __forceinline__ Ray calc_view(const DCamera &c, double x_per, double y_per) {
    d3 p00 = ld3(c.p00), n0 = ld3(c.n0), n1 = ld3(c.n1), pe = ld3(c.p_eye);
    d3 pp = vadd(vadd(p00, vmul(n0, x_per * c.s_x)), vmul(n1, y_per * c.s_y));
    d3 npe = normalize(vsub(pp, pe));
    return mkview(pe, npe);
}

// ---- CalcIntersect: t only (enough to pick the closest hit) ----------------------------------
// Plane::CalcIntersect, anyobject.h:69-87
__device__ __forceinline__ double plane_t(const DObject &o, const Ray &e) {
    d3 ni = ld3(o.n), pi = ld3(o.p);
    if (dot(ni, e.npe) >= 0) return -1;
    return -dot(ni, vsub(e.pe, pi)) / dot(ni, e.npe);
}
// Sphere::CalcIntersect, anyobject.h:128-158
__device__ __forceinline__ double sphere_t(const DObject &o, const Ray &e, int *outside_out) {
    d3 c = ld3(o.p);
    double r = o.radius;
    d3 ec = vsub(e.pe, c);
    int outside = (dot(ec, ec) - r * r <= -0.00001) ? -1 : 1;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    *outside_out = outside;
    if (delta < 0) return -1;
    return B - (double)outside * sqrt(delta);
}
// the same on a (centre, radius) record, for callers that only ask whether t >= 0 and how large it is: with the
// origin outside and the centre behind it (B < 0) the root B - sqrt(delta) is negative whatever delta is, so the
// square root is not taken (any negative value stands for a miss there)
__device__ __forceinline__ double sphere_t_rec(const double4 &q, const Ray &e) {
    d3 c = D3(q.x, q.y, q.z);
    double r = q.w;
    d3 ec = vsub(e.pe, c);
    double Cc = dot(ec, ec) - r * r;
    int outside = (Cc <= -0.00001) ? -1 : 1;
    double B = dot(e.npe, vsub(c, e.pe));
    if (outside > 0 && B < 0) return -1;
    double delta = B * B - Cc;
    if (delta < 0) return -1;
    return B - (double)outside * sqrt(delta);
}
// Environment::CalcIntersect, anyobject.h:199-220, as every caller on the path reaches it: through an
// AnyObject*, so the default argument is the base class's 0 (anyobject.h:23), or with an explicit 0
// (shader.h:128,174).  `inside` is therefore 0 with the origin inside the sphere and -1 outside; t is the FAR
// root.  q = (centre, radius).
__device__ __forceinline__ double environment_t_rec(const double4 &q, const Ray &e, int *inside_out) {
    d3 c = D3(q.x, q.y, q.z);
    double r = q.w;
    d3 ec = vsub(e.pe, c);
    *inside_out = (dot(ec, ec) - r * r >= 0.00001) ? -1 : 0;
    double B = dot(e.npe, vsub(c, e.pe));
    double Cc = dot(ec, ec) - r * r;
    double delta = B * B - Cc;
    if (delta < 0) return -1;
    return B + sqrt(delta);
}
// Quadric::CalcIntersect, anyobject.h:283-345, as the path calls it (outside = 0 on entry).  pow(x, 2) is x * x
// (exact either way).  *branch: 0 a miss (t = -1), 1 a hit from the A == 0 branch, 2 a hit from the quadratic
// branch.  A hit returns whatever t the formula gives, negative ones too.
__device__ __forceinline__ double quadric_t(const DQuadric &q, const Ray &e, int *outside_out, int *branch) {
    const d3 d = vsub(e.pe, ld3(q.pc));
    const double qd[3] = {dot(ld3(q.n0), d), dot(ld3(q.n1), d), dot(ld3(q.n2), d)};
    const double rd[3] = {dot(ld3(q.n0), e.npe), dot(ld3(q.n1), e.npe), dot(ld3(q.n2), e.npe)};
    double A = 0;
    double B = q.k * rd[2];
    double C = q.k * qd[2] + q.a[4];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double u = rd[i] / q.s[i], v = qd[i] / q.s[i];
        A += q.a[i] * (u * u);
        B += q.c2[i] * rd[i] * qd[i];
        C += q.a[i] * (v * v);
    }
    *outside_out = (C < 0) ? -1 : 1; // Fp of :285-296 is C, term by term
    *branch = 0;
    if (A == 0) {
        if (B != 0) { *branch = 1; return -(C / B); }
        return -1.0;
    }
    const double delta = B * B - 4 * A * C;
    if (delta >= 0) { *branch = 2; return (-B - sqrt(delta)) / (2 * A); }
    return -1.0;
}
// The normal at ph, anyobject.h:315-319 / 331-337, before the factor `outside`.  In the A == 0 branch the
// reference's loop over the three axes starts from an uninitialised index (`for (int i; ...`, :316): it is
// taken to run from 0 like its twin at :333, which is also what g++ -O2 makes of it.
__device__ __forceinline__ d3 quadric_normal(const DQuadric &q, d3 ph) {
    d3 nh = vmul(ld3(q.n2), q.k);
    const d3 d = vsub(ph, ld3(q.pc));
    nh = vadd(nh, vmul(ld3(q.n0), q.c2[0] * dot(ld3(q.n0), d)));
    nh = vadd(nh, vmul(ld3(q.n1), q.c2[1] * dot(ld3(q.n1), d)));
    nh = vadd(nh, vmul(ld3(q.n2), q.c2[2] * dot(ld3(q.n2), d)));
    return normalize(nh);
}
// Triangle::CalcIntersect, anyobject.h:475-522
__device__ __forceinline__ double tri_t(const DTriangle &t, const Ray &e, int *outside_out, d3 *ph_out,
                                        double *u_out, double *v_out) {
    d3 normal = ld3(t.normal), pc = ld3(t.p_center);
    int outside = (dot(normal, vsub(e.pe, pc)) <= 0) ? -1 : 1;
    *outside_out = outside;
    if (dot(normal, e.npe) >= 0) return -1;
    double tt = -dot(normal, vsub(e.pe, pc)) / dot(normal, e.npe);
    if ((outside < 0 && tt > 0) || (tt < 0)) return -1;
    d3 ph = vadd(e.pe, vmul(e.npe, tt));
    d3 p0 = ld3(t.p0), p1 = ld3(t.p1), p2 = ld3(t.p2);
    d3 AA = cross(vsub(p1, p0), vsub(p2, p0));
    d3 A1 = cross(vsub(p2, ph), vsub(p0, ph));
    d3 A2 = cross(vsub(p0, ph), vsub(p1, ph));
    double u = (AA.x != 0) ? A1.x / AA.x : ((AA.y != 0) ? A1.y / AA.y : A1.z / AA.z);
    double v = (AA.x != 0) ? A2.x / AA.x : ((AA.y != 0) ? A2.y / AA.y : A2.z / AA.z);
    if (u >= 0 && v >= 0 && (u + v) <= 1) {
        *ph_out = ph; *u_out = u; *v_out = v;
        return tt;
    }
    return -1;
}

// ---- closest hit over an ordered stream of primitives -----------------------------------------
// The scene list is walked in order and a hit is taken when `t >= 0 && t < t_min`
// (main.cpp:68, shader.h:406), so the first object wins ties.  A Mesh first reduces its own
// triangles the same way and reports a miss unless its minimum is > 0 (anyobject.h:646-666).
// Primitives arrive here in scene-list order (possibly with gaps: non-candidates are misses),
// so the per-object reduction is done on the fly and committed when the object changes.
// The gather loops of shading2 / shading3 take an object's hit only when `t > 0` (shader.h:130,176):
// `positive` selects that rule for the commit of an object.
struct Closest {
    double t_best; int prim_best;
    double t_obj; int prim_obj; int cur_obj; int cur_is_mesh; int positive;
    __device__ __forceinline__ void init(int positive_only = 0) {
        t_best = RT_FLT_MAX; prim_best = -1; t_obj = RT_FLT_MAX; prim_obj = -1; cur_obj = -1; cur_is_mesh = 0;
        positive = positive_only;
    }
    __device__ __forceinline__ void commit() {
        if (cur_obj < 0) return;
        double tt = t_obj;
        if (cur_is_mesh) { if (!(t_obj > 0 && t_obj != RT_FLT_MAX)) tt = -1; }
        else if (t_obj == RT_FLT_MAX) tt = -1;
        if ((positive ? tt > 0 : tt >= 0) && tt < t_best) { t_best = tt; prim_best = prim_obj; }
        cur_obj = -1;
    }
    __device__ __forceinline__ void visit(int prim, int obj, int is_mesh, double t) {
        if (obj != cur_obj) { commit(); cur_obj = obj; cur_is_mesh = is_mesh; t_obj = RT_FLT_MAX; prim_obj = -1; }
        if (t >= 0 && t < t_obj) { t_obj = t; prim_obj = prim; }
    }
};

struct SceneView { // device pointers of an uploaded scene
    const DObject *objects;
    const DTriangle *tris;
    const int32_t *prim_obj;   // primitive -> object index
    const int32_t *prim_tri;   // primitive -> triangle index, or -1 for a plane / sphere
    const double4 *prim_rec;   // primitive -> 32-byte record for the candidate tests: a sphere's centre and
                               // radius; for a triangle or plane the first 8 bytes hold (object, triangle)
    const DTexture *textures;
    const uint8_t *texels;
    const int32_t *kind_prims; // the primitives sorted by kind: n_planes planes, n_spheres spheres, n_envs
                               // environments, n_quadrics quadrics, then the triangles of all meshes, each kind
                               // in scene-list order
    const DQuadric *quadrics;  // a quadric object's record is quadrics[object.tri_begin]
    int32_t n_objects, n_prims, n_planes, n_spheres, n_envs, n_quadrics;
    double far_origin;         // L1 norm of a ray origin beyond which the ray is tested against everything (see FAR_SCALE)
};

__device__ __forceinline__ double prim_t(const SceneView &s, int prim, const Ray &e, int *is_mesh, int *obj_out) {
    int obj = s.prim_obj[prim];
    int tri = s.prim_tri[prim];
    *obj_out = obj;
    if (tri >= 0) {
        int outside; d3 ph; double u, v;
        *is_mesh = 1;
        return tri_t(s.tris[tri], e, &outside, &ph, &u, &v);
    }
    *is_mesh = 0;
    const DObject &o = s.objects[obj];
    if (o.type == RT_OBJ_PLANE) return plane_t(o, e);
    int outside;
    if (o.type == RT_OBJ_QUADRIC) { int branch; return quadric_t(s.quadrics[o.tri_begin], e, &outside, &branch); }
    if (o.type == RT_OBJ_ENVIRONMENT) return environment_t_rec(make_double4(o.p[0], o.p[1], o.p[2], o.radius), e, &outside);
    return sphere_t(o, e, &outside);
}

// The closest hit over ALL primitives, one loop per kind of primitive instead of a dispatch per primitive.
// The ordered rule above is restated without order as the smallest (t, primitive) pair: objects are
// numbered in list order, so on equal t the smaller primitive index is the earlier object.  Planes and
// spheres are objects of their own; the triangles arrive in list order and go through Closest, which keeps
// the mesh rules.  POSITIVE: hits need t > 0 (gather rays, shader.h:130,176), else t >= 0.  ANY: stop at
// the first hit found, whichever it is (shading2 only asks whether there is one, shader.h:130-133).
// ENV = false: the caller knows that the scene holds no Environment and no quadric (their loops are compiled out).
template <bool POSITIVE, bool ANY, bool ENV = true>
__device__ __forceinline__ void closest_by_kind(const SceneView &s, const Ray &e, double &t_best, int &prim_best) {
    double bt = RT_FLT_MAX;
    int bprim = -1;
    auto consider = [&](double t, int prim) {
        if ((POSITIVE ? t > 0 : t >= 0) && (t < bt || (t == bt && prim < bprim))) { bt = t; bprim = prim; }
    };
    int k = 0;
    for (; k < s.n_planes; k++) {
        const int prim = s.kind_prims[k];
        consider(plane_t(s.objects[s.prim_obj[prim]], e), prim);
        if (ANY && bprim >= 0) break;
    }
    if (!(ANY && bprim >= 0))
        for (k = s.n_planes; k < s.n_planes + s.n_spheres; k++) {
            const int prim = s.kind_prims[k];
            consider(sphere_t_rec(s.prim_rec[prim], e), prim);
            if (ANY && bprim >= 0) break;
        }
    const int first_quadric = s.n_planes + s.n_spheres + (ENV ? s.n_envs : 0);
    const int first_tri = first_quadric + (ENV ? s.n_quadrics : 0);
    if (ENV && !(ANY && bprim >= 0))
        for (k = s.n_planes + s.n_spheres; k < first_quadric; k++) {
            const int prim = s.kind_prims[k];
            int inside;
            consider(environment_t_rec(s.prim_rec[prim], e, &inside), prim);
            if (ANY && bprim >= 0) break;
        }
    if (ENV && !(ANY && bprim >= 0))
        for (k = first_quadric; k < first_tri; k++) {
            const int prim = s.kind_prims[k];
            int outside, branch;
            consider(quadric_t(s.quadrics[s.objects[s.prim_obj[prim]].tri_begin], e, &outside, &branch), prim);
            if (ANY && bprim >= 0) break;
        }
    if (!(ANY && bprim >= 0) && first_tri < s.n_prims) {
        Closest c;
        c.init(POSITIVE ? 1 : 0);
        for (k = first_tri; k < s.n_prims; k++) {
            const int prim = s.kind_prims[k];
            int outside; d3 ph; double u, v;
            const double t = tri_t(s.tris[s.prim_tri[prim]], e, &outside, &ph, &u, &v);
            c.visit(prim, s.prim_obj[prim], 1, t);
            if (ANY && c.t_best != RT_FLT_MAX) break;
        }
        c.commit();
        if (c.prim_best >= 0) consider(c.t_best, c.prim_best);
    }
    t_best = bt; prim_best = bprim;
}

// ---- texture.h ----------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t texel(const uint8_t *px, int idx) {
    return RGB(px[3 * idx], px[3 * idx + 1], px[3 * idx + 2]);
}
// common tail of GetColorSphere / GetColorPlane, texture.h:92-103 / 110-121: each term is
// truncated twice, the three additions wrap.
__device__ __forceinline__ uint32_t bilinear(const DTexture &t, const uint8_t *pool, double x, double y) {
    const uint8_t *px = pool + t.offset;
    int ix = trunc_i32(x), iy = trunc_i32(y);
    double u = x - (double)ix, v = y - (double)iy;
    int W = t.width, H = t.height;
    int pre00 = iy % H * W + ix % W;
    int pre01 = (iy + 1) % H * W + ix % W;
    int pre11 = (iy + 1) % H * W + (ix + 1) % W;
    int pre10 = iy % H * W + (ix + 1) % W;
    uint32_t a = cmul(cmul(texel(px, pre00), 1 - u), 1 - v);
    uint32_t b = cmul(cmul(texel(px, pre01), 1 - u), v);
    uint32_t c = cmul(cmul(texel(px, pre11), u), v);
    uint32_t d = cmul(cmul(texel(px, pre10), u), 1 - v);
    return cadd(cadd(cadd(a, b), c), d);
}
__device__ __forceinline__ uint32_t tex_sphere(const DTexture &t, const uint8_t *pool, double theta, double phi) {
    double x = theta / (2 * RT_PI) * (double)t.width / t.sx; // texture.h:82
    double y = phi / RT_PI * (double)t.height / t.sy;        // texture.h:83
    return bilinear(t, pool, x, y);
}
__device__ __forceinline__ uint32_t tex_plane(const DTexture &t, const uint8_t *pool, double x, double y) {
    return bilinear(t, pool, x * (double)t.width, y * (double)t.height); // texture.h:108-109
}

// Shader::TextureMapping, shader.h:196-288 (normal map off, shader.h:23)
__device__ __forceinline__ void texture_mapping(const SceneView &s, const DObject &o, d3 ph, double uvx,
                                                double uvy, uint32_t *cm0, uint32_t *cm1) {
    if (o.type == RT_OBJ_QUADRIC) { *cm0 = 0u; *cm1 = 0u; return; } // no branch of shader.h:198-285 takes it: Color()
    const DTexture &t0 = s.textures[o.tex0];
    const DTexture &t1 = s.textures[o.tex1];
    if (o.type == RT_OBJ_SPHERE || o.type == RT_OBJ_ENVIRONMENT) { // shader.h:198
        d3 c = ld3(o.p);
        double x = dot(ld3(o.n0), vsub(ph, c)) / o.radius;
        double y = dot(ld3(o.n1), vsub(ph, c)) / o.radius;
        double z = dot(ld3(o.n2), vsub(ph, c)) / o.radius;
        double phi = RT_PI - acos(z);
        double theta = 0;
        if (z != 1 && z != -1) {
            double sss = y / sqrt(1 - z * z);
            if (sss > -1 && sss < 1) theta = acos(sss);
            else if (sss == 1) theta = 0;
            else if (sss == -1) theta = RT_PI;
            else theta = 0;
        }
        if (x < 0) theta = 2 * RT_PI - theta;
        *cm0 = tex_sphere(t0, s.texels, theta, phi);
        *cm1 = tex_sphere(t1, s.texels, theta, phi);
    } else if (o.type == RT_OBJ_PLANE) {
        d3 pi = ld3(o.p);
        double x = dot(ld3(o.n0), vsub(ph, pi)) / t0.sx;
        x = x - (double)trunc_i32(x);
        double y = dot(ld3(o.n1), vsub(ph, pi)) / t0.sy;
        y = y - (double)trunc_i32(y);
        x = (x < 0) ? x + 1 : x;
        y = (y < 0) ? y + 1 : y;
        *cm0 = tex_plane(t0, s.texels, x, y);
        *cm1 = tex_plane(t1, s.texels, x, y);
    } else {
        *cm0 = tex_plane(t0, s.texels, uvx, uvy);
        *cm1 = tex_plane(t1, s.texels, uvx, uvy);
    }
}

// Shader::SolidTexture, shader.h:296-319: the hit point projected along textures[0]'s n2 onto its plane; both
// maps are read with textures[0]'s frame, outside of it the object keeps its colour over the ambient one.
__device__ __forceinline__ void solid_texture(const SceneView &s, const DObject &o, d3 ph, uint32_t ambient,
                                              uint32_t *cm0, uint32_t *cm1) {
    const DTexture &t0 = s.textures[o.tex0];
    const d3 n2 = ld3(t0.n2);
    const double t = dot(vsub(ph, ld3(t0.p_center)), n2);
    const d3 pp = vsub(ph, vmul(n2, t));
    const double x = dot(vsub(pp, ld3(t0.p00)), ld3(t0.n0)) / t0.sx;
    const double y = dot(vsub(pp, ld3(t0.p00)), ld3(t0.n1)) / t0.sy;
    if (x > 0 && x < 1 && y > 0 && y < 1) {
        *cm0 = tex_plane(t0, s.texels, x, y);
        *cm1 = tex_plane(s.textures[o.tex1], s.texels, x, y);
    } else {
        *cm1 = ambient; // cm0 stays the colour CalcIntersect returned (shader.h:311)
    }
}
// Texture::Get3DFunction(ph, Point3D(0, 0, 20)), texture.h:164-177 as shader.h:66 calls it: bands by the sign
// of sin(0.3 pi (|v x (1,0,0)| + v . (1,0,0))), v = (ph - centre) without its y.  Only the sign of the sine is
// used, and any sine with a relative error bound has the right sign (sin is zero at no double but 0).
__device__ __noinline__ uint32_t function_3d(d3 ph) { // (out of line: rare, and its sine costs every caller registers)
    const d3 v = vsub(ph, D3(0, 0, 20));
    const d3 u = D3(v.x, 0, v.z), w = D3(1, 0, 0);
    const d3 c = D3(u.y * w.z - u.z * w.y, u.z * w.x - u.x * w.z, u.x * w.y - u.y * w.x); // vector3D.h:94-96
    double f = vlen(c) + dot(u, w);
    f = sin(f * RT_PI * 0.3);
    return (f - 0.0 > 0) ? RGB(250, 235, 190) : RGB(90, 190, 250);
}

#ifndef RT_POW25
#define RT_POW25(x) pow25(x) /* -D'RT_POW25(x)=pow(x,25.0)' selects the library power */
#endif
// ---- shader.h -------------------------------------------------------------------------------------
struct ShaderView {
    const DLight *lights;
    int32_t n_lights;
    uint32_t ambient, specular, border; // 0x00BBGGRR
};

// nhl of Shader::shading, shader.h:75-82
__device__ __forceinline__ d3 light_nhl(const DLight &L, d3 ph) {
    if (L.type == 0) return ld3(L.direction);
    return normalize(vsub(ph, ld3(L.position)));
}
__device__ __forceinline__ double spot_factor(const DLight &L, d3 nhl) { // pow(x,1), shader.h:332,370,426
    return (dot(ld3(L.direction), nhl) + 1) / 2.0;
}
// CalcDiffuse, shader.h:324-338
__device__ __forceinline__ uint32_t calc_diffuse(const DLight &L, uint32_t cm0, uint32_t cm1, d3 nh, d3 nhl) {
    double t = (dot(nh, vmul(nhl, -1)) + 1) / 2.0;
    if (L.type == 2) t = t * spot_factor(L, nhl);
    d3 lc = vmul(ld3(L.color), t);
    return cadd(cmulv(cm0, lc), cmulv(cm1, vsub(D3(1, 1, 1), lc)));
}
// x^25 for CalcSpecular's pow(s, 25) (shader.h:366), 0 <= x <= 1: five squarings and two products in
// double-double arithmetic (the rounding error of every product is recovered with an FMA and carried
// along), so the returned double is the correctly rounded power except within ~2^-50 ulp of a tie --
// the accuracy of a correctly rounded library pow at a quarter of the instructions.
struct dd2 { double h, l; };
__host__ __device__ __forceinline__ dd2 dd_mul(dd2 a, dd2 b) {
    const double p = a.h * b.h;
    double e = __builtin_fma(a.h, b.h, -p);
    e = __builtin_fma(a.h, b.l, e);
    e = __builtin_fma(a.l, b.h, e);
    dd2 r;
    r.h = p + e;
    r.l = e - (r.h - p);
    return r;
}
__device__ __forceinline__ double pow25(double x) {
    dd2 x1; x1.h = x; x1.l = 0.0;
    const dd2 x2 = dd_mul(x1, x1), x4 = dd_mul(x2, x2), x8 = dd_mul(x4, x4), x16 = dd_mul(x8, x8);
    return dd_mul(dd_mul(x16, x8), x1).h;
}
// sin and cos of the gather rotation angles (shader.h:115,119,160,164), 0 <= x < 8.  The reference calls its
// C library, whose last bit differs between libraries, and that bit shows in the frame: a gather ray that
// starts on a sphere's surface hits or misses that sphere by the sign of a rounding error.  The path is
// therefore defined on the correctly rounded functions, computed here as the argument minus the nearest
// multiple of pi/2 (Cody-Waite with a 172-bit pi/2) and both Taylor series in double-double arithmetic:
// the correctly rounded value except within ~2^-45 ulp of a tie.
__host__ __device__ __forceinline__ dd2 dd_sum(double a, double b) { // a + b exactly
    dd2 r;
    r.h = a + b;
    const double bb = r.h - a;
    r.l = (a - (r.h - bb)) + (b - bb);
    return r;
}
__host__ __device__ __forceinline__ dd2 dd_add(dd2 a, dd2 b) {
    const dd2 s = dd_sum(a.h, b.h);
    const double e = s.l + (a.l + b.l);
    dd2 r;
    r.h = s.h + e;
    r.l = e - (r.h - s.h);
    return r;
}
__host__ __device__ __forceinline__ void sincos_rounded(double x, double *sin_out, double *cos_out) {
    const int k = (int)(x * RT_TWO_OVER_PI + 0.5);
    const double kd = (double)k;
    // x - k * pi/2: the first two products and the first difference are exact
    dd2 r = dd_sum(x - kd * RT_PIO2_1, -(kd * RT_PIO2_2));
    const double p3 = kd * RT_PIO2_3;
    dd2 tail; tail.h = -p3; tail.l = -(__builtin_fma(kd, RT_PIO2_3, -p3) + kd * RT_PIO2_3T);
    r = dd_add(r, tail);
    const dd2 z = dd_mul(r, r);
    dd2 ps; ps.h = RT_SIN_COEF[RT_SIN_TERMS - 1][0]; ps.l = RT_SIN_COEF[RT_SIN_TERMS - 1][1];
#pragma unroll
    for (int n = RT_SIN_TERMS - 2; n >= 0; n--) {
        dd2 c; c.h = RT_SIN_COEF[n][0]; c.l = RT_SIN_COEF[n][1];
        ps = dd_add(dd_mul(ps, z), c);
    }
    ps = dd_mul(ps, r);
    dd2 pc; pc.h = RT_COS_COEF[RT_COS_TERMS - 1][0]; pc.l = RT_COS_COEF[RT_COS_TERMS - 1][1];
#pragma unroll
    for (int n = RT_COS_TERMS - 2; n >= 0; n--) {
        dd2 c; c.h = RT_COS_COEF[n][0]; c.l = RT_COS_COEF[n][1];
        pc = dd_add(dd_mul(pc, z), c);
    }
    const int q = k & 3;
    const double sv = (q & 1) ? pc.h : ps.h, cv = (q & 1) ? ps.h : pc.h;
    *sin_out = (q & 2) ? -sv : sv;
    *cos_out = (q == 1 || q == 2) ? -cv : cv;
}
// reh of CalcSpecular (shader.h:352): the view direction mirrored at the surface; the same for every light
__device__ __forceinline__ d3 mirror_dir(d3 nh, d3 npe) {
    return normalize(vsub(npe, vmul(nh, 2 * dot(nh, npe))));
}
// CalcSpecular, shader.h:340-376 (SpecularType 0).  SKIP: without a highlight (s <= 0) return clr as it is -- 0^25 = 0,
// and clr * 1 + specular * 0 is clr itself, channel by channel; a spot light's factor could still turn the 0 into a
// NaN, so that case takes the long way.  (The one-kernel renderers save a tenth of their time with it; in the
// shading kernel of the searched path, which runs at its register limit, the extra branch costs more than it saves.)
template <bool SKIP = false>
__device__ __forceinline__ uint32_t calc_specular(const ShaderView &sh, const DLight &L, uint32_t clr, d3 reh,
                                                  d3 nhl) {
    double s = dot(vmul(nhl, -1), reh);
    if (SKIP && !(s > 0) && L.type != 2) return clr;
    s = (s > 0) ? s : 0;
    s = RT_POW25(s);
    if (L.type == 2) s = s * spot_factor(L, nhl);
    return cadd(cmul(clr, 1 - s), cmul(sh.specular, s));
}
// b of CalcBorder (shader.h:378-385): the same for every light
__device__ __forceinline__ double border_weight(d3 pe, d3 ph, d3 nh) {
    double b = 1 - dot(normalize(vsub(pe, ph)), nh);
    double mn = 0.8, mx = 0.9;
    b = (b - mn) / (mx - mn);
    return (b > 1) ? 1 : ((b < 0) ? 0 : b);
}
// CalcBorder's blend, shader.h:386
__device__ __forceinline__ uint32_t calc_border(const ShaderView &sh, uint32_t clr, double b) {
    return cadd(cmul(clr, 1 - b), cmul(sh.border, b));
}
// tail of CalcShadow, shader.h:412-432, given the closest occluder distance (SKIP: a lit point keeps its colour,
// clr * 1 + ambient * 0 is clr itself)
template <bool SKIP = false>
__device__ __forceinline__ uint32_t shadow_mix(const ShaderView &sh, const DLight &L, uint32_t clr, d3 nhl,
                                               double shd) {
    if (L.type == 2) shd = shd * spot_factor(L, nhl);
    else if (SKIP && shd == 1.0) return clr;
    return cadd(cmul(clr, shd), cmul(sh.ambient, 1 - shd));
}
// running mean over lights, shader.h:93: clr_sum / (k+1) * k + clr / (k+1) on Color, i.e. per channel
// trunc(c / (k+1)) with c an integer in 0..255 -- the correctly rounded double quotient truncates to the
// integer quotient (a non-integer c/(k+1) is at least 1/(k+1) away from the next integer) -- then an exact
// integer product taken mod 256.  The integer quotient is (c * M) >> 16 with M = 65536 / d + 1 for d <= 256.
__device__ __forceinline__ uint32_t light_mean(uint32_t clr_sum, uint32_t clr, int k) {
    const uint32_t d = (uint32_t)k + 1u;
    const uint32_t M = (d <= 256u) ? (65536u / d + 1u) : 0u;
    const uint32_t uk = (uint32_t)k;
    const uint32_t r = (((cr(clr_sum) * M) >> 16) * uk + ((cr(clr) * M) >> 16)) & 0xffu;
    const uint32_t g = (((cg(clr_sum) * M) >> 16) * uk + ((cg(clr) * M) >> 16)) & 0xffu;
    const uint32_t bl = (((cb(clr_sum) * M) >> 16) * uk + ((cb(clr) * M) >> 16)) & 0xffu;
    return RGB(r, g, bl);
}
