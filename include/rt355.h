/*
 * rt355.h -- C ABI of the MI355X (gfx950) renderer for the per-pixel path of
 * liz425/RayTracer.
 *
 * The reference has no FFI of its own: its render path is one C++ call,
 *     void setPixels(Scene sce, Shader shd, int alias)      image_synthesis/main.cpp:339-376
 * which fills the global `unsigned char *pixmap` (main.cpp:34).  This header is
 * the boundary a maintainer of the reference binds in its place: plain C, plain
 * pointers and sizes, no C++ or torch types.
 *
 * All geometry is `double`, exactly the values the reference objects hold after
 * their constructors ran.  Colours are 3 x unsigned char (color.h:16-25).
 *
 * Threading: one render at a time per scene handle.  Errors: every entry point
 * returns 0 on success or a negative RT355_ERR_* code; rt355_last_error() gives a
 * thread-local message.  Nothing throws across this boundary.
 */
#ifndef RT355_H
#define RT355_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ---------------------------------------------------------------- */
#define RT355_OK               0
#define RT355_ERR_ARG         -1 /* null pointer, bad size, inconsistent indices          */
#define RT355_ERR_UNSUPPORTED -2 /* feature of the reference outside the accelerated path */
#define RT355_ERR_DEVICE      -3 /* HIP runtime error (message holds hipGetErrorString)   */
#define RT355_ERR_NOMEM       -4

/* ---- scene objects, in scene-list order ------------------------------------------ */
/* Object kinds.  Values follow AnyObject::GetObjectType() strings
 * ("Plane" anyobject.h:89, "Sphere" :161, "Mesh" :672). */
#define RT355_OBJ_PLANE  0
#define RT355_OBJ_SPHERE 1
#define RT355_OBJ_MESH   2
#define RT355_OBJ_ENVIRONMENT 3 /* "Environment" anyobject.h:223: a sphere seen from inside whose (textured) colour
                                   Shader::shading returns unlit (shader.h:70-72); fields as for a Sphere */
#define RT355_OBJ_QUADRIC 4     /* Quadric and its subclasses Cone, Ellipsoid, Cylinder, EllipticParaboloid,
                                   HyperbolicParaboloid (anyobject.h:231-442), which differ in aCoor alone */

/* One entry of Scene::objs (main.cpp:229).  Order is significant: the closest-hit loop
 * keeps the first object on equal t (main.cpp:68). */
typedef struct rt355_object {
    double p[3];      /* Plane::pi (anyobject.h:43) | Sphere / Environment::pCenter (:101)   */
    double n[3];      /* Plane::ni (:46), used as given (not normalised, :86)               */
    double n0[3];     /* Plane::n0 (:44) | Sphere::n0 (:103); texture axes                  */
    double n1[3];     /* Sphere::n1 (:104).  For a plane the library derives
                         n1 = n0 x ni as the Plane ctor does (:57); this field is ignored. */
    double n2[3];     /* Sphere::n2 (:105)                                                   */
    double radius;    /* Sphere::radius (:102)                                               */
    double ks;        /* AnyObject::ks (:22): reflection coefficient (main.cpp:93-112)                       */
    double fresnel;   /* AnyObject::fresnel (:21): < 1 refracts (main.cpp:116-154)            */
    double ior;       /* AnyObject::IOR (:20), read when fresnel < 1                          */
    int32_t type;     /* RT355_OBJ_*                                                         */
    int32_t texture_type; /* AnyObject::texture_type (:29): 0 flat colour, 1 texture mapping (shader.h:53-58),
                             2 solid texture: textures[0] and [1] projected along textures[0]'s n2
                             (shader.h:59-63,296-319), 3 the 3D function of texture.h:164-177 (shader.h:64-68) */
    int32_t tex0;     /* index of textures[0] (lit colour map)  or -1                        */
    int32_t tex1;     /* index of textures[1] (dark colour map) or -1                        */
    int32_t tri_begin;/* Mesh: first triangle in rt355_scene_desc::triangles                 */
    int32_t tri_count;/* Mesh: number of triangles (Mesh::triangles, :541)                   */
    uint8_t rgb[3];   /* the object's own `clr` (Plane :42, Sphere :100, Quadric :240, Mesh :536) */
    uint8_t pad_[5];
    /* Quadric only (anyobject.h:233-239): p = pCenter, n0 / n1 / n2 = nCoor[0..2] as SetN stored them */
    double s[3];      /* Quadric::sCoor (SetS takes floats, :265: pass the stored doubles)  */
    int32_t a[5];     /* Quadric::aCoor = a02, a12, a22, a21, a00 (:234-235)                  */
    int32_t pad2_;
} rt355_object;

/* One Triangle of a Mesh (anyobject.h:444-532).  The library derives `normal` and
 * `p_center` exactly as the Triangle ctor does (:456-461). */
typedef struct rt355_triangle {
    double p0[3], p1[3], p2[3];
    double uv0[2], uv1[2], uv2[2]; /* Triangle::UV_p0..2 (:451) */
} rt355_triangle;

/* One Texture (texture.h:23-79) after loading: `rgb` is width*height*3 bytes in the
 * order the reference keeps them in memory (BMP row order, bottom row first, R,G,B
 * after the swap at texture.h:73-78).  The bytes are copied by rt355_scene_create. */
typedef struct rt355_texture {
    const uint8_t *rgb;
    double sx, sy;      /* Texture::sx, sy (texture.h:28-29) */
    int32_t width, height;
    /* The projector's frame, read by texture_type 2 only (texture.h:32-35): AnyObject::AddTexture leaves
     * it at n0 = (1,0,0), n1 = (0,1,0), n2 = (0,0,1), p_center = (0,0,0).  The library derives p00 as
     * the Texture ctor does (texture.h:45). */
    double n0[3], n1[3], n2[3], p_center[3];
} rt355_texture;

typedef struct rt355_scene_desc {
    const rt355_object   *objects;   int32_t n_objects;   int32_t pad0_;
    const rt355_triangle *triangles; int32_t n_triangles; int32_t pad1_;
    const rt355_texture  *textures;  int32_t n_textures;  int32_t pad2_;
} rt355_scene_desc;

/* ---- camera, lights, shader -------------------------------------------------------- */
/* The public camera fields of Scene (main.cpp:217-222).  The library derives
 * n_view, p_center, n0, n1, p00 as Scene::SetCamera does (main.cpp:255-262). */
typedef struct rt355_camera {
    double v_up[3], v_view[3], p_eye[3];
    double dist, s_x, s_y;
} rt355_camera;

/* light_src (light.h:12-56).  `direction` is the stored, already normalised member
 * (light.h:30); the library does not normalise it again.  Types 0 directional, 1 point,
 * 2 spot (soft spot, shader.h:332,370,426), 3 area (a point light to Shader::shading and
 * CalcShadow, shader.h:80-82; its area is what CalcSoftShadow aims at). */
typedef struct rt355_light {
    double position[3];
    double direction[3];
    double color[3];
    int32_t type;
    int32_t pad_;
    /* The area of an area light (light.h:19-23,44-54), read by RT355_SHADING_SOFT only, and there of lights[0]
     * alone (shader.h:441-443).  The constructors of the other types leave n0 = (1,0,0), n1 = (0,1,0),
     * sx = sy = 1 (light.h:19-23): pass those for them. */
    double n0[3], n1[3];
    double sx, sy;
} rt355_light;

/* Shader (shader.h:25-37): the private light list plus the three public colours. */
typedef struct rt355_shader {
    const rt355_light *lights;
    int32_t n_lights;
    uint8_t ambient[3];   /* Shader::ambient_color  */
    uint8_t specular[3];  /* Shader::specular_color */
    uint8_t border[3];    /* Shader::border_color   */
    uint8_t pad_[3];
} rt355_shader;

/* ---- render options ------------------------------------------------------------------ */
/* All three modes produce the same frame, byte for byte; they differ in how the ray/primitive
 * pairs that the fp64 reference formulas must evaluate are found.  (The searches bound every primitive by
 * where the reference's own test can report a hit: a triangle whose inside test is rounding noise -- its
 * deciding cross-product component, anyobject.h:512-513, vanishing beside the others -- is taken or left by the
 * reference anywhere on its plane, and is therefore a candidate of every ray, like a plane.) */
#define RT355_MODE_ALLPAIRS 0 /* fp32 conservative test of EVERY ray/primitive pair (FP32-VALU bound) */
#define RT355_MODE_BRUTE    1 /* fp64 formulas on every pair, no search (debug / tiny scenes)         */
#define RT355_MODE_PACKET   2 /* as ALLPAIRS, but a workgroup first culls the primitive table against
                                 the beam of its 1024 neighbouring rays; the default of the hosts     */

/* Which of the reference's shaders colours a hit.  HEAD of the reference calls CalcSubPixel2 with
 * shading3 (main.cpp:205,313,331); CalcSubPixel with Shader::shading is the variant commented out beside
 * it (main.cpp:312,330).  The two gather shaders cast one ray per direction of the scene's direction set
 * (rt355_scene_set_directions) from every hit and use neither lights, textures nor reflection. */
#define RT355_SHADING_LIGHTS    0 /* CalcSubPixel + Shader::shading (main.cpp:42-158, shader.h:42-96)           */
#define RT355_SHADING_OCCLUSION 1 /* CalcSubPixel2 (main.cpp:163-209) + Shader::shading2, ambient occlusion
                                     (shader.h:100-140)                                                       */
#define RT355_SHADING_BLEED     2 /* CalcSubPixel2 + Shader::shading3, colour bleeding (shader.h:143-191)       */
#define RT355_SHADING_SOFT      3 /* CalcSubPixel + Shader::shading with the line commented out at shader.h:91
                                     in place of the one at :90: CalcSoftShadow (shader.h:439-483), whose shadow
                                     ray aims at a point of lights[0]'s area chosen per sub-sample by the strata
                                     of main.cpp:286-288,307-308 (see rt355x_soft_args below)                  */

typedef struct rt355_opts {
    int32_t width, height; /* the reference's globals (main.cpp:33)                         */
    int32_t alias;         /* setPixels' third argument: 0 = one ray through the pixel corner
                              (main.cpp:326-331), a > 0 = a x a jittered sub-samples per pixel,
                              averaged (main.cpp:289-325), jitter from rt355_jitter below      */
    int32_t row_begin;     /* slab [row_begin,row_end) of rows to render; 0,0 = whole frame */
    int32_t row_end;
    int32_t mode;          /* RT355_MODE_*                                                  */
    uint32_t seed;         /* seed of the jitter (alias > 0) and of the gather rotation     */
    int32_t shading;       /* RT355_SHADING_*                                               */
    /* Interleaved row tiles, the partition for frames whose cost is uneven from top to bottom (sky above,
     * geometry below): the slab [row_begin,row_end) is cut into tiles of tile_rows rows (the last may be
     * shorter), tile k belongs to share k % tile_count, and the call renders the tiles of share tile_phase
     * only, in place (the other rows of the pixmap are not touched).  tile_count <= 1: the whole slab. */
    int32_t tile_rows;     /* rows per tile (> 0 when tile_count > 1)                        */
    int32_t tile_count;    /* number of shares the tiles are dealt to                       */
    int32_t tile_phase;    /* the share this call renders, 0 <= tile_phase < tile_count      */
    int32_t pad_;
} rt355_opts;

/* The jitter of CalcPixel.  The reference draws `(double)rand() / RAND_MAX` twice per sub-sample
 * (main.cpp:292-293) from libc's serial generator, in pixel order; that stream cannot be drawn in
 * parallel and differs between C libraries.  rt355 keeps the value set (a 31-bit integer divided by
 * RAND_MAX = 2147483647, so 0 <= jitter <= 1) and takes the integer from a counter-based generator
 * instead (four rounds of the SplitMix64 finaliser over seed, x, y and 2 * sample + dim), so every
 * sub-sample's jitter depends only on where it is.  sample = q * alias + p, dim 0 = x, 1 = y.
 * Defined inline (rt355x_jitter) so that the library's device code, the CPU checker and the reference
 * harness share one definition; the library also exports it as rt355_jitter for hosts in other languages. */
#if defined(__HIPCC__)
#define RT355_INLINE static __host__ __device__ inline
#else
#define RT355_INLINE static inline
#endif
RT355_INLINE uint64_t rt355x_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
RT355_INLINE double rt355x_jitter(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t dim) {
    uint64_t z = rt355x_mix64((uint64_t)seed);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)x);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)y);
    z = rt355x_mix64(z ^ (((uint64_t)(uint32_t)sample << 1) | (uint64_t)(dim & 1)));
    return (double)(z >> 33) / 2147483647.0;
}

/* The random rotation of the gather shaders.  shading2 / shading3 draw `(double)rand() / RAND_MAX` twice
 * per shaded hit (shader.h:106-107,149-150) and turn the direction set by theta = value * 2 * PI about z
 * and then about x; like the jitter, the value comes from the counter-based generator instead, keyed by the
 * sub-sample the hit belongs to.  axis 0 = random1, 1 = random2. */
RT355_INLINE double rt355x_rotation(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t axis) {
    uint64_t z = rt355x_mix64((uint64_t)seed);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)x);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)y);
    z = rt355x_mix64(z ^ (((uint64_t)(uint32_t)sample << 1) | (uint64_t)(axis & 1)));
    z = rt355x_mix64(z ^ 0x726f74617465ull);
    return (double)(z >> 33) / 2147483647.0;
}

/* The soft-shadow strata of CalcPixel.  Per pixel the reference draws k0, k1 = (int)(rand() / RAND_MAX * 100)
 * and r = rand() / RAND_MAX (main.cpp:286-288) and gives sub-sample (p, q) the parameters
 * m = ((p + k0) % alias + r) / alias, n = ((q + k1) % alias + r) / alias (main.cpp:307-308).  As with the jitter,
 * the three draws come from the counter-based generator instead (which = 0, 1, 2 for k0, k1, r). */
RT355_INLINE double rt355x_strata(uint32_t seed, int32_t x, int32_t y, int32_t which) {
    uint64_t z = rt355x_mix64((uint64_t)seed);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)x);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)y);
    z = rt355x_mix64(z ^ (uint64_t)(uint32_t)which);
    z = rt355x_mix64(z ^ 0x737472617461ull);
    return (double)(z >> 33) / 2147483647.0;
}
/* What CalcSubPixel receives for sub-sample (p, q) of pixel (x, y) from the call the reference keeps beside the
 * one it makes, `CalcSubPixel(eye, sce.objs, shd, 0, 1.0, 1, m, n)` (main.cpp:312), given the signature
 * `(..., int outside, int object_index = 0, double m = 0, double n = 0)` (main.cpp:42): the seventh argument
 * lands in object_index, so object_index = (int)m, the function's m is the caller's n, and its n stays 0.
 * alias == 0 is the call of main.cpp:330, which passes neither: all three are 0.  Rays cast by the recursion
 * (main.cpp:109,126,138,148) pass neither as well. */
RT355_INLINE void rt355x_soft_args(uint32_t seed, int32_t x, int32_t y, int32_t p, int32_t q, int32_t alias,
                                   int32_t *object_index, double *m, double *n) {
    *object_index = 0; *m = 0; *n = 0;
    if (alias <= 0) return;
    const int32_t k0 = (int32_t)(rt355x_strata(seed, x, y, 0) * 100), k1 = (int32_t)(rt355x_strata(seed, x, y, 1) * 100);
    const double r = rt355x_strata(seed, x, y, 2);
    const double caller_m = ((p + k0) % alias + r) * 1.0 / alias;   /* main.cpp:307 */
    const double caller_n = ((q + k1) % alias + r) * 1.0 / alias;   /* main.cpp:308 */
    *object_index = (int32_t)caller_m;                              /* 0, or 1 when r == 1 in the last stratum */
    *m = caller_n;
}

typedef struct rt355_stats {
    uint64_t primary_rays;   /* camera rays cast                                            */
    uint64_t shadow_rays;    /* shadow rays cast: (hit, light) pairs with N.L > 0           */
    uint64_t pair_tests;     /* RT355_MODE_ALLPAIRS: ray/table-row pairs the search tested   */
    uint64_t exact_tests;    /* pairs evaluated by the fp64 reference formulas              */
    uint64_t overflow_rays;  /* rays whose candidate list overflowed (re-done brute force)  */
    double   ms_device;      /* HIP-event time of the device work of this call              */
    uint64_t packet_survivors; /* RT355_MODE_PACKET: (wave, primitive) pairs that passed both beam tests */
    uint64_t primary_hits;   /* camera rays that hit something                              */
    uint64_t secondary_rays; /* reflection rays cast (main.cpp:97-108)                      */
    uint64_t gather_rays;    /* gather rays cast: directions with ray . nh > 0 (shader.h:122,167) */
} rt355_stats;

typedef struct rt355_scene rt355_scene; /* opaque, library-owned; bound to one HIP device (rt355_scene_create) or
                                           dealing its calls to one scene per device (rt355_scene_create_multi) */

/* Flatten and upload a scene to the current HIP device.
 * Replaces: the object-list walk done implicitly by every CalcIntersect call through
 * vector<AnyObject*> (main.cpp:65-79); the reference has no upload step. */
int rt355_scene_create(const rt355_scene_desc *desc, rt355_scene **out);
void rt355_scene_destroy(rt355_scene *scene);

/* The same scene on several GPUs of this process: flattened and uploaded once to each of devices[0 .. n_devices - 1]
 * (HIP device ordinals; a device may be named more than once), for a caller that is ONE process calling setPixels
 * once per frame, as the reference's main() is (main.cpp:640).  The handle is used like any other:
 *   rt355_render        deals the 32-row tiles of the slab to the devices (tile k to device k mod n), every GPU
 *                       writing its rows over its own link into the caller's one pixmap -- the kernels' own stores
 *                       where the pixmap is page-locked (rt355_host_alloc, rt355_host_register), each device's
 *                       staging memory and a host copy where it is not; opts->tile_count must be left at 0.
 *                       With the option "gather" set to 1 (rt355_scene_set_option) the shares are rendered into
 *                       HBM, copied to devices[0] over the GPUs' own links and read back from there in one transfer.
 *                       With the option "balance" set to 1 the slab is dealt in n contiguous runs of rows instead, whose
 *                       cuts follow the times the devices' shares took in the calls before (a renderer of successive
 *                       frames of one scene; the first call cuts the slab evenly);
 *   rt355_render_sweep  deals whole frames, frame f to device f mod n;
 *   rt355_scene_set_directions, rt355_scene_set_option reach every device; statistics are summed over the devices
 *                       (ms_device: the longest); the kernel-time queries and rt355_probe_intersect answer for
 *                       devices[0]; rt355_render_device(_noalloc) return RT355_ERR_UNSUPPORTED (a device pixmap
 *                       lies on one GPU).
 * Pixels are independent given the replicated scene, so the frame is the one a single device renders, byte for byte.
 * Replaces: setPixels(Scene, Shader, int alias) (main.cpp:339-376), as rt355_scene_create + rt355_render do. */
int rt355_scene_create_multi(const rt355_scene_desc *desc, const int32_t *devices, int32_t n_devices, rt355_scene **out);

/* Render rows [row_begin,row_end) into a caller-owned HOST pixmap of width*height*3
 * bytes, RGB, row 0 = bottom scanline, index (y*width+x)*3.
 * Replaces: setPixels(Scene, Shader, int alias) writing the global pixmap
 * (main.cpp:339-376, called at main.cpp:640). */
int rt355_render(rt355_scene *scene, const rt355_camera *cam, const rt355_shader *shader,
                 const rt355_opts *opts, unsigned char *pixmap, rt355_stats *stats);

/* Same, but the framebuffer is a DEVICE pointer (width*height*3 bytes) and the work is
 * enqueued on `hip_stream` (a hipStream_t, NULL = default stream).  Returns after
 * enqueueing; call rt355_scene_sync before reading `stats` or the pixels.
 * No counterpart in the reference (it has no device memory); this is the entry point
 * a GPU-resident pipeline and the benchmark use. */
int rt355_render_device(rt355_scene *scene, const rt355_camera *cam,
                        const rt355_shader *shader, const rt355_opts *opts,
                        void *d_pixmap, void *hip_stream, rt355_stats *stats);

/* rt355_render_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc,
 * processes whose device memory is spoken for).  No counterpart in the reference.  Not for use while the stream is
 * being captured into a HIP graph: like rt355_render_device it records timing events and enqueues the copy of its
 * statistics, which the next call on the handle waits for.  It calls no hipMalloc / hipFree /
 * hipHostMalloc or other allocating routine: the device memory the kernels need beyond the scene, the
 * lights and the pixmap -- the statistics counters, the device copy of the light list and, with alias > 0,
 * the sub-sample colours -- lives in an array inside the library's code object (RT355_NOALLOC_BYTES, about
 * 16 MiB per GPU, held until the process ends).  Limits (RT355_ERR_UNSUPPORTED beyond them, with a
 * message): only frames that are rendered without a search -- scenes of fewer than 64 primitives, or
 * mode RT355_MODE_BRUTE -- at most RT355_NOALLOC_MAX_LIGHTS lights and, with alias > 0, at most
 * RT355_NOALLOC_MAX_SUBSAMPLES sub-samples per call (split the frame into row slabs to stay below).
 * Every call on a GPU uses that GPU's one array: calls on one GPU that overlap in time (two streams, two
 * host threads, two scene handles) race; calls that follow one another on a stream do not.  Pixels,
 * statistics and error behaviour are those of rt355_render_device. */
#define RT355_NOALLOC_MAX_LIGHTS 64
#define RT355_NOALLOC_MAX_SUBSAMPLES (1 << 22)
int rt355_render_device_noalloc(rt355_scene *scene, const rt355_camera *cam,
                                const rt355_shader *shader, const rt355_opts *opts,
                                void *d_pixmap, void *hip_stream, rt355_stats *stats);

/* Page-locked host memory for the pixmap handed to rt355_render (in place of `new unsigned char[...]`,
 * main.cpp:544): the readback is then one DMA transfer straight into the caller's buffer.  rt355_render
 * accepts any host pointer; one that is not page-locked is filled through a staging buffer at the cost
 * of a host copy.  No counterpart in the reference. */
void *rt355_host_alloc(uint64_t bytes);
void rt355_host_free(void *p);
/* Page-lock memory the caller already has -- the `new unsigned char[width * height * 3]` of main.cpp:544 left as it
 * is, or a pixmap in memory that several processes share (one rank per GPU, every rank's rt355_render writing its
 * rows of the one frame) -- for the current device; rt355_render then treats it like memory of rt355_host_alloc.
 * Unregister before the memory is freed.  No counterpart in the reference. */
int rt355_host_register(void *p, uint64_t bytes);
int rt355_host_unregister(void *p);
/* The address at which the device reaches a buffer of rt355_host_alloc (NULL for any other pointer): passed to
 * rt355_render_device as d_pixmap it makes the frame arrive in host memory without a copy and without the call
 * waiting for it -- several scene handles, each on its own stream, then keep several frames of a camera sweep in
 * flight, and rt355_scene_sync(handle) tells when a handle's frame is complete. */
void *rt355_host_device_pointer(void *p);

/* A camera sweep: frame i is rendered with cams[i] into pixmaps[i] (host memory, each as for rt355_render), all with
 * the same shader and options; stats, if not NULL, receives n_frames records.  The same pixels as n_frames calls of
 * rt355_render, but the transfer of a frame to the host runs while the next frame is rendered.
 * Replaces: setPixels (main.cpp:339-376) called once per camera -- the call at main.cpp:640 inside a loop that moves
 * Scene's camera fields (main.cpp:529-541) between frames; the reference itself renders one frame per run. */
int rt355_render_sweep(rt355_scene *scene, const rt355_camera *cams, int32_t n_frames, const rt355_shader *shader,
                       const rt355_opts *opts, unsigned char *const *pixmaps, rt355_stats *stats);

/* Wait for the scene's outstanding render and fill the stats passed to it. */
int rt355_scene_sync(rt355_scene *scene);

/* Summed HIP-event duration (ms) of the candidate-search kernel launches of the last synced
 * render, and their count.  Used by the benchmark's roofline figure. */
int rt355_scene_search_kernel_ms(rt355_scene *scene, double *ms_sum, int32_t *launches);
/* Same for the fp64 kernels: the primary resolve (closest hit, hit record, first shadow rays), then
 * one shading launch per group of lights. */
int rt355_scene_resolve_kernel_ms(rt355_scene *scene, double *ms_sum, int32_t *launches);
/* HIP-event duration (ms) of the index-th of those launches (0 = primary resolve, 1.. = shading
 * of light group index - 1). */
int rt355_scene_resolve_launch_ms(rt355_scene *scene, int32_t index, double *ms);

/* Debug probe: objs[obj]->CalcIntersect(View(pe,npe)) for n host rays (n x 6 doubles: pe, npe)
 * evaluated by the device's fp64 formulas; out = n x 9 doubles (t, ph[3], nh[3], uv[2]; t = -1
 * and zeros on a miss).  Mirrors AnyObject::CalcIntersect (anyobject.h:23) one call at a time so
 * the device arithmetic can be checked in isolation. */
int rt355_probe_intersect(rt355_scene *scene, int obj, const double *rays, int n, double *out);
/* Debug probe: rt355_sincos evaluated on the device for n arguments -> n x 2 doubles (sin, cos). */
int rt355_probe_sincos(const double *x, int n, double *out);

/* HIP-event duration (ms) of the index-th candidate-search launch of the last synced render
 * (0 = camera rays; then, per group of lights, one launch for the lights with a position and one
 * for the directional lights, each covering all such lights of the group). */
int rt355_scene_search_launch_ms(rt355_scene *scene, int32_t index, double *ms);

const char *rt355_last_error(void);
int rt355_abi_version(void);

/* rt355x_jitter, rt355x_rotation and rt355x_strata as exported functions. */
double rt355_jitter(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t dim);
double rt355_rotation(uint32_t seed, int32_t x, int32_t y, int32_t sample, int32_t axis);
double rt355_strata(uint32_t seed, int32_t x, int32_t y, int32_t which);

/* Tuning options of a scene handle, by name.  No counterpart in the reference.  Their initial values are read from the
 * environment once, when the handle is created (RT355_<NAME IN CAPITALS>, e.g. RT355_TAIL_PARTS=1); no render call
 * reads the environment.  None of them changes a frame's bytes.
 *   "readback"        rt355_render into a page-locked pixmap: 0 = the handle times both ways on its first frames of
 *                     a kind and keeps the faster (default), 1 = the kernels write the pixmap directly over the link,
 *                     2 = rendered into HBM and copied in parts
 *   "render_parts"    parts a one-kernel frame is rendered and copied in (default 6)
 *   "tail_parts"      parts the last stage of a searched frame is shaded and copied in (default 4); "no_tail_parts" 1
 *                     = such a frame is enqueued and copied whole
 *   "packet_rays"     rays per lane of the packet search, 1 or 4 (default 0: both are timed on the first frames)
 *   "heavy_min_prims" the packet search runs as two launches from this many primitives (default 300000; 0 = always),
 *   "heavy_after"     a wave handing the rest of its walk over after this many primitives (default 256)
 *   "heavy_workers"   to this many waves (default 16)
 *   "heavy_small_rays" a call of at most this many rays (default 6291456: a share of a frame) searches a scene of 32768
 *                     primitives or more in two launches too, handing over after 64 (0 = never)
 *   "gather_general"  1 = the gather shaders' general kernel also for scenes of at most four planes and spheres
 *   "refract_whole"   1 = a searched scene with glass in it is rendered by the one kernel that walks every ray's tree
 *                     (default 0: the frame's own pipeline renders every camera ray's hit, the trees start from there);
 *                     "refract_single" 1 = one lane per tree in that second step instead of four (measurement aid)
 *   "walk_blocks"     workgroups of the searched gather's walk (default 8192)
 *   "shade_block"     threads of a workgroup of the camera rays' shading kernel, "direct_block" of the one-kernel
 *                     renderers of refracted and mirror rays: 64 (default), 128 or 256
 *   "gather"          multi-device scenes only: 0 = every GPU writes its rows into the host pixmap (default), 1 = rows
 *                     gathered on the first device and read back from there
 *   "balance"         multi-device scenes only: 0 = a frame is dealt in interleaved 32-row tiles (default), 1 = in
 *                     contiguous slabs whose cuts are moved after every call by what each device's share took
 *   "tail_bottom_first", "tail_even", "copy_streams", "last_copied", "no_copy_prime", "debug_readback": measurement aids
 * RT355_ERR_ARG for a name that is not listed or a value out of range. */
int rt355_scene_set_option(rt355_scene *scene, const char *name, int64_t value);

/* The direction set of the gather shaders: the global `geo` that main.cpp:639 fills from Geodesic()
 * (main.cpp:408-450: the `v` lines of an OBJ file, centred on their mean and normalised).  xyz holds n
 * directions as the reference holds them after that function; they are copied.  n = 0 clears the set
 * (every hit then comes out black, as 0/0 does in shader.h:137,188). */
int rt355_scene_set_directions(rt355_scene *scene, const double *xyz, int32_t n);

/* sin and cos as the device computes them for the gather rotation: correctly rounded (double-double
 * evaluation), 0 <= x < 8.  Runs on the host; for checks against a C library. */
int rt355_sincos(double x, double *sin_out, double *cos_out);

#ifdef __cplusplus
}
#endif
#endif /* RT355_H */
