"""Scene builders: the reference's default scene and the synthetic benchmark scenes.

Everything is generated from fixed seeds with SplitMix64 so the same bytes are produced
on every machine (no dependence on numpy's generator versions).
"""
from __future__ import annotations

import math

import numpy as np

from .binding import OBJECT_DTYPE, OBJ_SPHERE, Scene, Shader, camera, light_src, _normalize as _n

_M64 = (1 << 64) - 1


class SplitMix64:
    def __init__(self, seed: int):
        self.s = seed & _M64

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & _M64
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)

    def uniform(self, lo=0.0, hi=1.0) -> float:
        return lo + (hi - lo) * ((self.next() >> 11) * (1.0 / (1 << 53)))

    def array(self, n: int) -> np.ndarray:
        """n uniform doubles in [0,1), vectorised SplitMix64."""
        idx = np.arange(1, n + 1, dtype=np.uint64)
        with np.errstate(over="ignore"):
            z = np.uint64(self.s) + idx * np.uint64(0x9E3779B97F4A7C15)
            self.s = int(z[-1]) if n else self.s
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
        return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


# ---- the reference's own scene (main.cpp:526-630) -------------------------------------
def default_scene() -> Scene:
    """The four objects main() pushes (main.cpp:547-581): two planes, two spheres, flat colour."""
    s = Scene()
    s.add_plane((0, 13, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
    s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    s.add_sphere((-10.5, 0, 0), 10, (100, 200, 100))
    s.add_sphere((10.5, 0, 0), 10, (200, 100, 100))
    return s


def default_lights(n=1):
    """light_src1 of main.cpp:623 (directional) and, for n == 2, a point light at the
    position of light_src2 (main.cpp:624)."""
    ls = [light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1))]
    if n >= 2:
        ls.append(light_src(1, (30, -30, 50), (0, 0, -1), (1, 1, 1)))
    return ls


def default_shader(n_lights=1) -> Shader:
    """Shader(light, Color(0x000000), Color(0xdddddd), Color(0x000000)), main.cpp:630"""
    return Shader(default_lights(n_lights), 0x000000, 0xDDDDDD, 0x000000)


def area_light(position=(-18.0, -28.0, 42.0), n0=(1.0, 0.0, 0.0), n1=(0.0, 0.6, 0.8), sx=36.0, sy=14.0, color=(1, 1, 1)):
    """light_src(3, position, direction, color, n0, n1, n2, sx, sy) of light.h:44-54: an area light whose area
    starts at `position` and spans n0 * sx and n1 * sy (what CalcSoftShadow aims its shadow rays at)."""
    return light_src(3, position, (0.3, 0.5, -1), color, n0, n1, sx, sy)


def soft_shader(kind="area") -> Shader:
    """Shaders for RT355_SHADING_SOFT.  CalcSoftShadow aims every light's shadow ray at lights[0]'s area
    (shader.h:441-443): "area" = an area light first, then the directional light of main.cpp:623 and a spot light;
    "point" = the two lights of default_shader(2), whose constructors leave the unit area of light.h:19-23."""
    if kind == "point":
        return default_shader(2)
    return Shader([area_light(), light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)),
                   light_src(2, (0, -25, 30), (0, 0.6, -1), (1, 0.9, 0.8))], 0x141720, 0xDDDDDD, 0x202020)


def default_camera(width, height):
    return camera((0, -20, 0), (0, 1, 0), (0, 0, 1), 10.0, width, height)


# ---- synthetic: random spheres + ring of point lights -----------------------------------
def spheres_scene(n_spheres=10000, seed=355) -> Scene:
    """n spheres, centres uniform in [-60,60]x[10,130]x[-10,50], radii uniform [0.3,1.5],
    colours uniform 40..255, over the floor plane z = -10."""
    rng = SplitMix64(seed)
    s = Scene()
    s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    u = rng.array(n_spheres * 7).reshape(n_spheres, 7)
    objs = np.zeros(n_spheres, dtype=OBJECT_DTYPE)
    objs["type"] = OBJ_SPHERE
    objs["p"][:, 0] = -60 + 120 * u[:, 0]
    objs["p"][:, 1] = 10 + 120 * u[:, 1]
    objs["p"][:, 2] = -10 + 60 * u[:, 2]
    objs["radius"] = 0.3 + 1.2 * u[:, 3]
    objs["rgb"] = (40 + np.floor(216 * u[:, 4:7])).astype(np.uint8)
    objs["fresnel"] = 1.0
    objs["ior"] = 1.0
    objs["tex0"] = objs["tex1"] = -1
    objs["n0"] = (1, 0, 0); objs["n1"] = (0, 1, 0); objs["n2"] = (0, 0, 1)
    s.objects.extend(list(objs))
    return s


def ring_lights(n=8, radius=80.0, z=80.0, center=(0.0, 70.0)):
    """n point lights (type 1) on a ring, colour (1,1,1)."""
    return [light_src(1, (center[0] + radius * math.cos(2 * math.pi * k / n),
                          center[1] + radius * math.sin(2 * math.pi * k / n), z), (0, 0, -1), (1, 1, 1))
            for k in range(n)]


def spheres_shader(n_lights=8) -> Shader:
    return Shader(ring_lights(n_lights), 0x101010, 0xDDDDDD, 0x000000)


# ---- synthetic: textured height-field mesh ------------------------------------------------
def checker_texture(size=512, cells=16, seed=356, dark=False) -> np.ndarray:
    """Procedural checker + noise, (size,size,3) uint8."""
    rng = SplitMix64(seed + (1 if dark else 0))
    yy, xx = np.mgrid[0:size, 0:size]
    chk = (((xx * cells) // size + (yy * cells) // size) & 1).astype(np.float64)
    noise = rng.array(size * size * 3).reshape(size, size, 3)
    base = np.where(chk[..., None] > 0, np.array([220.0, 180.0, 90.0]), np.array([60.0, 110.0, 200.0]))
    img = base * (0.35 if dark else 1.0) + 30.0 * noise
    return np.clip(img, 0, 255).astype(np.uint8)


def heightfield_tris(nx: int, ny: int, seed=356, x0=-60.0, x1=60.0, y0=10.0, y1=130.0, zbase=-8.0,
                     amp=6.0) -> np.ndarray:
    """A height field z = f(x,y) over [x0,x1]x[y0,y1] as 2*nx*ny triangles wound so their
    normals point up (+z, toward the camera and the lights), with vt in [0,1]^2."""
    rng = SplitMix64(seed)
    gx = np.linspace(x0, x1, nx + 1)
    gy = np.linspace(y0, y1, ny + 1)
    X, Y = np.meshgrid(gx, gy, indexing="xy")
    jitter = rng.array((nx + 1) * (ny + 1)).reshape(ny + 1, nx + 1)
    Z = (zbase + amp * (np.sin(X * 0.11) * np.cos(Y * 0.07) + 0.5 * np.sin(X * 0.31 + Y * 0.23))
         + 0.35 * (jitter - 0.5))
    U = (X - x0) / (x1 - x0) * 0.999
    Vv = (Y - y0) / (y1 - y0) * 0.999
    P = np.stack([X, Y, Z], axis=-1)
    T = np.stack([U, Vv], axis=-1)
    a, b = P[:-1, :-1], P[:-1, 1:]
    c, d = P[1:, :-1], P[1:, 1:]
    ta, tb, tc, td = T[:-1, :-1], T[:-1, 1:], T[1:, :-1], T[1:, 1:]
    # (a,b,d) and (a,d,c): (b-a)x(d-a) has +z
    t1 = np.concatenate([a, b, d, ta, tb, td], axis=-1).reshape(-1, 15)
    t2 = np.concatenate([a, d, c, ta, td, tc], axis=-1).reshape(-1, 15)
    out = np.empty((t1.shape[0] * 2, 15), dtype=np.float64)
    out[0::2] = t1
    out[1::2] = t2
    return out


def mesh_scene(n_triangles=100_000, seed=356, tex_size=512, textured=True) -> Scene:
    """Floor plane + one textured height-field mesh of about n_triangles triangles."""
    n = max(1, int(round(math.sqrt(n_triangles / 2.0))))
    s = Scene()
    s.add_plane((0, 0, -16), (0, 0, 1), (1, 0, 0), (230, 230, 230))
    m = s.add_mesh(heightfield_tris(n, n, seed))
    if textured:
        s.add_texture(m, checker_texture(tex_size, 16, seed, dark=False), 1, 1)
        s.add_texture(m, checker_texture(tex_size, 16, seed, dark=True), 1, 1)
        s.set_texture_type(m, 1)
    return s


def mesh_shader() -> Shader:
    """Two point lights above the height field."""
    return Shader([light_src(1, (-50, 20, 70), (0, 0, -1), (1, 1, 1)),
                   light_src(1, (60, 90, 90), (0, 0, -1), (1, 1, 1))], 0x101010, 0xDDDDDD, 0x000000)


def orbit_camera(frame: int, width: int, height: int, n_frames=60, center=(0.0, 70.0, 0.0), radius=90.0,
                 eye_z=35.0):
    """Camera `frame` of a sweep orbiting the scene centre in 360/n_frames degree steps."""
    ang = 2 * math.pi * frame / n_frames - math.pi / 2
    eye = (center[0] + radius * math.cos(ang), center[1] + radius * math.sin(ang), eye_z)
    view = (center[0] - eye[0], center[1] - eye[1], center[2] - eye[2])
    return camera(eye, view, (0, 0, 1), 10.0, width, height)


# ---- small named cases shared by the golden-vector generator and the parity tests ---------
def _small_texture(w, h, seed):
    rng = SplitMix64(seed)
    return np.floor(rng.array(w * h * 3) * 256).astype(np.uint8).reshape(h, w, 3)


def _grid_mesh(n, seed, x0=-14.0, x1=14.0, y0=-4.0, y1=10.0, zbase=-6.0, amp=2.5):
    return heightfield_tris(n, n, seed, x0, x1, y0, y1, zbase, amp)


def small_case(name: str):
    """Returns (Scene, Shader, camera, width, height) for a named parity case."""
    if name == "default_1light":
        w, h = 200, 150
        return default_scene(), default_shader(1), default_camera(w, h), w, h
    if name == "default_2lights":
        w, h = 200, 150
        return default_scene(), default_shader(2), default_camera(w, h), w, h
    if name == "no_lights":
        w, h = 64, 48
        return default_scene(), Shader([], 0x000000, 0xDDDDDD, 0x000000), default_camera(w, h), w, h
    if name == "empty_scene":
        w, h = 32, 24
        return Scene(), default_shader(1), default_camera(w, h), w, h
    if name == "textured":
        # textured wall + textured floor + textured sphere + flat sphere; directional, spot, point
        w, h = 200, 150
        s = Scene()
        p1 = s.add_plane((0, 13, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
        p2 = s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
        s1 = s.add_sphere((-10.5, 0, 0), 10, (100, 200, 100), (1, 0, 0), (0, 1, 0), (0, 0, 1))
        s.add_sphere((10.5, 0, 2), 9, (200, 100, 100))
        s.add_texture(p1, _small_texture(16, 8, 11), 7.0, 5.0); s.add_texture(p1, _small_texture(8, 8, 12), 7.0, 5.0)
        s.add_texture(p2, _small_texture(12, 12, 13), 9.5, 9.5); s.add_texture(p2, _small_texture(12, 12, 14), 9.5, 9.5)
        s.add_texture(s1, _small_texture(32, 16, 15), 1.0, 1.0); s.add_texture(s1, _small_texture(32, 16, 16), 0.5, 0.5)
        for o in (p1, p2, s1):
            s.set_texture_type(o, 1)
        sh = Shader([light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)),
                     light_src(2, (0, -25, 30), (0, 0.6, -1), (1, 0.9, 0.8)),
                     light_src(1, (30, -30, 50), (0, 0, -1), (0.7, 0.8, 1.0))], 0x141720, 0xDCC8FA, 0xF0F0F0)
        return s, sh, default_camera(w, h), w, h
    if name == "mesh":
        # 512-triangle textured grid mesh between a floor and two spheres, two point lights
        w, h = 160, 90
        s = Scene()
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
        s.add_sphere((-9, 12, 2), 4, (100, 200, 100))
        m = s.add_mesh(_grid_mesh(16, 21))
        s.add_sphere((8, 2, 3), 3, (200, 100, 100))
        s.add_texture(m, _small_texture(16, 16, 22), 1, 1); s.add_texture(m, _small_texture(16, 16, 23), 1, 1)
        s.set_texture_type(m, 1)
        sh = Shader([light_src(1, (30, -30, 50), (0, 0, -1), (1, 1, 1)),
                     light_src(1, (-25, 5, 30), (0, 0, -1), (1, 1, 1))], 0x101010, 0xDDDDDD, 0x000000)
        return s, sh, default_camera(w, h), w, h
    if name == "mesh_flat":
        # untextured mesh listed BEFORE the floor, directional + point light
        w, h = 96, 64
        s = Scene()
        s.add_mesh(_grid_mesh(6, 31), (150, 200, 110))
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
        sh = Shader([light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)),
                     light_src(1, (5, -10, 40), (0, 0, -1), (1, 1, 1))], 0x000000, 0xDDDDDD, 0x000000)
        return s, sh, default_camera(w, h), w, h
    if name == "spheres300":
        w, h = 160, 120
        s = spheres_scene(300, seed=77)
        return s, Shader(ring_lights(3), 0x101010, 0xDDDDDD, 0x202020), default_camera(w, h), w, h
    if name == "spheres300_dir":
        # directional + spot + point light over enough primitives to take the search path
        w, h = 160, 120
        s = spheres_scene(300, seed=78)
        sh = Shader([light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)),
                     light_src(2, (0, 20, 90), (0.1, 0.5, -1), (1, 0.9, 0.8)),
                     light_src(0, (50, 0, 100), (0.3, 0.2, -1), (0.6, 0.7, 1.0))], 0x101010, 0xDDDDDD, 0x202020)
        return s, sh, default_camera(w, h), w, h
    if name == "mesh_dir":
        # mesh + spheres lit by a directional light coming from below the height field's front side
        w, h = 128, 96
        s = Scene()
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
        m = s.add_mesh(_grid_mesh(8, 41), (150, 200, 110))
        s.add_sphere((-6, 2, 1), 2.5, (100, 200, 100))
        s.add_sphere((5, 6, 4), 2.0, (200, 100, 100))
        sh = Shader([light_src(0, (0, -50, 100), (0.2, 0.6, -1), (1, 1, 1)),
                     light_src(0, (0, 0, -100), (0, 0.1, 1), (1, 1, 1))], 0x000000, 0xDDDDDD, 0x000000)
        return s, sh, default_camera(w, h), w, h
    if name == "inside":
        # eye inside a big sphere, a light inside another, overlapping spheres, a plane behind the eye
        w, h = 120, 90
        s = Scene()
        s.add_sphere((0, -18, 0), 30, (90, 120, 200))
        s.add_sphere((-4, -5, 0), 5, (200, 100, 100))
        s.add_sphere((1, -6, 1), 4, (100, 200, 100))
        s.add_plane((0, -40, 0), (0, 1, 0), (-1, 0, 0), (214, 147, 44))
        s.add_sphere((30, -30, 50), 6, (250, 250, 0))
        sh = Shader([light_src(1, (30, -30, 50), (0, 0, -1), (1, 1, 1)),
                     light_src(1, (2, -12, 6), (0, 0, -1), (1, 1, 1))], 0x0A0A0A, 0xDDDDDD, 0x000000)
        return s, sh, default_camera(w, h), w, h
    if name == "mirrors":
        # the default scene with reflecting objects (ks of anyobject.h:22): a faint floor, a mirror wall, one
        # glossy and one almost perfect mirror sphere; object 0 reflects, so rays that miss go on (main.cpp:94)
        w, h = 160, 120
        s = default_scene()
        for o, ks in zip(s.objects, (0.25, 0.5, 0.6, 0.95)):
            o["ks"] = ks
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "mirrors_mesh":
        # a reflecting textured mesh over a matt floor, two mirror spheres, three lights incl. a directional one
        w, h = 128, 96
        s, sh, cam, _, _ = small_case("mesh_dir")
        for o, ks in zip(s.objects, (0.0, 0.35, 0.8, 0.5, 0.5, 0.5)):
            o["ks"] = ks
        return s, sh, default_camera(w, h), w, h
    if name == "environment":
        # a floor, a matt and a mirror sphere under a textured sky (the Environment of main.cpp:591-595, which HEAD
        # constructs and leaves out of the list): sky seen directly, in the mirror, and by the shadow rays
        w, h = 160, 120
        s = Scene()
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230), ks=0.2)
        s.add_sphere((-10.5, 0, 0), 10, (100, 200, 100))
        s.add_sphere((10.5, 0, 0), 10, (200, 100, 100), ks=0.6)
        env = s.add_environment((0, -50, 20), 1000, (255, 255, 0), (-1, 1, 0), (-1, -1, 0), (0, 0, 1))
        s.add_texture(env, _small_texture(64, 32, 51), 1, 1); s.add_texture(env, _small_texture(64, 32, 52), 1, 1)
        s.set_texture_type(env, 1)
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "environment_spheres":
        # enough spheres to be searched, some of them mirrors, inside a flat-coloured Environment listed FIRST
        # (so that rays that miss everything else still end on object 0 -- and chains go on with its ks)
        w, h = 128, 96
        s = Scene()
        s.add_environment((0, 60, 10), 400, (90, 140, 230), ks=0.3)
        sp = spheres_scene(150, seed=79)
        for i, o in enumerate(sp.objects):
            if i % 4 == 1:
                o["ks"] = 0.7
        s.objects.extend(sp.objects)
        return s, Shader(ring_lights(3), 0x101010, 0xDDDDDD, 0x202020), default_camera(w, h), w, h
    if name == "solid_texture":
        # texture_type 2 (shader.h:59-63,296-319): two maps projected along a tilted axis onto a sphere, the back
        # wall and a mesh; outside the projector's rectangle the objects keep their own colour.  The second sphere uses
        # the frame AnyObject::AddTexture leaves (projection along z about the origin).
        w, h = 160, 120
        s = default_scene()
        frame = (_n((1, 0.2, 0)), _n((-0.1, 0.1, 1)), _n((0.1, -1, 0.2)), (-3.0, 0.0, 1.0))
        for o, (sx, sy) in ((0, (70.0, 50.0)), (2, (24.0, 18.0))):
            s.add_texture(o, _small_texture(32, 16, 61 + o), sx, sy, frame=frame)
            s.add_texture(o, _small_texture(16, 16, 71 + o), sx, sy, frame=frame)
            s.set_texture_type(o, 2)
        s.add_texture(3, _small_texture(24, 24, 81), 12.0, 40.0); s.add_texture(3, _small_texture(8, 8, 82), 1, 1)
        s.set_texture_type(3, 2)
        m = s.add_mesh(heightfield_tris(6, 6, seed=5, x0=-20, x1=-6, y0=-6, y1=4, zbase=-8, amp=2.0), ks=0.3)
        s.add_texture(m, _small_texture(16, 32, 91), 25.0, 25.0, frame=(_n((0, 1, 0)), _n((0, 0, 1)), _n((1, 0, 0)), (-13.0, 0.0, -6.0)))
        s.add_texture(m, _small_texture(16, 32, 92), 1, 1)
        s.set_texture_type(m, 2)
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "function_3d":
        # texture_type 3 (shader.h:64-68, texture.h:164-177), as main.cpp:586-588 sets it on sphere3: bands on the
        # spheres, the floor and an Environment; textures[0] is only called through
        w, h = 160, 120
        s = default_scene()
        for o in (1, 2, 3):
            s.add_texture(o, _small_texture(8, 8, 17), 0.5, 0.5); s.add_texture(o, _small_texture(8, 8, 17), 0.5, 0.5)
            s.set_texture_type(o, 3)
        s.objects[3]["ks"] = 0.4
        env = s.add_environment((0, 0, 0), 500, (40, 40, 90))
        s.add_texture(env, _small_texture(8, 8, 18), 1, 1)
        s.set_texture_type(env, 3)
        del s.objects[0]   # the back wall: the sky behind the spheres
        return s, default_shader(2), default_camera(w, h), w, h
    if name in ("quadrics", "quadrics_spheres"):
        # one of each Quadric subclass (anyobject.h:355-442), the default Quadric (a plane) and a general one, on
        # tilted axes; the unbounded ones cross each other and the floor.  "quadrics_spheres": the same among
        # enough spheres for the searched modes, where a quadric is a candidate of every ray.
        w, h = 160, 120
        s = Scene()
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230), ks=0.2)
        ax = (_n((1, 0.1, 0)), _n((-0.1, 1, 0.2)), _n((0, -0.2, 1)))
        s.add_quadric("Ellipsoid", (-12, 2, 0), *ax, (6, 4, 8), (200, 90, 90), ks=0.5)
        s.add_quadric("Cylinder", (14, 8, 0), *ax, (3, 3, 1), (90, 200, 90))
        s.add_quadric("Cone", (2, 10, 6), (1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 2.5), (90, 90, 220))
        s.add_quadric("EllipticParaboloid", (-2, -2, -9), *ax, (2, 2, 0.4), (220, 200, 80), ks=0.3)
        s.add_quadric("HyperbolicParaboloid", (20, 30, -5), (1, 0, 0), (0, 1, 0), (0, 0, 1), (4, 4, 0.5), (80, 200, 220))
        s.add_quadric("Quadric", (0, 38, 0), (1, 0, 0), (0, 0, 1), (0, -1, 0), (1, 1, 2), (180, 180, 180))
        s.add_quadric((2, -1, 1, 1, -3), (-22, 12, 4), *ax, (3, 5, 2), (160, 120, 200))
        if name == "quadrics_spheres":
            w, h = 128, 96
            sp = spheres_scene(120, seed=83)
            for i, o in enumerate(sp.objects):
                if i % 5 == 1:
                    o["ks"] = 0.6
            s.objects.extend(sp.objects)
            return s, Shader(ring_lights(3), 0x101010, 0xDDDDDD, 0x202020), default_camera(w, h), w, h
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "glass":
        # refraction (main.cpp:116-154): the reference's scene with both spheres of glass, one of them a mirror
        # too, in a closed room like main()'s six planes (a ray that leaves a glass object and hits nothing never
        # ends in the reference: it goes on from the origin with the glass object's coefficients, main.cpp:94,116)
        w, h = 160, 120
        s = default_scene()
        s.add_plane((-40, 0, 0), (1, 0, 0), (0, 1, 0), (220, 180, 180))
        s.add_plane((40, 0, 0), (-1, 0, 0), (0, 1, 0), (180, 220, 180))
        s.add_plane((0, 0, 45), (0, 0, -1), (1, 0, 0), (180, 180, 220))
        s.add_plane((0, -70, 0), (0, 1, 0), (1, 0, 0), (210, 210, 170))
        s.objects[0]["ks"] = 0.3
        s.objects[2]["fresnel"] = 0.2; s.objects[2]["ior"] = 1.5; s.objects[2]["ks"] = 0.3
        s.objects[3]["fresnel"] = 0.6; s.objects[3]["ior"] = 1.33; s.objects[3]["ks"] = 0.5
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "glass_mixed":
        # glass of every kind inside an Environment (so that every ray ends somewhere): a pane (a Plane), an
        # ellipsoid, a paraboloid, a mesh (whose inside call never hits, anyobject.h:499,659), a textured sphere,
        # a nearly index-matched sphere
        w, h = 128, 96
        s = Scene()
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230), ks=0.2)
        pane = s.add_plane((0, 4, 0), (0.1, -1, 0.05), (1, 0, 0), (200, 220, 255), ks=0.1)
        ax = (_n((1, 0.1, 0)), _n((-0.1, 1, 0.2)), _n((0, -0.2, 1)))
        ell = s.add_quadric("Ellipsoid", (-12, 12, 0), *ax, (6, 4, 8), (200, 90, 90), ks=0.4)
        par = s.add_quadric("EllipticParaboloid", (12, 16, -9), *ax, (2, 2, 0.4), (220, 200, 80))
        msh = s.add_mesh(heightfield_tris(5, 5, seed=9, x0=-6, x1=6, y0=8, y1=16, zbase=-6, amp=2.0), ks=0.2)
        sp1 = s.add_sphere((8, 0, -4), 5, (100, 200, 100), ks=0.3)
        sp2 = s.add_sphere((-7, -2, -6), 3.5, (210, 210, 100))
        s.add_texture(sp1, _small_texture(32, 16, 33), 1, 1); s.add_texture(sp1, _small_texture(32, 16, 34), 1, 1)
        s.set_texture_type(sp1, 1)
        env = s.add_environment((0, 0, 0), 300, (120, 160, 240), ks=0.1)
        for o, fr, ior in ((pane, 0.3, 1.5), (ell, 0.25, 1.4), (par, 0.5, 1.2), (msh, 0.5, 1.3), (sp1, 0.4, 1.6),
                           (sp2, 0.1, 1.05)):
            s.objects[o]["fresnel"] = fr; s.objects[o]["ior"] = ior
        return s, default_shader(2), default_camera(w, h), w, h
    if name == "main_all":
        # everything main() constructs (main.cpp:547-617), the objects it leaves out of the list at HEAD pushed
        # too: six planes, three spheres (sphere3 a perfect mirror with the 3D-function texture), the textured
        # Environment, two meshes with ks = 1 (height fields stand in for geodesic.obj and dodecahandle.obj, which
        # the repository does not hold), both lights of main.cpp:623-624
        w, h = 160, 120
        s = Scene()
        s.add_plane((0, 13, 0), (0, -1, 0), (-1, 0, 0), (200, 200, 200))
        s.add_plane((0, -100, 0), (0, 1, 0), (-1, 0, 0), (214, 147, 44))
        s.add_plane((0, 0, -10), (0, 0, 1), (1, 0, 0), (230, 230, 230))
        s.add_plane((0, 0, 100), (0, 0, -1), (1, 0, 0), (255, 0, 0))
        s.add_plane((100, 0, 0), (-1, 0, 0), (0, 1, 0), (0, 255, 0))
        s.add_plane((-100, 0, 0), (1, 0, 0), (0, 1, 0), (0, 255, 0))
        s.add_sphere((-10.5, 0, 0), 10, (100, 200, 100))
        s.add_sphere((10.5, 0, 0), 10, (200, 100, 100))
        s3 = s.add_sphere((-20, -20, 20), 10, (73, 179, 248), ks=1)
        s.add_texture(s3, _small_texture(16, 16, 3), 0.5, 0.5); s.add_texture(s3, _small_texture(16, 16, 3), 0.5, 0.5)
        s.set_texture_type(s3, 3)
        env = s.add_environment((0, -50, 20), 1000, (255, 255, 0), (-1, 1, 0), (-1, -1, 0), (0, 0, 1))
        s.add_texture(env, _small_texture(64, 32, 4), 1, 1); s.add_texture(env, _small_texture(64, 32, 4), 1, 1)
        s.set_texture_type(env, 1)
        m1 = s.add_mesh(heightfield_tris(5, 5, seed=11, x0=-8, x1=8, y0=-12, y1=-4, zbase=2, amp=3.0), ks=1)
        s.objects[m1]["ior"] = 1.1
        s.add_texture(m1, _small_texture(16, 16, 5), 0.1, 0.1); s.add_texture(m1, _small_texture(16, 16, 6), 0.1, 0.1)
        m2 = s.add_mesh(heightfield_tris(4, 4, seed=12, x0=-26, x1=-14, y0=-34, y1=-26, zbase=14, amp=2.0), ks=1)
        s.add_texture(m2, _small_texture(16, 16, 5), 1, 1); s.add_texture(m2, _small_texture(16, 16, 5), 1, 1)
        s.set_texture_type(m2, 1)
        lights = [light_src(0, (-100, -30, 100), (-1, 1, -1), (1, 1, 1)), light_src(3, (30, -30, 50), (0, 0, -1), (1, 1, 1))]
        return s, Shader(lights, 0x000000, 0xDDDDDD, 0x000000), default_camera(w, h), w, h
    raise KeyError(name)


SMALL_CASES = ["default_1light", "default_2lights", "no_lights", "empty_scene", "textured", "mesh",
               "mesh_flat", "spheres300", "spheres300_dir", "mesh_dir", "inside", "mirrors", "mirrors_mesh",
               "environment", "environment_spheres", "solid_texture", "function_3d",
               "quadrics", "quadrics_spheres", "glass", "glass_mixed", "main_all"]
