"""ctypes/numpy mirror of include/rt355.h.

Host-side scene containers laid out exactly as the C ABI expects, so a scene built in
Python (tests, benchmark) is handed to the HIP library -- or to the CPU checker -- without
copies.  Names follow the reference's scene vocabulary (objects, triangles, textures,
lights; image_synthesis/anyobject.h, light.h, texture.h).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import List, Sequence

import numpy as np

OBJ_PLANE, OBJ_SPHERE, OBJ_MESH, OBJ_ENVIRONMENT, OBJ_QUADRIC = 0, 1, 2, 3, 4
# aCoor of Quadric's subclasses (anyobject.h:355-442) and of the default Quadric, "a plane" (:242-249)
QUADRIC_KINDS = {"Quadric": (0, 0, 0, 1, 0), "Cone": (1, 1, -1, 0, 0), "Ellipsoid": (1, 1, 1, 0, -1),
                 "Cylinder": (1, 1, 0, 0, -1), "EllipticParaboloid": (1, 1, 0, -1, 0),
                 "HyperbolicParaboloid": (1, -1, 0, -1, 0)}
MODE_ALLPAIRS, MODE_BRUTE, MODE_PACKET = 0, 1, 2
SHADING_LIGHTS, SHADING_OCCLUSION, SHADING_BLEED, SHADING_SOFT = 0, 1, 2, 3

ERR = {0: "OK", -1: "ERR_ARG", -2: "ERR_UNSUPPORTED", -3: "ERR_DEVICE", -4: "ERR_NOMEM"}

# struct rt355_object (232 bytes)
OBJECT_DTYPE = np.dtype(
    {
        "names": ["p", "n", "n0", "n1", "n2", "radius", "ks", "fresnel", "ior", "type",
                  "texture_type", "tex0", "tex1", "tri_begin", "tri_count", "rgb", "pad_", "s", "a", "pad2_"],
        "formats": [("f8", 3), ("f8", 3), ("f8", 3), ("f8", 3), ("f8", 3), "f8", "f8", "f8", "f8",
                    "i4", "i4", "i4", "i4", "i4", "i4", ("u1", 3), ("u1", 5), ("f8", 3), ("i4", 5), "i4"],
        "offsets": [0, 24, 48, 72, 96, 120, 128, 136, 144, 152, 156, 160, 164, 168, 172, 176, 179, 184, 208, 228],
        "itemsize": 232,
    }
)
# struct rt355_light (144 bytes)
LIGHT_DTYPE = np.dtype(
    {
        "names": ["position", "direction", "color", "type", "pad_", "n0", "n1", "sx", "sy"],
        "formats": [("f8", 3), ("f8", 3), ("f8", 3), "i4", "i4", ("f8", 3), ("f8", 3), "f8", "f8"],
        "offsets": [0, 24, 48, 72, 76, 80, 104, 128, 136],
        "itemsize": 144,
    }
)


class CTexture(C.Structure):
    _fields_ = [("rgb", C.c_void_p), ("sx", C.c_double), ("sy", C.c_double),
                ("width", C.c_int32), ("height", C.c_int32),
                ("n0", C.c_double * 3), ("n1", C.c_double * 3), ("n2", C.c_double * 3), ("p_center", C.c_double * 3)]


class CSceneDesc(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("n_objects", C.c_int32), ("pad0_", C.c_int32),
                ("triangles", C.c_void_p), ("n_triangles", C.c_int32), ("pad1_", C.c_int32),
                ("textures", C.c_void_p), ("n_textures", C.c_int32), ("pad2_", C.c_int32)]


class CCamera(C.Structure):
    _fields_ = [("v_up", C.c_double * 3), ("v_view", C.c_double * 3), ("p_eye", C.c_double * 3),
                ("dist", C.c_double), ("s_x", C.c_double), ("s_y", C.c_double)]


class CShader(C.Structure):
    _fields_ = [("lights", C.c_void_p), ("n_lights", C.c_int32), ("ambient", C.c_uint8 * 3),
                ("specular", C.c_uint8 * 3), ("border", C.c_uint8 * 3), ("pad_", C.c_uint8 * 3)]


class COpts(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("alias", C.c_int32),
                ("row_begin", C.c_int32), ("row_end", C.c_int32), ("mode", C.c_int32),
                ("seed", C.c_uint32), ("shading", C.c_int32),
                ("tile_rows", C.c_int32), ("tile_count", C.c_int32), ("tile_phase", C.c_int32), ("pad_", C.c_int32)]


class CStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("pair_tests", C.c_uint64), ("exact_tests", C.c_uint64),
                ("overflow_rays", C.c_uint64), ("ms_device", C.c_double), ("packet_survivors", C.c_uint64), ("primary_hits", C.c_uint64),
                ("secondary_rays", C.c_uint64), ("gather_rays", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


assert C.sizeof(CTexture) == 128 and C.sizeof(CSceneDesc) == 48 and C.sizeof(CCamera) == 96
assert C.sizeof(CShader) == 24 and C.sizeof(COpts) == 48 and C.sizeof(CStats) == 80


def _normalize(v: Sequence[float]):
    """Normalize() of image_synthesis/vector3D.h:79-88, same operation order."""
    x, y, z = float(v[0]), float(v[1]), float(v[2])
    l = math.sqrt(x * x + y * y + z * z)
    if l != 0:
        return (x / l, y / l, z / l)
    return (x, y, z)


def rgb_of(clr) -> tuple:
    """Color(unsigned int) of color.h:27-31, or a 3-sequence of bytes."""
    if isinstance(clr, int):
        return ((clr & 0xFF0000) >> 16, (clr & 0xFF00) >> 8, clr & 0xFF)
    return tuple(int(c) & 0xFF for c in clr)


@dataclass
class Texture:
    """A loaded Texture (texture.h:23-79): rgb is (height, width, 3) uint8 in the
    reference's in-memory order (row 0 = bottom row of the BMP)."""
    rgb: np.ndarray
    sx: float = 1.0
    sy: float = 1.0
    # the projector's frame of a solid texture (texture.h:32-35), the defaults of the Texture ctor
    n0: tuple = (1.0, 0.0, 0.0)
    n1: tuple = (0.0, 1.0, 0.0)
    n2: tuple = (0.0, 0.0, 1.0)
    p_center: tuple = (0.0, 0.0, 0.0)


@dataclass
class Scene:
    """Scene::objs (main.cpp:229) flattened: objects in list order, the triangles of all
    meshes, and the textures they refer to."""
    objects: List[np.void] = field(default_factory=list)
    triangles: List[np.ndarray] = field(default_factory=list)  # each (n,15) float64
    textures: List[Texture] = field(default_factory=list)
    _n_tri: int = 0

    # -- builders mirroring the reference constructors --------------------------------
    def _new(self, kind, rgb, ks=0.0):
        o = np.zeros((), dtype=OBJECT_DTYPE)
        o["type"] = kind
        o["rgb"] = rgb_of(rgb)
        o["ks"] = ks
        o["fresnel"] = 1.0
        o["ior"] = 1.0
        o["tex0"] = o["tex1"] = -1
        o["n0"] = (1, 0, 0); o["n1"] = (0, 1, 0); o["n2"] = (0, 0, 1)
        return o

    def add_plane(self, pi, ni, n0, clr, ks=0.0):
        """Plane(Point3D pi, Vector3D ni, Vector3D n0, Color clr, double ks) anyobject.h:55"""
        o = self._new(OBJ_PLANE, clr, ks)
        o["p"] = pi; o["n"] = ni; o["n0"] = n0
        self.objects.append(o)
        return len(self.objects) - 1

    def add_sphere(self, center, radius, clr, n0=(1, 0, 0), n1=(0, 1, 0), n2=(0, 0, 1), ks=0.0):
        """Sphere(Point3D, double radius, Color, Vector3D n0, n1, n2, double ks) anyobject.h:113"""
        o = self._new(OBJ_SPHERE, clr, ks)
        o["p"] = center; o["radius"] = radius
        o["n0"] = n0; o["n1"] = n1; o["n2"] = n2
        self.objects.append(o)
        return len(self.objects) - 1

    def add_environment(self, center, radius, clr, n0=(1, 0, 0), n1=(0, 1, 0), n2=(0, 0, 1), ks=0.0):
        """Environment(Point3D, double radius, Color, Vector3D n0, n1, n2, double ks) anyobject.h:181: the sky
        sphere seen from inside; its constructor normalises the three axes."""
        o = self._new(OBJ_ENVIRONMENT, clr, ks)
        o["p"] = center; o["radius"] = radius
        o["n0"] = _normalize(n0); o["n1"] = _normalize(n1); o["n2"] = _normalize(n2)
        self.objects.append(o)
        return len(self.objects) - 1

    def add_quadric(self, kind, center, n0, n1, n2, s, clr, ks=0.0):
        """A Quadric or one of its subclasses (anyobject.h:231-442) after SetPCenter, SetN, SetS, SetColor and
        SetKs: kind is a class name of QUADRIC_KINDS or the five aCoor themselves.  SetS takes floats
        (anyobject.h:265), so the scales are rounded to float as they are there; SetN stores the axes as given."""
        o = self._new(OBJ_QUADRIC, clr, ks)
        o["a"] = QUADRIC_KINDS[kind] if isinstance(kind, str) else tuple(int(v) for v in kind)
        o["p"] = center
        o["n0"] = n0; o["n1"] = n1; o["n2"] = n2
        o["s"] = tuple(float(np.float32(v)) for v in s)
        self.objects.append(o)
        return len(self.objects) - 1

    def add_mesh(self, tris: np.ndarray, clr=(150, 200, 110), ks=0.0):
        """A Mesh (anyobject.h:534-675) given as (n,15) float64 rows
        p0 p1 p2 uv0 uv1 uv2; default colour as Mesh::clr (anyobject.h:536)."""
        tris = np.ascontiguousarray(tris, dtype=np.float64).reshape(-1, 15)
        o = self._new(OBJ_MESH, clr, ks)
        o["tri_begin"] = self._n_tri
        o["tri_count"] = len(tris)
        self._n_tri += len(tris)
        self.triangles.append(tris)
        self.objects.append(o)
        return len(self.objects) - 1

    def add_texture(self, obj: int, rgb: np.ndarray, sx=1.0, sy=1.0, frame=None):
        """AnyObject::AddTexture (anyobject.h:32-35); the first call sets textures[0], the
        second textures[1].  frame = (n0, n1, n2, p_center): a Texture constructed with its projector
        (texture.h:42) and pushed onto AnyObject::textures directly, for texture_type 2."""
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        t = Texture(rgb, float(sx), float(sy))
        if frame is not None:
            t.n0, t.n1, t.n2, t.p_center = (tuple(float(c) for c in v) for v in frame)
        self.textures.append(t)
        tid = len(self.textures) - 1
        o = self.objects[obj]
        if o["tex0"] < 0:
            o["tex0"] = tid
        elif o["tex1"] < 0:
            o["tex1"] = tid
        else:
            raise ValueError("only textures[0] and textures[1] are used on this path")
        return tid

    def set_texture_type(self, obj: int, texture_type: int):
        self.objects[obj]["texture_type"] = texture_type

    # -- C view ------------------------------------------------------------------------
    def pack(self) -> "PackedScene":
        return PackedScene(self)

    @property
    def n_primitives(self) -> int:
        return sum(int(o["tri_count"]) if o["type"] == OBJ_MESH else 1 for o in self.objects)


class PackedScene:
    """Owns the contiguous arrays a CSceneDesc points into."""

    def __init__(self, s: Scene):
        self.objects = np.array(s.objects, dtype=OBJECT_DTYPE) if s.objects else np.zeros(0, OBJECT_DTYPE)
        self.triangles = (np.ascontiguousarray(np.concatenate(s.triangles, axis=0))
                          if s.triangles else np.zeros((0, 15), np.float64))
        self.tex_arrays = [np.ascontiguousarray(t.rgb) for t in s.textures]
        self.ctex = (CTexture * max(1, len(s.textures)))()
        for i, t in enumerate(s.textures):
            self.ctex[i].rgb = self.tex_arrays[i].ctypes.data
            self.ctex[i].sx, self.ctex[i].sy = t.sx, t.sy
            self.ctex[i].n0[:], self.ctex[i].n1[:], self.ctex[i].n2[:] = t.n0, t.n1, t.n2
            self.ctex[i].p_center[:] = t.p_center
            self.ctex[i].height, self.ctex[i].width = t.rgb.shape[0], t.rgb.shape[1]
        d = CSceneDesc()
        d.objects = self.objects.ctypes.data if len(self.objects) else None
        d.n_objects = len(self.objects)
        d.triangles = self.triangles.ctypes.data if len(self.triangles) else None
        d.n_triangles = len(self.triangles)
        d.textures = C.addressof(self.ctex) if s.textures else None
        d.n_textures = len(s.textures)
        self.desc = d


def light_src(type_: int, position, direction, color=(1, 1, 1), n0=(1, 0, 0), n1=(0, 1, 0), sx=1.0, sy=1.0) -> np.void:
    """light_src(int type, Point3D position, Vector3D direction, Vector3D color), light.h:27-32:
    the stored direction is Normalize(direction_in).  n0, n1, sx, sy: the area of the type 3 constructor
    (light.h:44-54, stored as given; n2 is read nowhere); the other constructors leave the defaults of
    light.h:19-23, which CalcSoftShadow reads all the same when the light is lights[0]."""
    l = np.zeros((), dtype=LIGHT_DTYPE)
    l["type"] = type_
    l["position"] = position
    l["direction"] = _normalize(direction)
    l["color"] = color
    l["n0"] = n0; l["n1"] = n1; l["sx"] = sx; l["sy"] = sy
    return l


class Shader:
    """Shader(vector<light_src*>, Color ambient, Color specular, Color border), shader.h:32."""

    def __init__(self, lights, ambient=0x000000, specular=0xDDDDDD, border=0x000000):
        self.lights = np.array(list(lights), dtype=LIGHT_DTYPE) if len(lights) else np.zeros(0, LIGHT_DTYPE)
        c = CShader()
        c.lights = self.lights.ctypes.data if len(self.lights) else None
        c.n_lights = len(self.lights)
        c.ambient[:] = rgb_of(ambient)
        c.specular[:] = rgb_of(specular)
        c.border[:] = rgb_of(border)
        self.c = c


def camera(p_eye=(0, -20, 0), v_view=(0, 1, 0), v_up=(0, 0, 1), dist=10.0, width=800, height=600,
           s_x=None, s_y=None) -> CCamera:
    """The camera main() sets up (main.cpp:529-541): s_x = dist*4*width/height, s_y = dist*4."""
    c = CCamera()
    c.p_eye[:] = p_eye; c.v_view[:] = v_view; c.v_up[:] = v_up
    c.dist = dist
    c.s_x = dist * 4.0 * width / height if s_x is None else s_x
    c.s_y = dist * 4 if s_y is None else s_y
    return c


def opts(width, height, alias=0, row_begin=0, row_end=0, mode=MODE_PACKET, seed=0, shading=SHADING_LIGHTS,
         tile_rows=0, tile_count=0, tile_phase=0) -> COpts:
    """rt355_opts.  tile_*: the interleaved row tiles of the multi-GPU partition (tile k of the slab belongs to
    share k % tile_count; the call renders share tile_phase)."""
    return COpts(width, height, alias, row_begin, row_end, mode, seed, shading, tile_rows, tile_count, tile_phase, 0)


_M64 = 0xFFFFFFFFFFFFFFFF


def _mix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def _keyed(seed, x, y, sample, dim):
    z = _mix64(seed & 0xFFFFFFFF)
    z = _mix64(z ^ (x & 0xFFFFFFFF))
    z = _mix64(z ^ (y & 0xFFFFFFFF))
    return _mix64(z ^ (((sample & 0xFFFFFFFF) << 1) | (dim & 1)))


def jitter(seed, x, y, sample, dim) -> float:
    """rt355_jitter of include/rt355.h (the jitter of CalcPixel, main.cpp:292-293, from a counter-based generator)."""
    return (_keyed(seed, x, y, sample, dim) >> 33) / 2147483647.0


def strata(seed, x, y, which) -> float:
    """rt355_strata of include/rt355.h (k0, k1 and r of CalcPixel, main.cpp:286-288)."""
    z = _mix64(seed & 0xFFFFFFFF)
    z = _mix64(z ^ (x & 0xFFFFFFFF))
    z = _mix64(z ^ (y & 0xFFFFFFFF))
    z = _mix64(z ^ (which & 0xFFFFFFFF))
    return (_mix64(z ^ 0x737472617461) >> 33) / 2147483647.0


def rotation(seed, x, y, sample, axis) -> float:
    """rt355_rotation of include/rt355.h (random1 / random2 of shading2 and shading3, shader.h:106-107,149-150)."""
    return (_mix64(_keyed(seed, x, y, sample, axis) ^ 0x726F74617465) >> 33) / 2147483647.0


def geodesic(vertices) -> np.ndarray:
    """Geodesic() of main.cpp:408-450 after the file is read: the vertices minus their mean, normalised,
    in the reference's operation order.  Returns (n, 3) float64."""
    v = [tuple(float(c) for c in p) for p in vertices]
    sx = sy = sz = 0.0
    for x, y, z in v:                      # sum = sum + vertex, main.cpp:435
        sx, sy, sz = sx + x, sy + y, sz + z
    n = len(v)
    if n == 0:
        return np.zeros((0, 3), np.float64)
    sx, sy, sz = sx / n, sy / n, sz / n    # main.cpp:443
    out = []
    for x, y, z in v:                      # geo[i] + (Point3D(0,0,0) - sum), then normalize(): main.cpp:446-447
        out.append(_normalize((x + (0.0 - sx), y + (0.0 - sy), z + (0.0 - sz))))
    return np.array(out, dtype=np.float64).reshape(-1, 3)


def read_geodesic(path) -> np.ndarray:
    """The file half of Geodesic(): the `v x y z` lines of an OBJ file (main.cpp:425-440)."""
    verts = []
    with open(path) as f:
        for line in f:
            if line[:2] == "v ":
                parts = line[2:].split()
                verts.append((float(parts[0]), float(parts[1]), float(parts[2])))
    return geodesic(verts)


def read_obj(path, mesh_origin=(0.0, 0.0, 0.0), scale=1.0) -> np.ndarray:
    """The OBJ reader of Mesh::Mesh (anyobject.h:546-644) as (n, 15) float64 rows p0 p1 p2 uv0 uv1 uv2, ready
    for Scene.add_mesh: `v` lines become mesh_origin + p * scale, faces are read as `a b c`, then `a/p b/q c/r`,
    then `a/p/l b/q/m c/r/n` (the format escalates when a line does not fit and stays; only the first three
    corners count; a face that fits none ends the reading), UVs stay (0, 0) without `vt` lines.  A file that
    cannot be opened gives no triangles, as there.  Indices are read as decimal integers (sscanf's %i also
    takes 0x.. and 0.. prefixes, which OBJ files do not use)."""
    import re
    ox, oy, oz = (float(c) for c in mesh_origin)
    scale = float(scale)
    num = r"([+-]?\d+)"
    formats = {1: re.compile(rf"f\s+{num}\s+{num}\s+{num}"),
               2: re.compile(rf"f\s+{num}/{num}\s+{num}/{num}\s+{num}/{num}"),
               3: re.compile(rf"f\s+{num}/{num}/{num}\s+{num}/{num}/{num}\s+{num}/{num}/{num}")}
    verts, uvs, rows = [], [], []
    n_normals, face_type, has_texture = 0, 1, False

    def leading_floats(text, n):  # operator>> on an istringstream: values until one fails to parse, the rest stay 0
        out = [0.0] * n
        for i, tok in enumerate(text.split()[:n]):
            try:
                out[i] = float(tok)
            except ValueError:
                break
        return out
    try:
        f = open(path, "r", errors="replace")
    except OSError:
        return np.zeros((0, 15), np.float64)
    with f:
        for line in f.read().split("\n"):
            if line[:2] == "v ":
                x, y, z = leading_floats(line[2:], 3)
                verts.append((ox + (x - 0) * scale, oy + (y - 0) * scale, oz + (z - 0) * scale))
            elif line[:2] == "vt":
                has_texture = True
                uvs.append(tuple(leading_floats(line[2:], 2)))
            elif line[:2] == "vn":
                n_normals += 1
            elif line[:2] == "f ":
                idx = None
                while face_type <= 3:
                    m = formats[face_type].match(line)
                    if m:
                        idx = [int(g) for g in m.groups()]
                        break
                    face_type += 1
                if idx is None:
                    break
                per = face_type  # numbers per corner
                a, b, c = idx[0], idx[per], idx[2 * per]
                p, q, r = (idx[1], idx[per + 1], idx[2 * per + 1]) if per >= 2 else (1, 1, 1)
                if not all(1 <= i <= len(verts) for i in (a, b, c)):
                    raise ValueError(f"{path}: face refers to a vertex that was not read: {line}")
                uv = [(0.0, 0.0)] * 3
                if has_texture:
                    if not all(1 <= i <= len(uvs) for i in (p, q, r)):
                        raise ValueError(f"{path}: face refers to a texture coordinate that was not read: {line}")
                    uv = [uvs[p - 1], uvs[q - 1], uvs[r - 1]]
                rows.append([*verts[a - 1], *verts[b - 1], *verts[c - 1], *uv[0], *uv[1], *uv[2]])
    return np.array(rows, dtype=np.float64).reshape(-1, 15)


def read_bmp(path) -> np.ndarray:
    """Texture::Texture (texture.h:42-79): 54-byte header, width at byte 18, height at byte 22, then
    3 * width * height bytes as they come (no row padding, bottom row first), B and R swapped.  Returns
    (height, width, 3) uint8 in the reference's in-memory order, ready for Scene.add_texture."""
    with open(path, "rb") as f:
        info = f.read(54).ljust(54, b"\0")
        w = int.from_bytes(info[18:22], "little", signed=True)
        h = int.from_bytes(info[22:26], "little", signed=True)
        if w <= 0 or h <= 0:
            raise ValueError(f"{path}: bad size in the header")
        px = np.frombuffer(f.read(3 * w * h).ljust(3 * w * h, b"\0"), dtype=np.uint8).reshape(h, w, 3)
    return np.ascontiguousarray(px[:, :, ::-1])


def write_bmp(path, rgb: np.ndarray):
    """A 24-bit BMP the way Texture::Texture reads one: rgb is (height, width, 3) in memory order (row 0 = the
    file's first row); rows are NOT padded, which is what that reader assumes."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    hdr = bytearray(54)
    hdr[0:2] = b"BM"
    hdr[2:6] = (54 + 3 * w * h).to_bytes(4, "little")
    hdr[10] = 54; hdr[14] = 40; hdr[26] = 1; hdr[28] = 24
    hdr[18:22] = w.to_bytes(4, "little"); hdr[22:26] = h.to_bytes(4, "little")
    with open(path, "wb") as f:
        f.write(bytes(hdr))
        f.write(rgb[:, :, ::-1].tobytes())


def icosphere(level=1):
    """Vertices of an icosahedron subdivided `level` times (12, 42, 162, 642 ... points on the unit sphere):
    a stand-in for the reference's geodesic.obj, which its repository does not hold."""
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
             (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [_normalize(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11),
             (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                verts.append(_normalize(tuple((verts[a][i] + verts[b][i]) / 2.0 for i in range(3))))
                cache[key] = len(verts) - 1
            return cache[key]
        nf = []
        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return verts


def fnv1a64(buf) -> int:
    """FNV-1a 64 over a byte buffer (offset 0xcbf29ce484222325, prime 0x100000001b3).
    Pure Python: meant for small frames; the CPU checker exports a C version for 4K ones."""
    h = 0xCBF29CE484222325
    for b in bytes(memoryview(buf)):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h
