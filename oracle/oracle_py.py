"""Loader for the CPU checker (librt_oracle.so) and, where it was built, the harness
around the reference's own headers (oracle/_ref/libref_harness.so).

TEST INFRASTRUCTURE.  Imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg only -- never by the raytracer_amd package.
"""
from __future__ import annotations

import ctypes as C
import os
import tempfile

import numpy as np

from raytracer_amd.binding import CStats, CTexture, COpts

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "librt_oracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref_harness.so")


class Oracle:
    def __init__(self, path: str = ORACLE_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.rt_oracle_render.restype = C.c_int
        L.rt_oracle_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_int]
        L.rt_oracle_render_gather.restype = C.c_int
        L.rt_oracle_render_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.rt_oracle_sincos.restype = None
        L.rt_oracle_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_oracle_sincos_rounded.restype = None
        L.rt_oracle_sincos_rounded.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.rt_oracle_fnv1a64.restype = C.c_uint64
        L.rt_oracle_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        L.rt_oracle_intersect.restype = C.c_int
        L.rt_oracle_intersect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt_oracle_view.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.rt_oracle_color_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                         C.c_double, C.c_void_p]
        L.rt_oracle_texture.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p]
        self.L = L

    def render(self, packed, cam, shader, o: COpts, threads=0, stride=1, out=None, dirs=None):
        """Returns (pixmap (H,W,3) uint8, stats dict).  dirs: the gather shaders' direction set, (n, 3) float64."""
        if out is None:
            out = np.zeros((o.height, o.width, 3), dtype=np.uint8)
        st = CStats()
        d = np.ascontiguousarray(dirs if dirs is not None else np.zeros((0, 3)), dtype=np.float64).reshape(-1, 3)
        rc = self.L.rt_oracle_render_gather(C.byref(packed.desc), C.byref(cam), C.byref(shader.c), C.byref(o),
                                            d.ctypes.data if len(d) else None, len(d),
                                            out.ctypes.data, C.byref(st), threads, stride)
        if rc != 0:
            raise RuntimeError(f"rt_oracle_render failed: {rc}")
        return out, st.as_dict()

    def sincos(self, x: float, rounded=False):
        """The C library's sin and cos, or (rounded) the correctly rounded ones from binary128."""
        sv, cv = C.c_double(), C.c_double()
        (self.L.rt_oracle_sincos_rounded if rounded else self.L.rt_oracle_sincos)(float(x), C.byref(sv), C.byref(cv))
        return sv.value, cv.value

    def fnv(self, arr: np.ndarray) -> int:
        a = np.ascontiguousarray(arr)
        return int(self.L.rt_oracle_fnv1a64(a.ctypes.data, a.nbytes))

    def intersect(self, packed, obj: int, pe, npe) -> np.ndarray:
        pe = np.asarray(pe, dtype=np.float64); npe = np.asarray(npe, dtype=np.float64)
        out = np.zeros(9, dtype=np.float64)
        rc = self.L.rt_oracle_intersect(C.byref(packed.desc), obj, pe.ctypes.data, npe.ctypes.data,
                                        out.ctypes.data)
        assert rc == 0
        return out

    def view(self, cam, W, H, x, y) -> np.ndarray:
        out = np.zeros(6, dtype=np.float64)
        self.L.rt_oracle_view(C.byref(cam), W, H, x, y, out.ctypes.data)
        return out

    def color_op(self, op, c, c2=(0, 0, 0), v=0.0, v2=0.0, v3=0.0):
        a = np.array(c, dtype=np.uint8); b = np.array(c2, dtype=np.uint8)
        out = np.zeros(3, dtype=np.uint8)
        self.L.rt_oracle_color_op(op, a.ctypes.data, b.ctypes.data, v, v2, v3, out.ctypes.data)
        return out

    def texture(self, tex, kind, a, b):
        rgb = np.ascontiguousarray(tex.rgb)
        ct = CTexture(rgb.ctypes.data, tex.sx, tex.sy, rgb.shape[1], rgb.shape[0])
        out = np.zeros(3, dtype=np.uint8)
        self.L.rt_oracle_texture(C.byref(ct), kind, a, b, out.ctypes.data)
        return out


class Reference:
    """The reference's own CalcIntersect / Shader::shading / Texture code (this container only)."""

    def __init__(self, path: str = REF_SO):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        L = C.CDLL(path)
        L.ref_render.restype = C.c_int
        L.ref_render.argtypes = [C.c_void_p] * 5 + [C.c_char_p]
        L.ref_render_gather.restype = C.c_int
        L.ref_render_gather.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_char_p]
        L.ref_intersect_many.restype = C.c_int
        L.ref_intersect_many.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
        L.ref_color_op.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                   C.c_void_p]
        L.ref_texture.restype = C.c_int
        L.ref_texture.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p]
        L.ref_adaptor_roundtrip.restype = C.c_int
        L.ref_adaptor_roundtrip.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.ref_load_mesh.restype = C.c_int
        L.ref_load_mesh.argtypes = [C.c_char_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
        L.ref_load_texture.restype = C.c_int
        L.ref_load_texture.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_int]
        self.L = L
        self.tmp = tempfile.mkdtemp(prefix="rt355_ref_")

    @staticmethod
    def available() -> bool:
        return os.path.exists(REF_SO)

    def render(self, packed, cam, shader, o: COpts, dirs=None):
        out = np.zeros((o.height, o.width, 3), dtype=np.uint8)
        d = np.ascontiguousarray(dirs if dirs is not None else np.zeros((0, 3)), dtype=np.float64).reshape(-1, 3)
        rc = self.L.ref_render_gather(C.byref(packed.desc), C.byref(cam), C.byref(shader.c), C.byref(o),
                                      d.ctypes.data if len(d) else None, len(d), out.ctypes.data, self.tmp.encode())
        if rc != 0:
            raise RuntimeError(f"ref_render failed: {rc}")
        return out

    def load_mesh(self, path, origin=(0.0, 0.0, 0.0), scale=1.0, cap=100000) -> np.ndarray:
        """Mesh::Mesh(path, origin, scale) of the reference: (n, 15) float64 rows p0 p1 p2 uv0 uv1 uv2."""
        o = np.array(origin, dtype=np.float64)
        out = np.zeros((cap, 15), dtype=np.float64)
        n = self.L.ref_load_mesh(str(path).encode(), o.ctypes.data, float(scale), out.ctypes.data, cap)
        assert 0 <= n <= cap
        return out[:n].copy()

    def load_texture(self, path) -> np.ndarray:
        """Texture::Texture(path) of the reference: (height, width, 3) uint8 in memory order (power-of-two sizes)."""
        w, h = C.c_int(), C.c_int()
        out = np.zeros(1 << 22, dtype=np.uint8)
        rc = self.L.ref_load_texture(str(path).encode(), C.byref(w), C.byref(h), out.ctypes.data, out.nbytes)
        if rc != 0:
            raise RuntimeError(f"ref_load_texture failed: {rc}")
        return out[: w.value * h.value * 3].reshape(h.value, w.value, 3).copy()

    def adaptor_roundtrip(self, packed, shader) -> int:
        return int(self.L.ref_adaptor_roundtrip(C.byref(packed.desc), C.byref(shader.c), self.tmp.encode()))

    def intersect_many(self, packed, obj: int, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((len(rays), 9), dtype=np.float64)
        rc = self.L.ref_intersect_many(C.byref(packed.desc), obj, rays.ctypes.data, len(rays),
                                       out.ctypes.data, self.tmp.encode())
        assert rc == 0
        return out

    def color_op(self, op, c, c2=(0, 0, 0), v=0.0, v2=0.0, v3=0.0):
        a = np.array(c, dtype=np.uint8); b = np.array(c2, dtype=np.uint8)
        out = np.zeros(3, dtype=np.uint8)
        self.L.ref_color_op(op, a.ctypes.data, b.ctypes.data, v, v2, v3, out.ctypes.data)
        return out

    def texture_many(self, tex, kind, ab: np.ndarray) -> np.ndarray:
        rgb = np.ascontiguousarray(tex.rgb)
        ct = CTexture(rgb.ctypes.data, tex.sx, tex.sy, rgb.shape[1], rgb.shape[0])
        ab = np.ascontiguousarray(ab, dtype=np.float64).reshape(-1, 2)
        out = np.zeros((len(ab), 3), dtype=np.uint8)
        rc = self.L.ref_texture(C.byref(ct), kind, ab.ctypes.data, len(ab), out.ctypes.data, self.tmp.encode())
        assert rc == 0
        return out
