"""Host-side mirror of the reference's render call on top of the C ABI (include/rt355.h).

    setPixels(Scene sce, Shader shd, int alias)   -- image_synthesis/main.cpp:339-376

becomes ``Renderer(scene).render(camera, shader, width, height)``.  The HIP library is the
only backend: there is no CPU fallback, and a missing librt355.so is an error.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .binding import ERR, MODE_PACKET, SHADING_LIGHTS, CStats, COpts, PackedScene, Scene, Shader, opts as make_opts

_LIB_PATH = os.environ.get("RT355_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "librt355.so")
_lib = None


class Rt355Error(RuntimeError):
    pass


def lib():
    """Load raytracer_amd/lib/librt355.so (built by __graft_entry__.build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise Rt355Error(f"{_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        # A process must hold ONE HIP runtime.  PyTorch ships its own libamdhip64; if it is going to be used in
        # this process (bench.py, the tests) it has to be loaded first so that this library binds to the same
        # copy -- loaded the other way round, the second runtime finds no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        L.rt355_last_error.restype = C.c_char_p
        L.rt355_scene_create.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
        L.rt355_scene_create_multi.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_void_p)]
        L.rt355_scene_destroy.argtypes = [C.c_void_p]
        L.rt355_scene_destroy.restype = None
        L.rt355_render.argtypes = [C.c_void_p] * 6
        L.rt355_render_device.argtypes = [C.c_void_p] * 7
        L.rt355_render_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rt355_render_device_noalloc.argtypes = [C.c_void_p] * 7
        L.rt355_host_alloc.restype = C.c_void_p
        L.rt355_host_alloc.argtypes = [C.c_uint64]
        L.rt355_host_free.restype = None
        L.rt355_host_free.argtypes = [C.c_void_p]
        L.rt355_host_register.argtypes = [C.c_void_p, C.c_uint64]
        L.rt355_host_unregister.argtypes = [C.c_void_p]
        L.rt355_host_device_pointer.restype = C.c_void_p
        L.rt355_host_device_pointer.argtypes = [C.c_void_p]
        L.rt355_scene_sync.argtypes = [C.c_void_p]
        L.rt355_scene_search_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.rt355_scene_resolve_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
        L.rt355_scene_resolve_launch_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
        L.rt355_scene_search_launch_ms.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double)]
        L.rt355_probe_intersect.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.rt355_probe_sincos.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.rt355_scene_set_directions.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.rt355_scene_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        L.rt355_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        _lib = L
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        raise Rt355Error(f"{what}: {ERR.get(rc, rc)}: {lib().rt355_last_error().decode()}")


class Renderer:
    """A scene uploaded to the current HIP device (rt355_scene_create) or, with devices=[...], to several GPUs of this
    process (rt355_scene_create_multi): render / render_sweep then share every frame / every sweep between them."""

    def __init__(self, scene, devices=None):
        self.packed = scene.pack() if isinstance(scene, Scene) else scene
        assert isinstance(self.packed, PackedScene)
        self.handle = C.c_void_p()
        if devices is None:
            _check(lib().rt355_scene_create(C.byref(self.packed.desc), C.byref(self.handle)), "rt355_scene_create")
        else:
            devs = (C.c_int32 * len(devices))(*devices)
            _check(lib().rt355_scene_create_multi(C.byref(self.packed.desc), devs, len(devices), C.byref(self.handle)),
                   "rt355_scene_create_multi")
        self._stats = CStats()

    def close(self):
        if self.handle:
            lib().rt355_scene_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, cam, shader: Shader, width: int, height: int, alias=0, row_begin=0, row_end=0,
               mode=MODE_PACKET, out: np.ndarray | None = None, seed=0, shading=SHADING_LIGHTS,
               tile_rows=0, tile_count=0, tile_phase=0):
        """Host framebuffer path (rt355_render).  Returns (pixmap (H,W,3) uint8, stats dict)."""
        o = make_opts(width, height, alias, row_begin, row_end, mode, seed, shading, tile_rows, tile_count, tile_phase)
        if out is None:
            out = np.zeros((height, width, 3), dtype=np.uint8)
        st = CStats()
        _check(lib().rt355_render(self.handle, C.byref(cam), C.byref(shader.c), C.byref(o), out.ctypes.data,
                                  C.byref(st)), "rt355_render")
        return out, st.as_dict()

    def render_sweep(self, cams, shader: Shader, o: COpts, pixmaps):
        """rt355_render_sweep: frame i with cams[i] into pixmaps[i] ((H, W, 3) uint8 arrays, page-locked or not).
        Returns the list of the frames' stats dicts."""
        n = len(cams)
        cam_arr = (type(cams[0]) * n)(*cams) if n else None
        ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pixmaps])
        st = (CStats * n)()
        _check(lib().rt355_render_sweep(self.handle, cam_arr, n, C.byref(shader.c), C.byref(o), ptrs, st), "rt355_render_sweep")
        return [s.as_dict() for s in st]

    def set_directions(self, dirs):
        """The direction set `geo` of the gather shaders (main.cpp:639), (n, 3) float64 as Geodesic() leaves it."""
        d = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        _check(lib().rt355_scene_set_directions(self.handle, d.ctypes.data if len(d) else None, len(d)),
               "rt355_scene_set_directions")

    def set_option(self, name: str, value: int):
        """rt355_scene_set_option: a tuning option of the handle (include/rt355.h lists them)."""
        _check(lib().rt355_scene_set_option(self.handle, name.encode(), int(value)), "rt355_scene_set_option")

    def render_device(self, cam, shader: Shader, o: COpts, d_pixmap: int, stream: int = 0):
        """Device framebuffer path (rt355_render_device): enqueue only; call sync() for stats."""
        _check(lib().rt355_render_device(self.handle, C.byref(cam), C.byref(shader.c), C.byref(o),
                                         C.c_void_p(d_pixmap), C.c_void_p(stream), C.byref(self._stats)),
               "rt355_render_device")

    def render_device_noalloc(self, cam, shader: Shader, o: COpts, d_pixmap: int, stream: int = 0):
        """rt355_render_device_noalloc: the same for callers that cannot allocate (unsearched frames only)."""
        _check(lib().rt355_render_device_noalloc(self.handle, C.byref(cam), C.byref(shader.c), C.byref(o),
                                                 C.c_void_p(d_pixmap), C.c_void_p(stream), C.byref(self._stats)),
               "rt355_render_device_noalloc")

    def sync(self) -> dict:
        _check(lib().rt355_scene_sync(self.handle), "rt355_scene_sync")
        return self._stats.as_dict()

    def search_kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        _check(lib().rt355_scene_search_kernel_ms(self.handle, C.byref(ms), C.byref(n)), "search_kernel_ms")
        return ms.value, n.value

    def resolve_kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        _check(lib().rt355_scene_resolve_kernel_ms(self.handle, C.byref(ms), C.byref(n)), "resolve_kernel_ms")
        return ms.value, n.value

    def resolve_launch_ms(self):
        """Per-launch durations (ms) of the last synced render's fp64 kernels: primary resolve, then shading groups."""
        _, n = self.resolve_kernel_ms()
        out = []
        for i in range(n):
            ms = C.c_double()
            _check(lib().rt355_scene_resolve_launch_ms(self.handle, i, C.byref(ms)), "resolve_launch_ms")
            out.append(ms.value)
        return out

    def search_launch_ms(self):
        """Per-launch durations (ms) of the last synced render's search kernels."""
        _, n = self.search_kernel_ms()
        out = []
        for i in range(n):
            ms = C.c_double()
            _check(lib().rt355_scene_search_launch_ms(self.handle, i, C.byref(ms)), "search_launch_ms")
            out.append(ms.value)
        return out

    def probe_intersect(self, obj: int, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((len(rays), 9), dtype=np.float64)
        _check(lib().rt355_probe_intersect(self.handle, obj, rays.ctypes.data, len(rays), out.ctypes.data),
               "rt355_probe_intersect")
        return out


class HostPixmap:
    """A page-locked pixmap (rt355_host_alloc) as a (height, width, 3) uint8 array: the readback of rt355_render
    is then one DMA transfer into it.  Stands where the reference has `new unsigned char[width * height * 3]`
    (main.cpp:544)."""

    def __init__(self, width: int, height: int):
        n = width * height * 3
        self.ptr = lib().rt355_host_alloc(n)
        if not self.ptr:
            raise Rt355Error("rt355_host_alloc failed")
        self.array = np.ctypeslib.as_array((C.c_uint8 * n).from_address(self.ptr)).reshape(height, width, 3)
        # where kernels reach it: what rt355_render_device takes to render straight into this pixmap
        self.device_ptr = lib().rt355_host_device_pointer(self.ptr)

    def close(self):
        if self.ptr:
            self.array = None
            lib().rt355_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sincos(x: float):
    """rt355_sincos: the correctly rounded sin and cos the device uses for the gather rotation (host evaluation)."""
    sv, cv = C.c_double(), C.c_double()
    _check(lib().rt355_sincos(float(x), C.byref(sv), C.byref(cv)), "rt355_sincos")
    return sv.value, cv.value


def probe_sincos(x) -> np.ndarray:
    """rt355_probe_sincos: the same function evaluated on the device; (n, 2) float64 (sin, cos)."""
    x = np.ascontiguousarray(x, dtype=np.float64).ravel()
    out = np.zeros((len(x), 2), dtype=np.float64)
    _check(lib().rt355_probe_sincos(x.ctypes.data, len(x), out.ctypes.data), "rt355_probe_sincos")
    return out
